// Stub of util/src/log_exceptions.h for base/src/irls_optim.h: the two checks it makes.
#pragma once
#include <stdexcept>
#define THROW_CHECK_GT(a, b) do { if (!((a) > (b))) throw std::invalid_argument(#a " > " #b); } while (0)
#define THROW_CHECK_EQ(a, b) do { if (!((a) == (b))) throw std::invalid_argument(#a " == " #b); } while (0)
