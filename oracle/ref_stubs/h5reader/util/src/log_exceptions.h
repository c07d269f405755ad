// Shadows the reference's util/src/log_exceptions.h (pybind11 exception plumbing): failed checks THROW, like there.
#pragma once
#include <stdexcept>
#include <string>
#define PXO_STUB_THROW_CHECK(cond) { if (!(cond)) throw std::invalid_argument(std::string("THROW_CHECK failed: ") + #cond); }
#define THROW_CHECK(c) PXO_STUB_THROW_CHECK(c)
#define THROW_CHECK_NE(a, b) PXO_STUB_THROW_CHECK((a) != (b))
#define THROW_CHECK_EQ(a, b) PXO_STUB_THROW_CHECK((a) == (b))
#define THROW_CHECK_GE(a, b) PXO_STUB_THROW_CHECK((a) >= (b))
#define THROW_CHECK_LT(a, b) PXO_STUB_THROW_CHECK((a) < (b))
#define THROW_CHECK_GT(a, b) PXO_STUB_THROW_CHECK((a) > (b))
#define THROW_CHECK_LE(a, b) PXO_STUB_THROW_CHECK((a) <= (b))
#define THROW_CHECK_MSG(c, m) PXO_STUB_THROW_CHECK(c)
#define THROW_CUSTOM_CHECK_MSG(c, e, m) PXO_STUB_THROW_CHECK(c)
#define THROW_EXCEPTION(exception, msg) throw exception(msg)
