#pragma once
#include "highfive/mini_highfive.hpp"
