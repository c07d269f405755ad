#pragma once
#include "pybind11/pybind11.h"
