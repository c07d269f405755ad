#pragma once
#include "../../h5reader/pybind11/pybind11.h"
