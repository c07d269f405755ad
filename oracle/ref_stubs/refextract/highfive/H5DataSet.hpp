#pragma once
#include "highfive/H5Object.hpp"
