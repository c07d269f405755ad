#pragma once
#include <cmath>
