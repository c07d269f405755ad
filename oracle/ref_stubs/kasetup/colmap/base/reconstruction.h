#pragma once
#include "colmap/base/projection.h"
