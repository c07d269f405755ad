#pragma once
#include "colmap/base/reconstruction.h"
