#pragma once
#include "colmap/base/camera_models.h"
