#pragma once
#include "../../../basetup/colmap/util/timer.h"
