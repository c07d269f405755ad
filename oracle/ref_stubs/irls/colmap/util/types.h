#pragma once
#include "../../../colmap/util/types.h"
