#pragma once
#include <unordered_map>
#include <unordered_set>
#include "../../../interp/colmap/util/types.h"
