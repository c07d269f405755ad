#pragma once
#include "../../../kasetup/util/src/statistics.h"
