#pragma once
#include "../../../kasetup/base/src/parallel_optimizer.h"
