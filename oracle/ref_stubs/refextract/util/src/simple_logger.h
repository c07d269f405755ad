#pragma once
#include "../../../h5reader/util/src/simple_logger.h"
