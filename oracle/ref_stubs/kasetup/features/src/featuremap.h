#pragma once
#include "features/src/featureset.h"
