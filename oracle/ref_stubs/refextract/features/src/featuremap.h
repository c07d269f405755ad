#pragma once
#include "../../../kasetup/features/src/featureset.h"
