#pragma once
#include "util/src/types.h"
namespace pixsfm { struct Reference { DescriptorMatrixXd descriptor; const double* DescriptorData() const { return descriptor.data(); } }; }
