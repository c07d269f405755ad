// The REAL util/src/types.h of the reference (aliases + CreateDescriptorMatrixd), reached by a computed include because the
// stub tree under it shadows that name (oracle/Makefile passes -DPXO_REF_ROOT=<reference checkout>).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
#define PXO_STR2(x) #x
#define PXO_STR(x) PXO_STR2(x)
#include PXO_STR(PXO_REF_ROOT/pixsfm/util/src/types.h)
