// Placeholder translation unit for the fused elementwise operators around the quantized GEMMs.
#include "ql_common.h"
