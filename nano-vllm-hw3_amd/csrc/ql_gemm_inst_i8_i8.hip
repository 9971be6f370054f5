// Instantiates the fused dequant GEMM kernels for one arithmetic variant (compiled in parallel with
// the other variants; see ql_gemm_kernel.h).
#include "ql_gemm_kernel.h"
namespace ql {
kern_t pick_i8_i8(int mt, int wn, int wk, int cp, bool rope) { return pick_kernel_impl<ACT_I8, false, false>(mt, wn, wk, cp, rope); }
}  // namespace ql
