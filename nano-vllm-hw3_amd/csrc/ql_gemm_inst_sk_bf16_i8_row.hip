// Instantiates the weight-only kernels on the stream-K schedule for one arithmetic variant (compiled in parallel
// with the other variants; see ql_gemm_kernel.h).
#include "ql_gemm_kernel.h"
namespace ql {
kern_t pick_sk_bf16_i8_row(int mt, int wn, int wk, int cp, bool rope) { return pick_sk_impl<false, false>(mt, wn, wk, cp, rope); }
}  // namespace ql
