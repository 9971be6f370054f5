// Instantiates the bf16 x bf16 kernels of the output projection (plain logits, and the greedy sampler in the
// epilogue); compiled in parallel with the other variants, see ql_gemm_kernel.h.
#include "ql_gemm_kernel.h"
namespace ql {
kern_t pick_head(int mt, int wn, int wk, int cp, bool amax) {
    return amax ? pick_head_impl<true>(mt, wn, wk, cp) : pick_head_impl<false>(mt, wn, wk, cp);
}
}  // namespace ql
