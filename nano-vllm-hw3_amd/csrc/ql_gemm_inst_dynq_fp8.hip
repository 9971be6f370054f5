// Instantiates the fp8 x fp8 kernels that quantize their bf16 activations while staging them (see ql_gemm_kernel.h).
#include "ql_gemm_kernel.h"
namespace ql {
kern_t pick_fp8_fp8_dynq(int mt, int wn, int wk, int cp) { return pick_dynq_impl<ACT_FP8, true>(mt, wn, wk, cp); }
}  // namespace ql
