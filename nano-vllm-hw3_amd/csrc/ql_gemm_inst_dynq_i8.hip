// Instantiates the int8 x int8 kernels that quantize their bf16 activations while staging them (see ql_gemm_kernel.h).
#include "ql_gemm_kernel.h"
namespace ql {
kern_t pick_i8_i8_dynq(int mt, int wn, int wk, int cp) { return pick_dynq_impl<ACT_I8, false>(mt, wn, wk, cp); }
}  // namespace ql
