import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
from nanovllm import _qlinear as Q, ops
from nanovllm.utils import quantization as NQ
lib = Q.load()
Q.workspace.reserve("cuda", 128 << 20)
M, I, K = 150, 160, 1152
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
w = (torch.randn(2 * I, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
il = NQ.interleave_gate_up
qw, sw = NQ.per_row_quant(w, torch.int8)
for tile in [(1, 2, 4, 4), (2, 2, 4, 4), (2, 4, 2, 4), (4, 8, 1, 2)]:
    for G in (16, 40):
        lib.ql_gemm_force_config(*tile, -G)
        plain = NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)
        plain2 = NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)
        ref = ops.silu_mul(plain)
        got = NQ.per_row_weight_only_matmul(x, il(qw), il(sw), None, torch.int8, silu=True)
        got2 = NQ.per_row_weight_only_matmul(x, il(qw), il(sw), None, torch.int8, silu=True)
        # the interleaved weight as a plain GEMM: columns permuted
        pil = NQ.per_row_weight_only_matmul(x, il(qw), il(sw), None, torch.int8)
        lib.ql_gemm_force_config(*tile, 1)
        plain_c = NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)
        got_c = NQ.per_row_weight_only_matmul(x, il(qw), il(sw), None, torch.int8, silu=True)
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        bad = (got != ref).nonzero()
        print(tile, G, "plan", Q.plan_query(2, M, 2 * I, K), "plain==plain2", torch.equal(plain, plain2), "got==got2", torch.equal(got, got2),
              "mismatch", bad.shape[0], "classic ok", torch.equal(got_c, ops.silu_mul(plain_c)),
              "plain sk vs classic maxdiff", float((plain.float() - plain_c.float()).abs().max()),
              "il plain consistent", torch.equal(il(plain.t().contiguous()).t(), pil))
        if bad.shape[0]:
            rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
            print("   rows", rows[:20], "cols", cols[:40])
