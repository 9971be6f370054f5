"""The allocation-free entry point ql_gemm_w8a16_noalloc against ql_gemm_w8a16 (same kernels, workspace inside the
library): bit-identical outputs on two calls in a row, the second with a smaller input, through the C ABI, and both
against the CPU oracle.

The allocation-free entry point plans with the heuristics only (its fixed scratch is sized for their splits), while
ql_gemm_w8a16 consults the table of measured plans first; another tile or split changes the f32 summation order.  The
reference call is therefore made with the plan table cleared (and the table is reloaded afterwards), so that both
routes run the same plan whatever the shipped table holds."""
import contextlib

import numpy as np
import pytest
import torch

from oracle import qlinear_oracle as O
from tests.util import assert_gemm_close, bits16

pytestmark = pytest.mark.gpu

NOALLOC_BYTES = 16 << 20


def _run(lib, Q, fn_noalloc, x, qw, sw, bias, gran, group):
    M, K = x.shape
    N = qw.shape[0]
    s = torch.cuda.current_stream().cuda_stream
    y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    b = bias.data_ptr() if bias is not None else None
    if fn_noalloc:
        Q.check(lib.ql_gemm_w8a16_noalloc(Q.QL_INT8, gran, x.data_ptr(), qw.data_ptr(), sw.data_ptr(), b, y.data_ptr(),
                                          M, N, K, group, s))
    else:
        need = lib.ql_gemm_workspace_bytes(M, N, K, gran)
        ws = torch.zeros(max(need, 16), dtype=torch.uint8, device="cuda")
        Q.check(lib.ql_gemm_w8a16(Q.QL_INT8, gran, x.data_ptr(), qw.data_ptr(), sw.data_ptr(), b, y.data_ptr(), M, N, K,
                                  group, ws.data_ptr() if need else None, need, s))
    torch.cuda.synchronize()
    return y


@contextlib.contextmanager
def heuristics_only(lib, Q):
    """No plan-table entries while the block runs: every entry point plans with the built-in heuristics."""
    lib.ql_gemm_plan_clear()
    try:
        yield
    finally:
        Q.load_plan_table(Q.PLAN_TABLE, replace=True)


def _oracle_check(y, x, qw, sw, bias, gran, group, what):
    xb, qn, sn = bits16(x), qw.cpu().numpy(), sw.cpu().numpy()
    bb = None if bias is None else bits16(bias)
    if gran == 0:       # QL_PER_ROW
        ref = O.w8a16_rowscale_matmul(xb, qn, sn, bb, "int8", return_f64=True)
        l1 = O.abs_product_sum(xb, qn.astype(np.float32) * sn[:, None])
    else:
        ref = O.per_group_matmul(xb, qn, sn, bb, group, "int8", return_f64=True)
        deq = (qn.astype(np.float32).reshape(qn.shape[0], -1, group) * sn[..., None]).reshape(qn.shape)
        l1 = O.abs_product_sum(xb, deq)
    if bb is not None:                                               # (the f64 forms return the sum before the bias)
        ref = ref + O.bf16_to_f32(bb).astype(np.float64)[None, :]
        l1 = l1 + np.abs(O.bf16_to_f32(bb)).astype(np.float64)[None, :] * 2.0 ** 10   # (half an ulp of the bias itself)
    assert_gemm_close(bits16(y), ref, l1, what=what, double_round=(gran != 0 and bias is not None))


def test_noalloc_matches_workspace_entry_point_twice():
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    lib = Q.load()
    assert Q.QL_PER_ROW == 0
    g = torch.Generator(device="cuda").manual_seed(0)
    # (M, N, K, granularity, group): split-K shapes first (down_proj of Qwen3-0.6B splits K three ways), the second
    # call of each pair is smaller in every dimension; then a per-group problem and one that does not split at all
    calls = [(128, 1024, 3072, Q.QL_PER_ROW, 0), (48, 512, 2048 + 1024, Q.QL_PER_ROW, 0), (16, 256, 1024, Q.QL_PER_ROW, 0),
             (64, 1024, 3072, Q.QL_PER_GROUP, 128), (7, 96, 4096, Q.QL_PER_GROUP, 64), (200, 4096, 1024, Q.QL_PER_ROW, 0)]
    for (M, N, K, gran, group) in calls:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        if gran == Q.QL_PER_ROW:
            qw, sw = NQ.per_row_quant(w, torch.int8)
        else:
            qw, sw = NQ.per_group_quant(w, group, torch.int8)
        for b in (None, bias):
            with heuristics_only(lib, Q):
                assert lib.ql_gemm_plan_count() == 0
                ref = _run(lib, Q, False, x, qw, sw, b, gran, group)
            assert torch.isfinite(ref.float()).all()
            got = _run(lib, Q, True, x, qw, sw, b, gran, group)
            assert torch.equal(got, ref), (M, N, K, gran)
            _oracle_check(got, x, qw, sw, b, gran, group, f"noalloc {M}x{N}x{K} gran {gran}")
            again = _run(lib, Q, True, x, qw, sw, b, gran, group)       # the scratch now holds this call's own leftovers
            assert torch.equal(again, ref), (M, N, K, gran, "second call")


def test_noalloc_allocates_nothing_and_is_capturable():
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    lib = Q.load()
    M, N, K = 96, 1024, 3072
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    qw, sw = NQ.per_row_quant((torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16), torch.int8)
    with heuristics_only(lib, Q):
        ref = _run(lib, Q, False, x, qw, sw, None, Q.QL_PER_ROW, 0)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    _run(lib, Q, True, x, qw, sw, None, Q.QL_PER_ROW, 0)             # first call on the device: outside capture
    free0 = torch.cuda.mem_get_info()[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Q.check(lib.ql_gemm_w8a16_noalloc(Q.QL_INT8, Q.QL_PER_ROW, x.data_ptr(), qw.data_ptr(), sw.data_ptr(), None,
                                          y.data_ptr(), M, N, K, 0, torch.cuda.current_stream().cuda_stream))
    for _ in range(3):
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
    assert torch.cuda.mem_get_info()[0] >= free0 - (4 << 20)          # nothing of the scratch's size was allocated


def test_noalloc_rejects_oversized_problems():
    from nanovllm import _qlinear as Q
    lib = Q.load()
    # Qwen3-4B down_proj at 4096 rows: 32 x 20 tiles x 3 slices x 128 KiB of slabs, beyond the 16 MiB array
    M, N, K = 4096, 2560, 9728
    with heuristics_only(lib, Q):                                     # (what the entry point itself plans with)
        need = lib.ql_gemm_workspace_bytes(M, N, K, Q.QL_PER_ROW)
    if need <= NOALLOC_BYTES:
        pytest.skip(f"the plan for this shape needs only {need} bytes")
    x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    qw = torch.zeros(N, K, device="cuda", dtype=torch.int8)
    sw = torch.ones(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rc = lib.ql_gemm_w8a16_noalloc(Q.QL_INT8, Q.QL_PER_ROW, x.data_ptr(), qw.data_ptr(), sw.data_ptr(), None, y.data_ptr(),
                                   M, N, K, 0, None)
    assert rc == -3 and b"built-in buffer" in lib.ql_last_error()
