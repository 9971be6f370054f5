"""The numpy oracle's per-row (Triton) half against vectors produced by the reference's own
``_per_row_quant_kernel`` / ``_per_row_post_process_mm`` under Triton's CPU interpreter
(tests/golden/triton_golden.npz, generator: tests/golden/make_golden_triton.py).

The interpreter's narrowing casts are not the compiled kernels' (truncation to bf16, a carry-less
round-half-up to e4m3; compiled Triton rounds to nearest even like torch), so every comparison applies the
interpreter's cast (tests/util.py) to the oracle's PRE-cast f32 values and demands bit equality; the oracle's
own round-to-nearest-even results are then checked to differ from the vectors only by that last rounding."""
import numpy as np
import pytest

from oracle import qlinear_oracle as O
from util import interp_f32_to_bf16, interp_f32_to_e4m3

CASES = ["m1_n64_k128", "m3_n64_k256", "m16_n48_k512", "m130_n64_k256", "m5_n8_k3072", "m130_n200_k384",
         "m7_n130_k128"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("side", ["w", "x"])
def test_fp8_per_row_quant(triton_golden, case, side):
    g = triton_golden
    t = g[f"{case}/{side}"]
    gq, gs = g[f"{case}/fp8/q{side}"], g[f"{case}/fp8/s{side}"]
    y, s = O.per_row_quant(t, "fp8", return_scaled=True)
    assert np.array_equal(s, gs)                               # row_max / 448 incl. the 1e-8 floor: bit-exact
    bits, valid = interp_f32_to_e4m3(y)
    assert valid.mean() > 0.98
    assert np.array_equal(bits[valid], gq[valid])              # x * (448 / row_max) up to the bits the cast keeps
    # the shipped (round-to-nearest-even) result differs from the vector only where the interpreter's
    # carry-less rounding does
    q, _ = O.per_row_quant(t, "fp8")
    diff = (q != gq) & valid
    m = (y.view(np.uint32) >> np.uint32(20)) & np.uint32(7)
    cut = (y.view(np.uint32) & np.uint32(1 << 19)) > 0
    tie_even = cut & ((y.view(np.uint32) & np.uint32((1 << 19) - 1)) == 0) & (m % 2 == 0)
    explained = (cut & (m == 7)) | tie_even
    assert not (diff & ~explained).any()
    assert diff.mean() < 0.05


@pytest.mark.parametrize("case", CASES)
def test_int8_gemm_bit_exact(triton_golden, case):
    """int32 accumulation and the epilogue acc * sa[m] * sb[n] (in that order, f32): bit-exact before the cast."""
    g = triton_golden
    qx, sx, qw, sw = (g[f"{case}/int8/{k}"] for k in ("qx", "sx", "qw", "sw"))
    c = O.w8a8_gemm(qx, sx, qw, sw, None, "int8", return_f32=True)
    assert np.array_equal(interp_f32_to_bf16(c), g[f"{case}/int8/y"])
    # and the oracle's own operands are the ones stored (they were produced by it: libdevice.round is not interpretable)
    q, s = O.per_row_quant(g[f"{case}/x"], "int8")
    assert np.array_equal(q.view(np.uint8), qx) and np.array_equal(s, sx)
    # RTNE output = truncated output or its successor in magnitude
    y = O.w8a8_gemm(qx, sx, qw, sw, None, "int8").astype(np.int32)
    d = y - g[f"{case}/int8/y"].astype(np.int32)
    assert ((d == 0) | (d == 1)).all()


@pytest.mark.parametrize("case", CASES)
def test_fp8_gemm(triton_golden, case):
    """fp8 x fp8 with f32 accumulation (order unspecified: the interpreter multiplies 128-k tiles with numpy).
    Tolerance: one bf16 ulp for the truncating cast plus 2**-18 * sum_k|a*b| * sa * sb for the f32 sums."""
    g = triton_golden
    qx, sx, qw, sw = (g[f"{case}/fp8/{k}"] for k in ("qx", "sx", "qw", "sw"))
    c = O.w8a8_gemm(qx, sx, qw, sw, None, "fp8", return_f32=True).astype(np.float64)
    got = O.bf16_to_f32(g[f"{case}/fp8/y"]).astype(np.float64)
    l1 = (np.abs(O.fp8_to_f32(qx)).astype(np.float64) @ np.abs(O.fp8_to_f32(qw)).astype(np.float64).T)
    l1 = l1 * sx[:, None].astype(np.float64) * sw[None, :].astype(np.float64)
    tol = 2.0 ** -7 * np.abs(c) + 2.0 ** -18 * l1 + 1e-30
    assert (np.abs(got - c) <= tol).all()
    assert (np.abs(got) <= np.abs(c) + 2.0 ** -18 * l1 + 1e-30).all()     # truncation never grows the magnitude
