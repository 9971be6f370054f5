"""GPU parity: the HIP library (through its C ABI) against the numpy oracle and the committed golden
vectors.  Integer/byte results (quantizers, int8 GEMM) must be bit-exact; floating-point GEMMs must
sit within half a bf16 ulp of the exactly-evaluated reference formula plus an explicit
f32-accumulation allowance (see tests/util.py::assert_gemm_close)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import qlinear_oracle as O
from tests.util import TDT, assert_gemm_close, bf16_t, bits8, bits16, q_t, rand_bf16

pytestmark = pytest.mark.gpu

CASES = ["m1_n64_k128", "m3_n64_k256", "m16_n48_k512", "m130_n64_k256", "m5_n8_k3072"]
DTYPES = ["int8", "fp8"]


@pytest.fixture(scope="module")
def NQ():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from nanovllm.utils import quantization as NQ
    from nanovllm import _qlinear as Q
    Q.load()
    return NQ


def fp8_w8a8_exact(xb, qw, sw, bias=None):
    """Exactly evaluated fp8 W8A8 forward (dynamic per-row activation quantization by the oracle, f64
    contraction, f64 scales) and the L1 bound of its f32 accumulation: the (ref, l1) pair assert_gemm_close takes."""
    qx, sx = O.per_row_quant(xb, "fp8")
    acc = O.w8a8_gemm(qx, sx, qw, sw, None, "fp8", return_acc=True)
    sc = sx[:, None].astype(np.float64) * np.asarray(sw, np.float64)[None, :]
    ref = acc * sc
    l1 = (np.abs(O.fp8_to_f32(qx)).astype(np.float64) @ np.abs(O.fp8_to_f32(qw)).astype(np.float64).T) * sc
    if bias is not None:
        ref = ref + O.bf16_to_f32(bias)[None, :]
        l1 = l1 + np.abs(O.bf16_to_f32(bias))[None, :]
    return ref, l1


# ----------------------------------------------------------------------------- quantizers
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_quantizers_match_golden_bit_exact(NQ, golden, case, dt):
    w = bf16_t(golden[f"{case}/w"])
    q, s = NQ.per_tensor_quant(w, TDT[dt])
    assert np.array_equal(bits8(q), golden[f"{case}/{dt}/tensor/qw"])
    assert float(s.float()) == golden[f"{case}/{dt}/tensor/sw"][0]
    assert s.dtype == torch.bfloat16 and s.dim() == 0
    for g in (64, 128):
        q, s = NQ.per_group_quant(w, g, TDT[dt])
        assert np.array_equal(bits8(q), golden[f"{case}/{dt}/g{g}/qw"])
        assert np.array_equal(s.cpu().numpy(), golden[f"{case}/{dt}/g{g}/sw"])


@pytest.mark.parametrize("case", CASES[:2])
@pytest.mark.parametrize("dt", DTYPES)
def test_fake_quant_matches_golden_bit_exact(NQ, golden, case, dt):
    w = bf16_t(golden[f"{case}/w"])
    assert np.array_equal(bits16(NQ.fake_per_row_quant(w, TDT[dt])), golden[f"{case}/{dt}/fake_row/w"])
    assert np.array_equal(bits16(NQ.fake_per_tensor_quant(w, TDT[dt])), golden[f"{case}/{dt}/fake_tensor/w"])
    assert np.array_equal(bits16(NQ.fake_per_group_quant(w, 128, TDT[dt])), golden[f"{case}/{dt}/fake_g128/w"])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(1, 8), (5, 136), (256, 1024), (130, 3072), (1, 16384), (33, 2048)])
def test_per_row_quant_bit_exact_vs_oracle(NQ, dt, shape):
    rng = np.random.default_rng(hash((dt, shape)) % 2 ** 31)
    x = rand_bf16(rng, shape, std=rng.choice([0.02, 1.0, 30.0]))
    x[0, :] = 0                          # all-zero row: amax floor 1e-8
    if shape[0] > 2:
        x[2, : shape[1] // 2] = x[2, 0]  # many equal values / ties on the grid
    q, s = NQ.per_row_quant(bf16_t(x), TDT[dt])
    qo, so = O.per_row_quant(x, dt)
    assert np.array_equal(s.cpu().numpy(), so)
    assert np.array_equal(bits8(q), qo.view(np.uint8))


@pytest.mark.parametrize("dt", DTYPES)
def test_group_and_tensor_quant_bit_exact_vs_oracle_large(NQ, dt):
    rng = np.random.default_rng(5)
    w = rand_bf16(rng, (1024, 3072), std=0.02)
    w[3, 256:384] = 0
    for g in (32, 64, 128, 256, 512):
        q, s = NQ.per_group_quant(bf16_t(w), g, TDT[dt])
        qo, so = O.per_group_quant(w, g, dt)
        assert np.array_equal(s.cpu().numpy(), so), g
        assert np.array_equal(bits8(q), qo.view(np.uint8)), g
    q, s = NQ.per_tensor_quant(bf16_t(w), TDT[dt])
    qo, so = O.per_tensor_quant(w, dt)
    assert s.float().item() == np.float32(so).item()
    assert np.array_equal(bits8(q), qo.view(np.uint8))
    for fn_g, fn_o in ((lambda t: NQ.fake_per_row_quant(t, TDT[dt]), lambda a: O.fake_per_row_quant(a, dt)),
                       (lambda t: NQ.fake_per_group_quant(t, 128, TDT[dt]), lambda a: O.fake_per_group_quant(a, 128, dt)),
                       (lambda t: NQ.fake_per_tensor_quant(t, TDT[dt]), lambda a: O.fake_per_tensor_quant(a, dt))):
        assert np.array_equal(bits16(fn_g(bf16_t(w))), fn_o(w))


# ----------------------------------------------------------------------------- W8A16 GEMMs
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_w8a16_group_and_tensor_match_golden(NQ, golden, case, dt):
    """per_group_matmul / per_tensor_matmul on the reference's own quantized weights and outputs."""
    xb, bb = golden[f"{case}/x"], golden[f"{case}/b"]
    x, b = bf16_t(xb), bf16_t(bb)
    for g in (64, 128):
        qw, sw = golden[f"{case}/{dt}/g{g}/qw"], golden[f"{case}/{dt}/g{g}/sw"]
        deq = (O._decode_q(qw, dt).reshape(qw.shape[0], -1, g) * sw[..., None]).reshape(qw.shape)
        l1 = O.abs_product_sum(xb, deq)
        y = NQ.per_group_matmul(x, q_t(qw, dt), torch.from_numpy(sw).cuda(), None, g, TDT[dt])
        assert_gemm_close(bits16(y), O.per_group_matmul(xb, qw, sw, None, g, dt, return_f64=True), l1, what=f"g{g}")
        # and directly against the reference's bf16 output: both are roundings of the same number
        ref = O.bf16_to_f32(golden[f"{case}/{dt}/g{g}/y"]).astype(np.float64)
        assert_gemm_close(bits16(y), ref, l1, double_round=True, what=f"g{g} vs golden")
        yb = NQ.per_group_matmul(x, q_t(qw, dt), torch.from_numpy(sw).cuda(), b, g, TDT[dt])
        refb = O.bf16_to_f32(golden[f"{case}/{dt}/g{g}/y_bias"]).astype(np.float64)
        assert_gemm_close(bits16(yb), refb, l1 + np.abs(O.bf16_to_f32(bb))[None, :], double_round=True, what="bias")
    qw, sw = golden[f"{case}/{dt}/tensor/qw"], golden[f"{case}/{dt}/tensor/sw"]
    l1 = O.abs_product_sum(xb, O._decode_q(qw, dt) * sw[0])
    y = NQ.per_tensor_matmul(x, q_t(qw, dt), torch.tensor(sw[0], dtype=torch.bfloat16).cuda(), None, TDT[dt])
    assert_gemm_close(bits16(y), O.per_tensor_matmul(xb, qw, sw[0], None, dt, return_f64=True), l1, what="tensor")
    ref = O.bf16_to_f32(golden[f"{case}/{dt}/tensor/y"]).astype(np.float64)
    assert_gemm_close(bits16(y), ref, l1, double_round=True, what="tensor vs golden")


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_w8a16_per_row_vs_oracle_and_fake_route(NQ, golden, case, dt):
    xb, wb = golden[f"{case}/x"], golden[f"{case}/w"]
    qw, sw = NQ.per_row_quant(bf16_t(wb), TDT[dt])
    qo, so = O.per_row_quant(wb, dt)
    assert np.array_equal(bits8(qw), qo.view(np.uint8)) and np.array_equal(sw.cpu().numpy(), so)
    y = NQ.per_row_weight_only_matmul(bf16_t(xb), qw, sw, None, TDT[dt])
    deq = O._decode_q(qo, dt) * so[:, None]
    assert_gemm_close(bits16(y), O.w8a16_rowscale_matmul(xb, qo, so, None, dt, return_f64=True),
                      O.abs_product_sum(xb, deq), what="per-row W8A16")
    # against the reference's fake-per-row forward: same tolerance the reference's self-test uses,
    # tightened 4x (max|d| / max|ref| < 0.02; reference asserts 0.08, quantization.py:549-553)
    ref = O.bf16_to_f32(golden[f"{case}/{dt}/fake_row/y"])
    got = O.bf16_to_f32(bits16(y))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 0.02


# ----------------------------------------------------------------------------- W8A8 GEMMs
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(1, 64, 128), (3, 96, 256), (16, 48, 512), (130, 256, 1024), (256, 160, 3072),
                                 (77, 1024, 2048)])
def test_w8a8_vs_oracle(NQ, dt, mnk):
    M, N, K = mnk
    rng = np.random.default_rng(M * 7 + N)
    xb, wb, bb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05), rand_bf16(rng, (N,))
    qw, sw = NQ.per_row_quant(bf16_t(wb), TDT[dt])
    qwo, swo = O.per_row_quant(wb, dt)
    for bias in (None, bb):
        y = NQ.per_row_matmul(bf16_t(xb), qw, sw, None if bias is None else bf16_t(bias), TDT[dt])
        yo = O.per_row_matmul(xb, qwo, swo, bias, dt)
        if dt == "int8":
            assert np.array_equal(bits16(y), yo), "int8 W8A8 must be bit-exact (exact int32 accumulation)"
        else:
            ref, l1 = fp8_w8a8_exact(xb, qwo, swo, bias)
            assert_gemm_close(bits16(y), ref, l1, what="fp8 W8A8")     # every element


TRITON_CASES = ["m1_n64_k128", "m3_n64_k256", "m16_n48_k512", "m130_n64_k256", "m5_n8_k3072", "m130_n200_k384",
                "m7_n130_k128"]


@pytest.mark.parametrize("case", TRITON_CASES)
def test_per_row_path_vs_reference_triton_vectors(NQ, triton_golden, case):
    """The HIP per-row path against outputs of the reference's own Triton kernels (CPU interpreter).  The
    interpreter's narrowing casts differ from compiled Triton's round-to-nearest-even (tests/util.py), so:
    scales bit-exact; fp8 codes bit-exact wherever the interpreter's cast of the same f32 value agrees with RTNE;
    the int8 GEMM equal to the truncated vector or its successor in magnitude (= RTNE of the same f32 value, which
    the oracle test pins bit-exactly); the fp8 GEMM within one bf16 ulp + the f32-accumulation allowance."""
    from tests.util import interp_f32_to_e4m3
    g = triton_golden
    fp8 = torch.float8_e4m3fn
    for side in ("w", "x"):
        q, s = NQ.per_row_quant(bf16_t(g[f"{case}/{side}"]), fp8)
        assert np.array_equal(s.cpu().numpy(), g[f"{case}/fp8/s{side}"])
        yo, _ = O.per_row_quant(g[f"{case}/{side}"], "fp8", return_scaled=True)
        ib, valid = interp_f32_to_e4m3(yo)
        agree = valid & (ib == O.f32_to_fp8(yo))
        assert agree.mean() > 0.9
        assert np.array_equal(bits8(q)[agree], g[f"{case}/fp8/q{side}"][agree])
    # int8 GEMM on the stored operands
    qx, sx, qw, sw = (g[f"{case}/int8/{k}"] for k in ("qx", "sx", "qw", "sw"))
    y = NQ.w8a8_matmul(q_t(qx, "int8"), torch.from_numpy(sx).cuda(), q_t(qw, "int8"), torch.from_numpy(sw).cuda(),
                       None, torch.int8)
    d = bits16(y).astype(np.int32) - g[f"{case}/int8/y"].astype(np.int32)
    assert ((d == 0) | (d == 1)).all()
    assert np.array_equal(bits16(y), O.w8a8_gemm(qx, sx, qw, sw, None, "int8"))
    # fp8 GEMM on the stored operands (the interpreter's codes are valid e4m3 bytes like any other)
    qx, sx, qw, sw = (g[f"{case}/fp8/{k}"] for k in ("qx", "sx", "qw", "sw"))
    y = NQ.w8a8_matmul(q_t(qx, "fp8"), torch.from_numpy(sx).cuda(), q_t(qw, "fp8"), torch.from_numpy(sw).cuda(),
                       None, fp8)
    sc = sx[:, None].astype(np.float64) * sw[None, :].astype(np.float64)
    ref = O.w8a8_gemm(qx, sx, qw, sw, None, "fp8", return_acc=True) * sc
    l1 = (np.abs(O.fp8_to_f32(qx)).astype(np.float64) @ np.abs(O.fp8_to_f32(qw)).astype(np.float64).T) * sc
    assert_gemm_close(bits16(y), ref, l1, what="fp8 W8A8 vs exact")
    gold = O.bf16_to_f32(g[f"{case}/fp8/y"]).astype(np.float64)
    got = O.bf16_to_f32(bits16(y)).astype(np.float64)
    assert (np.abs(got - gold) <= 2.0 ** -7 * np.abs(ref) + 2.0 ** -17 * l1 + 1e-30).all()


def test_w8a8_reference_selftest_tolerance(NQ):
    """The reference's own acceptance test (quantization.py:527-555): randn operands,
    max|y - (x@w.T+b)| / max|.| < 0.08 for both dtypes."""
    torch.manual_seed(0)
    for (M, N, K) in [(1, 1536, 1536), (10, 1536, 1536), (512, 4096, 4096)]:
        for dt in (torch.int8, torch.float8_e4m3fn):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
            b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
            qw, sw = NQ.per_row_quant(w, dt)
            y = NQ.per_row_matmul(x, qw, sw, b, dt)
            y_ref = x.float() @ w.float().T + b.float()[None, :]
            err = (y.float() - y_ref).abs().max() / y_ref.abs().max()
            assert err < 0.08, (M, N, K, dt, float(err))


# ----------------------------------------------------------------------------- every tile / split
TILES = [(1, 1, 4, 2), (1, 1, 4, 4), (1, 1, 4, 8), (2, 1, 4, 2), (2, 1, 4, 4), (1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4),
         (8, 4, 1, 4), (2, 2, 2, 2), (2, 2, 2, 4), (2, 2, 2, 8), (4, 2, 2, 2), (4, 2, 2, 4),
         (1, 2, 4, 8), (1, 2, 4, 4), (2, 2, 4, 8), (2, 2, 4, 4), (1, 1, 8, 4), (2, 4, 2, 4), (4, 4, 2, 4),
         (1, 4, 1, 32), (1, 4, 2, 16), (2, 4, 2, 16), (1, 2, 4, 16), (1, 1, 8, 8), (4, 8, 1, 2), (4, 8, 1, 4)]


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("split", [1, 3])
def test_every_tile_configuration(NQ, tile, split):
    """Force each built tile (and a split-K that leaves a ragged last slice) on a shape with M, N
    tails; all four arithmetic variants must still match the oracle."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, N, K = 70, 200, 1184          # K = 9.25 * 128: ragged chunks and slices
    rng = np.random.default_rng(11)
    xb, wb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05)
    x = bf16_t(xb)
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
        for dt in DTYPES:
            qo, so = O.per_row_quant(wb, dt)
            qw, sw = q_t(qo, dt), torch.from_numpy(so).cuda()
            y = NQ.per_row_weight_only_matmul(x, qw, sw, None, TDT[dt])
            assert_gemm_close(bits16(y), O.w8a16_rowscale_matmul(xb, qo, so, None, dt, return_f64=True),
                              O.abs_product_sum(xb, O._decode_q(qo, dt) * so[:, None]), what=f"w8a16 {dt} {tile}")
            y8 = NQ.per_row_matmul(x, qw, sw, None, TDT[dt])
            yo = O.per_row_matmul(xb, qo, so, None, dt)
            if dt == "int8":
                assert np.array_equal(bits16(y8), yo), f"w8a8 int8 {tile} split {split}"
            else:
                ref8, l18 = fp8_w8a8_exact(xb, qo, so)
                assert_gemm_close(bits16(y8), ref8, l18, what=f"w8a8 fp8 {tile} split {split}")   # every element
            if tile[0] <= 4:
                g = 32 if dt == "int8" else 64
                qg, sg = O.per_group_quant(wb[:, :1152], g, dt)
                yg = NQ.per_group_matmul(bf16_t(xb[:, :1152]), q_t(qg, dt), torch.from_numpy(sg).cuda(), None, g, TDT[dt])
                deq = (O._decode_q(qg, dt).reshape(N, -1, g) * sg[..., None]).reshape(N, -1)
                assert_gemm_close(bits16(yg), O.per_group_matmul(xb[:, :1152], qg, sg, None, g, dt, return_f64=True),
                                  O.abs_product_sum(xb[:, :1152], deq), what=f"group {dt} {tile}")
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


def test_tile_weight_layout_and_round_trip(NQ):
    """tile_weight against the layout formula of include/qlinear.h (N not a multiple of 32: zero padding) and back."""
    N, K = 70, 96
    q = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
    t = NQ.tile_weight(q)
    assert t.shape == (N, K) and t.data.shape == (96, K)
    flat = t.data.view(-1).cpu().numpy()
    qc = q.cpu().numpy()
    n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
    idx = ((n // 32) * (K // 32) + k // 32) * 1024 + ((k % 32) // 16 * 32 + n % 32) * 16 + k % 16
    assert np.array_equal(flat[idx], qc)
    mask = np.ones(flat.size, bool); mask[idx.ravel()] = False
    assert (flat[mask] == 0).all()
    assert torch.equal(t.to_rowmajor(), q)


@pytest.mark.parametrize("tile", TILES)
def test_tiled_weights_every_tile(NQ, tile):
    """The tiled weight layout must give the same results as the row-major one: the int8 W8A8 GEMM and every W8A16
    variant bit for bit (same tile, same summation order), with M, N and K tails and a ragged split-K."""
    from nanovllm import _qlinear as Q, ops
    lib = Q.load()
    M, N, K = 70, 200, 1184
    rng = np.random.default_rng(17)
    x = bf16_t(rand_bf16(rng, (M, K)))
    w = bf16_t(rand_bf16(rng, (N, K), 0.05))
    bias = bf16_t(rand_bf16(rng, (N,), 0.5))
    try:
        for split in (1, 3):
            assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
            for tdt in (torch.int8, torch.float8_e4m3fn):
                qw, sw = NQ.per_row_quant(w, tdt)
                tw = NQ.tile_weight(qw)
                for fn in (NQ.per_row_weight_only_matmul, NQ.per_row_matmul):
                    assert torch.equal(fn(x, tw, sw, bias, tdt), fn(x, qw, sw, bias, tdt)), (fn.__name__, tdt, tile, split)
                if tile[0] <= 4:
                    qg, sg = NQ.per_group_quant(w[:, :1152].contiguous(), 128, tdt)
                    xg = x[:, :1152].contiguous()
                    assert torch.equal(NQ.per_group_matmul(xg, NQ.tile_weight(qg), sg, None, 128, tdt),
                                       NQ.per_group_matmul(xg, qg, sg, None, 128, tdt))
        if tile[1] % 2 == 0:
            assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], 1) == 0
            w2 = bf16_t(rand_bf16(rng, (2 * 160, K), 0.05))
            qw, sw = NQ.per_row_quant(w2, torch.int8)
            il = NQ.interleave_gate_up
            assert torch.equal(NQ.per_row_weight_only_matmul(x, NQ.tile_weight(il(qw)), il(sw), None, torch.int8, silu=True),
                               ops.silu_mul(NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)))
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


# ----------------------------------------------------------------------------- fused SiLU * mul epilogue
@pytest.mark.parametrize("tile", [t for t in TILES if t[1] % 2 == 0])
@pytest.mark.parametrize("split", [1, 3])
def test_fused_silu_epilogue_bit_identical(NQ, tile, split):
    """gate/up GEMM with the SiLU*mul epilogue on the strip-interleaved weight == the plain GEMM on the
    packed weight followed by the stand-alone SiLU*mul operator, bit for bit, for every arithmetic
    variant and every tile that pairs column strips (M and N tails included); with K split across workgroups the
    tile's last workgroup runs the gate/up exchange on the reduced tile (same slices on both routes)."""
    from nanovllm import _qlinear as Q, ops
    lib = Q.load()
    Q.workspace.reserve("cuda", 64 << 20)
    M, I, K = 70, 160, 1152
    rng = np.random.default_rng(5)
    xb, wb = rand_bf16(rng, (M, K)), rand_bf16(rng, (2 * I, K), 0.05)
    x, w = bf16_t(xb), bf16_t(wb)
    bias = bf16_t(rand_bf16(rng, (2 * I,), 0.5))
    il = NQ.interleave_gate_up
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
        for tdt in (torch.int8, torch.float8_e4m3fn):
            for b in (None, bias):
                bi = None if b is None else il(b)
                qw, sw = NQ.per_row_quant(w, tdt)
                for fn in (NQ.per_row_weight_only_matmul, NQ.per_row_matmul):
                    ref = ops.silu_mul(fn(x, qw, sw, b, tdt))
                    got = fn(x, il(qw), il(sw), bi, tdt, silu=True)
                    assert got.shape == (M, I) and torch.equal(got, ref), (fn.__name__, tdt, tile)
                qt, st = NQ.per_tensor_quant(w, tdt)
                st = st.float().reshape(1)
                assert torch.equal(NQ.per_tensor_matmul(x, il(qt), st, bi, tdt, silu=True),
                                   ops.silu_mul(NQ.per_tensor_matmul(x, qt, st, b, tdt)))
                if tile[0] <= 4:
                    qg, sg = NQ.per_group_quant(w, 128, tdt)
                    assert torch.equal(NQ.per_group_matmul(x, il(qg), il(sg), bi, 128, tdt, silu=True),
                                       ops.silu_mul(NQ.per_group_matmul(x, qg, sg, b, 128, tdt)))
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


def _rope_setup(M, Hq, Hkv, K, seed, slots_free=4096):
    from nanovllm.models.qwen3 import build_cos_sin
    g = torch.Generator(device="cuda").manual_seed(seed)
    N = (Hq + 2 * Hkv) * 128
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = (torch.randn(N, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    qn = (1 + 0.2 * torch.randn(128, device="cuda", generator=g)).to(torch.bfloat16)
    kn = (1 + 0.2 * torch.randn(128, device="cuda", generator=g)).to(torch.bfloat16)
    cs = build_cos_sin(128, 4096, 1e6, "cuda")
    pos = torch.randint(0, 4096, (M,), device="cuda", generator=g)
    slots = torch.randperm(slots_free, device="cuda", generator=g)[:M].to(torch.int32)
    if M > 3:
        slots[3] = -1                                   # a padded row: nothing is stored for it
    return x, w, bias, qn, kn, cs, pos, slots


def _rope_both_ways(NQ, mod, x, qn, kn, cs, pos, slots, Hq, Hkv, want_kv, quantized=None):
    """(fused, unfused) results: q, k, v and the two caches."""
    from nanovllm import ops
    outs = []
    for fused in (True, False):
        kc = torch.zeros(16, 256, Hkv, 128, device="cuda", dtype=torch.bfloat16)
        vc = torch.zeros_like(kc)
        if fused:
            q, k, v = NQ.qkv_rope_matmul(mod, x, pos, cs, qn, kn, 1e-6, Hq, Hkv, 128, kc, vc, slots, want_kv, quantized)
        else:
            qkv = mod(x) if quantized is None else mod.forward_quantized(*quantized)
            q, k, v = ops.qknorm_rope_store(qkv, pos, cs, qn, kn, 1e-6, Hq, Hkv, 128, kc, vc, slots, want_kv)
        torch.cuda.synchronize()
        outs.append((q, k, v, kc, vc))
    return outs


ROPE_TILES = [(1, 4, 2, 16), (1, 4, 1, 4), (2, 4, 2, 4), (2, 4, 1, 4), (4, 8, 1, 2), (4, 8, 1, 4)]    # QL_TILE_LIST_ROPE


@pytest.mark.parametrize("tile", ROPE_TILES)
@pytest.mark.parametrize("split", [1, 3])
def test_qkv_rope_epilogue_bit_identical(NQ, tile, split):
    """qkv GEMM with the q/k norm + RoPE + KV-cache store in its epilogue == the plain GEMM (same tile, same split)
    followed by the stand-alone operator, bit for bit: q, k, v and both caches; every arithmetic variant, every tile
    whose columns are whole heads, M tails, a padded (-1) slot, with and without the norm weights and a bias."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, Hq, Hkv, K = 70, 2, 1, 1184
    x, w, bias, qn, kn, cs, pos, slots = _rope_setup(M, Hq, Hkv, K, 3)
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
        mods = [NQ.QLinear(torch.int8, w, None, weight_only=True), NQ.QLinear(torch.float8_e4m3fn, w, bias, weight_only=True),
                NQ.QLinear(torch.int8, w, bias), NQ.QLinear(torch.float8_e4m3fn, w, None), NQ.QTensorLinear(torch.int8, w, None)]
        if tile[0] <= 4:
            mods.append(NQ.QGroupLinear(torch.int8, w[:, :1152].contiguous(), None, 128))
        for mi, mod in enumerate(mods):
            mod.tile()
            xx = x if mod.qw.shape[1] == K else x[:, :1152].contiguous()
            for norms in ((qn, kn), (None, None)):
                for want_kv in (True, False):
                    a, b = _rope_both_ways(NQ, mod, xx, norms[0], norms[1], cs, pos, slots, Hq, Hkv, want_kv)
                    for ta, tb, nm in zip(a, b, ("q", "k", "v", "k_cache", "v_cache")):
                        assert (ta is None) == (tb is None)
                        if ta is not None:
                            assert torch.equal(ta, tb), (mi, nm, tile, split, want_kv)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("tile", [(4, 8, 1, 2), (4, 8, 1, 4)])
def test_qkv_rope_epilogue_odd_head_count(NQ, tile):
    """An odd number of heads (Hq = 5, Hkv = 1: a Qwen3-14B rank under TP = 8) on the 256-column tiles, whose last
    tile holds ONE head: the tile's second, non-existent head must be skipped (it used to be stored as a V head one
    past the end, i.e. over the next token's V head 0 in v_out and in the cache).  Bit for bit against the unfused pair."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, Hq, Hkv, K = 150, 5, 1, 512
    x, w, bias, qn, kn, cs, pos, slots = _rope_setup(M, Hq, Hkv, K, 9)
    slots = torch.arange(M, device="cuda", dtype=torch.int32)          # adjacent slots: an overrun lands on a checked row
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], 1) == 0
        for mod in (NQ.QLinear(torch.int8, w, None, weight_only=True), NQ.QLinear(torch.int8, w, bias)):
            mod.tile()
            for _ in range(3):                                          # (the overrun raced with the legitimate store)
                a, b = _rope_both_ways(NQ, mod, x, qn, kn, cs, pos, slots, Hq, Hkv, True)
                for ta, tb, nm in zip(a, b, ("q", "k", "v", "k_cache", "v_cache")):
                    assert torch.equal(ta, tb), (nm, tile)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("M", [1, 16, 33, 130, 256, 700, 2048])
def test_qkv_rope_full_size_own_tiles(NQ, M):
    """Qwen3-0.6B qkv shape with the library's own tile choice for each route.  int8 W8A8 accumulates exactly, so the
    fused route must equal the unfused one bit for bit whatever tiles the two pick (also with the activations
    quantized by the preceding norm); the float variants are compared at the GEMM's rounding level."""
    from nanovllm import ops
    Hq, Hkv, K = 16, 8, 1024
    x, w, bias, qn, kn, cs, pos, slots = _rope_setup(M, Hq, Hkv, K, M)
    mod = NQ.QLinear(torch.int8, w, None).tile()
    a, b = _rope_both_ways(NQ, mod, x, qn, kn, cs, pos, slots, Hq, Hkv, M > 256)
    for ta, tb in zip(a, b):
        assert (ta is None) == (tb is None) and (ta is None or torch.equal(ta, tb))
    qx, sx = NQ.per_row_quant(x, torch.int8)
    a, b = _rope_both_ways(NQ, mod, None, qn, kn, cs, pos, slots, Hq, Hkv, False, quantized=(qx, sx))
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    # positions=None: the table rows of the tokens handed in already gathered (fused and stand-alone operator)
    rows = cs.index_select(0, pos)
    kc = torch.zeros(16, 256, Hkv, 128, device="cuda", dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    q2, _, _ = NQ.qkv_rope_matmul(mod, x, None, rows, qn, kn, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)
    assert torch.equal(q2, b[0]) and torch.equal(kc, b[3]) and torch.equal(vc, b[4])
    kc.zero_(); vc.zero_()
    q3, _, _ = ops.qknorm_rope_store(mod(x), None, rows, qn, kn, 1e-6, Hq, Hkv, 128, kc, vc, slots, False)
    assert torch.equal(q3, b[0]) and torch.equal(kc, b[3])
    for mk in (lambda: NQ.QLinear(torch.int8, w, None, weight_only=True), lambda: NQ.QGroupLinear(torch.float8_e4m3fn, w, None, 128)):
        mod = mk().tile()
        a, b = _rope_both_ways(NQ, mod, x, qn, kn, cs, pos, slots, Hq, Hkv, False)
        for ta, tb in ((a[0], b[0]), (a[3], b[3]), (a[4], b[4])):
            d = (ta.float() - tb.float()).abs()
            assert (d <= 2.0 ** -5 * tb.float().abs() + 3e-2).all()
            assert (ta == tb).float().mean() > 0.9


# ----------------------------------------------------------------------------- stream-K schedule
SK_TILES = [(1, 1, 4, 4), (1, 2, 4, 4), (2, 2, 4, 4), (2, 2, 2, 4), (1, 4, 1, 4), (2, 4, 1, 4), (4, 4, 1, 4), (2, 4, 2, 4),
            (4, 4, 2, 4), (4, 8, 1, 2), (4, 8, 1, 4), (1, 1, 8, 4), (1, 2, 4, 8), (1, 4, 2, 16), (8, 4, 1, 4)]   # QL_TILE_LIST_SK*
SK_ROPE_TILES = [(1, 4, 1, 4), (2, 4, 2, 4), (2, 4, 1, 4), (4, 8, 1, 2)]                                       # QL_TILE_LIST_SK_ROPE


@pytest.mark.parametrize("tile", SK_TILES)
@pytest.mark.parametrize("G", [8, 24, 48, 208])
def test_stream_k_schedule_every_tile(NQ, tile, G):
    """The stream-K schedule (a 1-D grid of G workgroups sharing the (tile, chunk) units, forced with split = -G) on
    every tile built for it: M, N tails, a ragged last chunk, workgroup ranges that start and end inside tiles, grids
    with more workgroups than units (G = 208 on the large tiles) and with the row tiles walked side by side (G a
    multiple of 8 x row tiles).  Per-row int8 / fp8 and per-group against the oracle, run-to-run bit-identical."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    Q.workspace.reserve("cuda", 128 << 20)
    M, N, K = 150, 200, 1184
    rng = np.random.default_rng(31)
    xb, wb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05)
    x = bf16_t(xb)
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], -G) == 0
        plan = Q.plan_query(0, M, N, K)
        kb = 32 * tile[3] * tile[2]
        units = -(-M // (32 * tile[0])) * -(-N // (32 * tile[1])) * -(-K // kb)
        assert plan == (tile[0], tile[1], tile[2], tile[3], -min(G, units)), plan     # (never more workgroups than units)
        for dt in DTYPES:
            qo, so = O.per_row_quant(wb, dt)
            qw, sw = q_t(qo, dt), torch.from_numpy(so).cuda()
            y = NQ.per_row_weight_only_matmul(x, qw, sw, None, TDT[dt])
            assert_gemm_close(bits16(y), O.w8a16_rowscale_matmul(xb, qo, so, None, dt, return_f64=True),
                              O.abs_product_sum(xb, O._decode_q(qo, dt) * so[:, None]), what=f"stream-K w8a16 {dt} {tile} G={G}")
            for _ in range(2):
                assert torch.equal(NQ.per_row_weight_only_matmul(x, NQ.tile_weight(qw), sw, None, TDT[dt]), y)
            if tile[0] <= 4:
                g = 32 if dt == "int8" else 128
                qg, sg = O.per_group_quant(wb[:, :1152], g, dt)
                yg = NQ.per_group_matmul(bf16_t(xb[:, :1152]), q_t(qg, dt), torch.from_numpy(sg).cuda(), None, g, TDT[dt])
                deq = (O._decode_q(qg, dt).reshape(N, -1, g) * sg[..., None]).reshape(N, -1)
                assert_gemm_close(bits16(yg), O.per_group_matmul(xb[:, :1152], qg, sg, None, g, dt, return_f64=True),
                                  O.abs_product_sum(xb[:, :1152], deq), what=f"stream-K group {dt} {tile} G={G}")
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("tile", [t for t in SK_TILES if t[1] % 2 == 0])
@pytest.mark.parametrize("G", [16, 40])
def test_stream_k_silu_epilogue_bit_identical(NQ, tile, G):
    """SiLU*mul epilogue under the stream-K schedule == plain GEMM (same tile, same schedule) + ql_silu_mul, bit for bit.
    Under this schedule the places where a tile's K range is cut between workgroups depend on the tile's position, so
    the f32 summation order of a column depends on the column tile it sits in: the unfused route runs on the SAME
    strip-interleaved weight and its output columns are put back in [gate | up] order before ql_silu_mul."""
    from nanovllm import _qlinear as Q, ops
    lib = Q.load()
    Q.workspace.reserve("cuda", 128 << 20)
    M, I, K = 150, 160, 1152
    rng = np.random.default_rng(6)
    x, w = bf16_t(rand_bf16(rng, (M, K))), bf16_t(rand_bf16(rng, (2 * I, K), 0.05))
    bias = bf16_t(rand_bf16(rng, (2 * I,), 0.5))
    il = NQ.interleave_gate_up
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], -G) == 0
        perm = il(torch.arange(2 * I, device="cuda"))          # interleaved position -> packed row

        def unfused(y_il):
            y = torch.empty_like(y_il)
            y[:, perm] = y_il
            return ops.silu_mul(y)
        for tdt in (torch.int8, torch.float8_e4m3fn):
            qw, sw = NQ.per_row_quant(w, tdt)
            for b in (None, bias):
                bi = None if b is None else il(b)
                ref = unfused(NQ.per_row_weight_only_matmul(x, il(qw), il(sw), bi, tdt))
                got = NQ.per_row_weight_only_matmul(x, il(qw), il(sw), bi, tdt, silu=True)
                assert torch.equal(got, ref), (tdt, tile, G)
                plain = ops.silu_mul(NQ.per_row_weight_only_matmul(x, qw, sw, b, tdt))       # (packed weight: other cuts)
                assert (got == plain).float().mean() > 0.99
            if tile[0] <= 4:
                qg, sg = NQ.per_group_quant(w, 128, tdt)
                assert torch.equal(NQ.per_group_matmul(x, il(qg), il(sg), None, 128, tdt, silu=True),
                                   unfused(NQ.per_group_matmul(x, il(qg), il(sg), None, 128, tdt)))
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("tile", SK_ROPE_TILES)
@pytest.mark.parametrize("G", [16, 40])
def test_stream_k_qkv_rope_epilogue_bit_identical(NQ, tile, G):
    """qkv epilogue under the stream-K schedule == plain GEMM (same tile, same schedule) + ql_qknorm_rope_store."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    Q.workspace.reserve("cuda", 128 << 20)
    M, Hq, Hkv, K = 150, 3, 1, 1184
    x, w, bias, qn, kn, cs, pos, slots = _rope_setup(M, Hq, Hkv, K, 13)
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], -G) == 0
        mods = [NQ.QLinear(torch.int8, w, bias, weight_only=True), NQ.QLinear(torch.float8_e4m3fn, w, None, weight_only=True)]
        if tile[0] <= 4:
            mods.append(NQ.QGroupLinear(torch.int8, w[:, :1152].contiguous(), None, 128))
        for mi, mod in enumerate(mods):
            mod.tile()
            xx = x if mod.qw.shape[1] == K else x[:, :1152].contiguous()
            a, b = _rope_both_ways(NQ, mod, xx, qn, kn, cs, pos, slots, Hq, Hkv, True)
            for ta, tb, nm in zip(a, b, ("q", "k", "v", "k_cache", "v_cache")):
                assert torch.equal(ta, tb), (mi, nm, tile, G)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("shape", [("4b gate_up", 19456, 2560), ("4b down", 2560, 9728), ("4b qkv", 6144, 2560)])
@pytest.mark.parametrize("M", [48, 256, 300])
def test_stream_k_full_size_sampled_rows(NQ, shape, M):
    """The schedule at the sizes it is built for (Qwen3-4B layer shapes, 256 and 512 workgroups, g128): sampled output
    rows against the exactly evaluated formula, determinism, and a tile-aligned grid (every workgroup whole tiles)."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    _, N, K = shape
    Q.workspace.reserve("cuda", 512 << 20)
    g = torch.Generator(device="cuda").manual_seed(N + M)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    qg, sg = NQ.per_group_quant(w, 128, torch.int8)
    tw = NQ.tile_weight(qg)
    rows = [0, M // 2, M - 1]
    xb = bits16(x[rows])
    deq = (qg.cpu().numpy().astype(np.float32).reshape(N, -1, 128) * sg.cpu().numpy()[..., None]).reshape(N, K)
    ref = O.per_group_matmul(xb, qg.cpu().numpy(), sg.cpu().numpy(), None, 128, "int8", return_f64=True)
    l1 = O.abs_product_sum(xb, deq)
    try:
        for tile, G in (((4, 8, 1, 2), 256), ((2, 4, 2, 4), 512), ((1, 2, 4, 4), 512), ((4, 4, 2, 4), 304)):
            assert lib.ql_gemm_force_config(*tile, -G) == 0
            y = NQ.per_group_matmul(x, tw, sg, None, 128, torch.int8)
            assert_gemm_close(bits16(y[rows]), ref, l1, what=f"stream-K {shape[0]} M={M} {tile} G={G}")
            assert torch.equal(NQ.per_group_matmul(x, tw, sg, None, 128, torch.int8), y)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("tile", [(1, 1, 8, 4), (1, 1, 4, 4), (1, 1, 4, 2), (1, 2, 4, 8), (2, 2, 4, 8), (2, 4, 2, 4)])
@pytest.mark.parametrize("split", [1, 2, 5])
def test_deferred_splitk_norm_bit_identical(NQ, tile, split):
    """GEMM with the split-K reduction left to the residual add + RMSNorm launch == the self-contained GEMM (same
    tile, same slices) followed by ql_add_rmsnorm, bit for bit (normed output and new residual): per-row int8 / fp8,
    per-tensor, per-group, with and without a bias, M and N tails, a ragged last slice."""
    from nanovllm import _qlinear as Q, ops
    lib = Q.load()
    M, N, K = 70, 200, 1184
    rng = np.random.default_rng(21)
    x = bf16_t(rand_bf16(rng, (M, K)))
    w = bf16_t(rand_bf16(rng, (N, K), 0.05))
    bias = torch.nn.Parameter(bf16_t(rand_bf16(rng, (N,), 0.5)), requires_grad=False)
    res = bf16_t(rand_bf16(rng, (M, N)))
    nw = bf16_t(O.f32_to_bf16((1 + 0.1 * rng.standard_normal(N)).astype(np.float32)))
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
        mods = [NQ.QLinear(torch.int8, w, None, weight_only=True), NQ.QLinear(torch.float8_e4m3fn, w, bias, weight_only=True),
                NQ.QTensorLinear(torch.int8, w, bias), NQ.QGroupLinear(torch.float8_e4m3fn, w[:, :1152].contiguous(), None, 128),
                NQ.QGroupLinear(torch.int8, w[:, :1152].contiguous(), bias, 384)]
        for mi, mod in enumerate(mods):
            mod.tile()
            xx = x if mod.qw.shape[1] == K else x[:, :1152].contiguous()
            ps = NQ.w8a16_partial_matmul(mod, xx)
            assert ps.parts.shape[1:] == (M, N) and ps.parts.shape[0] == lib.ql_gemm_partial_slices(M, N, xx.shape[1], ps.granularity)
            out, new_res = ops.add_rmsnorm(ps, res, nw, 1e-6)
            ref_out, ref_res = ops.add_rmsnorm(mod(xx), res, nw, 1e-6)
            assert torch.equal(new_res, ref_res) and torch.equal(out, ref_out), (mi, tile, split)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


def test_weight_only_partial_sum_into_a_quantizing_norm(NQ):
    """A weight-only o_proj / down_proj in front of a W8A8 linear (mixed per-module quantization): the norm that
    finishes the deferred GEMM must also hand out the quantized activations, equal to norm -> per_row_quant."""
    from nanovllm import ops
    M, N, K = 48, 256, 3072
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16, generator=g)
    nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
    mod = NQ.QLinear(torch.int8, w, None, weight_only=True).tile()
    ps = NQ.w8a16_partial_matmul(mod, x)
    out, new_res = ops.add_rmsnorm(ps, res, nw, 1e-6)
    for dt in (torch.int8, torch.float8_e4m3fn):
        q, s, r2 = ops.add_rmsnorm_quant(ps, res, nw, 1e-6, dt)
        q_ref, s_ref = NQ.per_row_quant(out, dt)
        assert torch.equal(r2, new_res) and torch.equal(s, s_ref) and torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("nk", [(1024, 2048), (1024, 3072), (2560, 9728), (4096, 1536)])
@pytest.mark.parametrize("M", [1, 16, 130, 256, 512])
def test_deferred_splitk_full_size_own_plan(NQ, nk, M):
    """o_proj / down_proj shapes with the library's own plans on both routes (they slice K differently, so the f32
    sums differ in order): the normed output of the deferred route must agree with the self-contained route to the
    GEMM's rounding level, and the module-level switch must route decode sizes through it."""
    from nanovllm import ops
    N, K = nk
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16, generator=g)
    nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
    mod = NQ.QLinear(torch.int8, w, None, weight_only=True).tile()
    ps = mod.forward_partial(x)
    if K >= 3072 and M <= 128:
        assert isinstance(ps, NQ.PartialSum) and ps.parts.shape[0] >= 2
    if torch.is_tensor(ps):                          # the plan has a single slice: nothing was deferred
        ps = NQ.w8a16_partial_matmul(mod, x)
    out, new_res = ops.add_rmsnorm(ps, res, nw, 1e-6)
    ref_out, ref_res = ops.add_rmsnorm(mod(x), res, nw, 1e-6)
    for a, b in ((new_res, ref_res), (out, ref_out)):
        d = (a.float() - b.float()).abs()
        assert (d <= 2.0 ** -6 * b.float().abs() + 2e-2).all()
        assert (a == b).float().mean() > 0.9
    w8a8 = NQ.QLinear(torch.int8, w, None).tile().forward_partial(x)          # W8A8 defers too (its own test below)
    assert torch.is_tensor(w8a8) or w8a8.sx is not None
    big = torch.randn(mod.PARTIAL_MAX_ROWS + 1, K, device="cuda", dtype=torch.bfloat16)
    assert torch.is_tensor(mod.forward_partial(big))


@pytest.mark.parametrize("M", [1, 16, 33, 130, 256, 2048])
def test_fused_silu_full_size_and_module(NQ, M):
    """Qwen3-0.6B gate/up shape through the module API with the library's own tile choice: int8 W8A8 is
    exact whatever the tile, so it must equal the unfused route bit for bit; the float variants are
    compared at the GEMM tolerance (the two routes may pick different tiles, i.e. summation orders)."""
    from nanovllm import ops
    from nanovllm.layers.linear import MergedColumnParallelLinear
    lin = MergedColumnParallelLinear(1024, [3072, 3072], bias=False).cuda()
    g = torch.Generator(device="cuda").manual_seed(M)
    lin.weight.data = (torch.randn(6144, 1024, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    x = torch.randn(M, 1024, device="cuda", dtype=torch.bfloat16, generator=g)
    plain, fused = NQ.QLinear.from_linear(lin, torch.int8), NQ.QLinear.from_linear(lin, torch.int8).enable_silu_mul()
    assert torch.equal(fused(x), ops.silu_mul(plain(x)))
    for mk in (lambda: NQ.QLinear.from_linear(lin, torch.int8, weight_only=True),
               lambda: NQ.QGroupLinear.from_linear(lin, torch.float8_e4m3fn, 128)):
        a, b = ops.silu_mul(mk()(x)).float(), mk().enable_silu_mul()(x).float()
        assert b.shape == (M, 3072)
        assert ((a - b).abs() <= 2.0 ** -6 * a.abs() + 2e-3).all()
    with pytest.raises(AssertionError):
        NQ.interleave_gate_up(torch.zeros(2 * 48, 8))            # halves must be multiples of 32 rows


@pytest.mark.parametrize("g", [32, 64, 128, 256, 512])
def test_group_sizes(NQ, g):
    M, N, K = 40, 96, 1024
    rng = np.random.default_rng(g)
    xb, wb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05)
    for dt in DTYPES:
        qg, sg = O.per_group_quant(wb, g, dt)
        y = NQ.per_group_matmul(bf16_t(xb), q_t(qg, dt), torch.from_numpy(sg).cuda(), None, g, TDT[dt])
        deq = (O._decode_q(qg, dt).reshape(N, -1, g) * sg[..., None]).reshape(N, -1)
        assert_gemm_close(bits16(y), O.per_group_matmul(xb, qg, sg, None, g, dt, return_f64=True),
                          O.abs_product_sum(xb, deq), what=f"g{g} {dt}")


@pytest.mark.parametrize("g", [96, 384, 1152])
@pytest.mark.parametrize("tile", [None, (1, 1, 8, 8), (2, 2, 4, 8), (1, 4, 2, 16), (2, 4, 2, 4), (4, 8, 1, 4)])
def test_group_sizes_unrelated_to_the_chunk_length(NQ, g, tile):
    """Group sizes that neither divide nor are divided by a tile's k chunk (the reference takes any divisor of K,
    quantization.py:205): quantizer and fake quantizer bit-exact, GEMM at the usual bar, on tiles whose wave
    chunk is 128, 256 and 512 k."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, N, K = 40, 96, 1152
    rng = np.random.default_rng(g)
    xb, wb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05)
    wb[3, :g] = 0
    try:
        if tile is not None:
            assert lib.ql_gemm_force_config(*tile, 1) == 0
        for dt in DTYPES:
            qo, so = O.per_group_quant(wb, g, dt)
            q, s = NQ.per_group_quant(bf16_t(wb), g, TDT[dt])
            assert np.array_equal(bits8(q), qo.view(np.uint8)) and np.array_equal(s.cpu().numpy(), so)
            assert np.array_equal(bits16(NQ.fake_per_group_quant(bf16_t(wb), g, TDT[dt])), O.fake_per_group_quant(wb, g, dt))
            y = NQ.per_group_matmul(bf16_t(xb), q, s, None, g, TDT[dt])
            deq = (O._decode_q(qo, dt).reshape(N, -1, g) * so[..., None]).reshape(N, -1)
            assert_gemm_close(bits16(y), O.per_group_matmul(xb, qo, so, None, g, dt, return_f64=True),
                              O.abs_product_sum(xb, deq), what=f"g{g} {dt} {tile}")
            yt = NQ.per_group_matmul(bf16_t(xb), NQ.tile_weight(q), s, None, g, TDT[dt])
            assert torch.equal(yt, y)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("g", [1, 4, 12, 20, 8, 16, 24, 48, 80, 120, 240])
def test_group_sizes_that_are_not_multiples_of_32(NQ, g):
    """The reference takes any divisor of K as the group size (quantization.py:205); sizes that are not multiples of
    the MFMA kernels' 32-k step run the plain GEMM on the library's f32 kernel (and, when not multiples of 8, the
    quantizers on a thread-per-group kernel): quantizer and fake quantizer bit-exact, GEMM at the usual
    bar, ragged M and N, bias, row-major and tiled weights; the fused forms do not exist for them and say so."""
    from nanovllm import _qlinear as Q
    M, N, K = 37, 50, 960
    rng = np.random.default_rng(1000 + g)
    xb, wb, bb = rand_bf16(rng, (M, K)), rand_bf16(rng, (N, K), 0.05), rand_bf16(rng, (N,), 0.5)
    for dt in DTYPES:
        qo, so = O.per_group_quant(wb, g, dt)
        q, s = NQ.per_group_quant(bf16_t(wb), g, TDT[dt])
        assert np.array_equal(bits8(q), qo.view(np.uint8)) and np.array_equal(s.cpu().numpy(), so)
        assert np.array_equal(bits16(NQ.fake_per_group_quant(bf16_t(wb), g, TDT[dt])), O.fake_per_group_quant(wb, g, dt))
        deq = (O._decode_q(qo, dt).reshape(N, -1, g) * so[..., None]).reshape(N, -1)
        y = NQ.per_group_matmul(bf16_t(xb), q, s, None, g, TDT[dt])
        assert_gemm_close(bits16(y), O.per_group_matmul(xb, qo, so, None, g, dt, return_f64=True),
                          O.abs_product_sum(xb, deq), what=f"g{g} {dt}")
        assert torch.equal(NQ.per_group_matmul(bf16_t(xb), NQ.tile_weight(q), s, None, g, TDT[dt]), y)
        # bias: added to the ROUNDED sum (quantization.py:302-305), one more rounding
        yb = NQ.per_group_matmul(bf16_t(xb), q, s, bf16_t(bb), g, TDT[dt])
        assert torch.equal(yb, (y.float() + bf16_t(bb).float()[None, :]).to(torch.bfloat16))
        with pytest.raises(Q.QLinearError, match="multiple of 32"):
            q64, s64 = NQ.per_group_quant(bf16_t(rand_bf16(rng, (64, K), 0.05)), g, TDT[dt])
            NQ.per_group_matmul(bf16_t(xb), q64, s64, None, g, TDT[dt], silu=True)


def test_group_aligned_main_loop_bit_identical_to_generic(NQ, tmp_path):
    """g = 128 puts the per-group kernels on the form of their main loop in which group boundaries are compile-time
    positions of the unrolled steps (zero C operand on a group's first MFMA, no clearing of the group accumulator);
    a second process with QL_GROUP_GENERIC=1 runs the same cases on the generic form: identical bits, every tile
    family (double-buffered, two-chunk-deep, 64-k chunks in both halves of a group, single-buffer with staged
    scales), K slices, the stream-K schedule and the SiLU epilogue."""
    import os
    import subprocess
    import sys
    from tests.util import group_form_outputs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "generic.pt")
    code = ("import sys, torch; sys.path[:0] = [%r, %r]; from tests.util import group_form_outputs; "
            "torch.save(group_form_outputs(), %r)" % (root, os.path.join(root, "nano-vllm-hw3_amd"), out))
    env = dict(os.environ, QL_GROUP_GENERIC="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    generic = torch.load(out)
    fast = group_form_outputs()
    assert set(fast) == set(generic) and len(fast) >= 120
    for k in fast:
        if "(4, 8, 1, 2)" in k:
            # 64-k chunks: the generic form folds a group's sum into the total at every chunk end (twice per group,
            # each half scaled by the group's scale), the aligned form once per group: f32 rounding apart
            torch.testing.assert_close(fast[k].float(), generic[k].float(), rtol=2 ** -7, atol=2 ** -7, msg=k)
        else:
            assert torch.equal(fast[k], generic[k]), k


# ----------------------------------------------------------------------------- fp16 / fp32 activations
@pytest.mark.parametrize("xdt", [torch.float16, torch.float32])
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(1, 64, 128), (5, 96, 256), (70, 200, 1184), (300, 256, 3072)])
def test_fp16_fp32_activations(NQ, xdt, dt, mnk):
    """The reference's matmuls take activations of any float type, compute in fp32 and return x's dtype
    (quantization.py:258-261, 299-305, 319-346).  Weight-only: the activations are split exactly into bf16 terms and
    the f32 slices of every term are summed, so the result must carry x's precision (half an ulp of x's dtype + the
    f32 accumulation allowance), not bf16's.  W8A8 int8: quantizer, integer contraction and the f32 epilogue are exact,
    so the result is bit-identical to the oracle's arithmetic cast to x's dtype."""
    M, N, K = mnk
    rng = np.random.default_rng(M + N + K)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(xdt).cuda()
    xf = x.float().cpu().numpy()
    wb = rand_bf16(rng, (N, K), 0.05)
    bias = bf16_t(rand_bf16(rng, (N,), 0.5))
    ulp = 2.0 ** -11 if xdt == torch.float16 else 2.0 ** -24

    def close(y, ref64, l1, what, first=None):
        """``first``: the value that was rounded to x's dtype before the bias was added in x's dtype (the reference's
        order for per-group / per-tensor): two roundings, the first one at the magnitude of that value."""
        assert y.dtype == xdt and tuple(y.shape) == ref64.shape
        got = y.float().cpu().numpy().astype(np.float64)
        tol = 1.002 * ulp * (np.abs(ref64) + (0 if first is None else np.abs(first))) + 2.0 ** -17 * l1 + 1e-30
        bad = np.abs(got - ref64) > tol
        assert not bad.any(), (what, int(bad.sum()), float(np.abs(got - ref64).max()))

    # per-row weight-only, per-group, per-tensor
    qo, so = O.per_row_quant(wb, dt)
    deq = O._decode_q(qo, dt).astype(np.float64) * so[:, None]
    l1 = np.abs(xf).astype(np.float64) @ np.abs(deq).T
    for b in (None, bias):
        bb = 0 if b is None else O.bf16_to_f32(bits16(b)).astype(np.float64)[None, :]
        y = NQ.per_row_weight_only_matmul(x, q_t(qo, dt), torch.from_numpy(so).cuda(), b, TDT[dt])
        close(y, xf.astype(np.float64) @ deq.T + bb, l1, f"per-row weight-only {xdt} {dt}")
    g = 32 if K % 128 else 128
    qg, sg = O.per_group_quant(wb, g, dt)
    deq = (O._decode_q(qg, dt).reshape(N, -1, g) * sg[..., None]).reshape(N, K).astype(np.float64)
    y = NQ.per_group_matmul(x, q_t(qg, dt), torch.from_numpy(sg).cuda(), bias, g, TDT[dt])
    close(y, xf.astype(np.float64) @ deq.T + O.bf16_to_f32(bits16(bias)).astype(np.float64)[None, :],
          np.abs(xf).astype(np.float64) @ np.abs(deq).T, f"per-group {xdt} {dt}", first=xf.astype(np.float64) @ deq.T)
    qt, st = O.per_tensor_quant(wb, dt)
    st = float(np.asarray(st).reshape(-1)[0])
    deq = O._decode_q(qt, dt).astype(np.float64) * st
    y = NQ.per_tensor_matmul(x, q_t(qt, dt), torch.tensor(st, dtype=torch.bfloat16).cuda(), None, TDT[dt])
    close(y, xf.astype(np.float64) @ deq.T, np.abs(xf).astype(np.float64) @ np.abs(deq).T, f"per-tensor {xdt} {dt}")
    # the quantizer on the widened values: bit-exact; the W8A8 matmul: int8 bit-exact, fp8 at the fp8 GEMM's tolerance
    qx, sx = NQ.per_row_quant(x, TDT[dt])
    qxo, sxo = O.per_row_quant(xf, dt)
    assert np.array_equal(bits8(qx), np.asarray(qxo).view(np.uint8)) and np.array_equal(sx.cpu().numpy(), sxo)
    y8 = NQ.per_row_matmul(x, q_t(qo, dt), torch.from_numpy(so).cuda(), None, TDT[dt])
    assert y8.dtype == xdt
    accd = O._decode_q(qxo, dt).astype(np.float64) @ O._decode_q(qo, dt).astype(np.float64).T
    if dt == "int8":
        want = ((accd.astype(np.float32) * sxo[:, None]).astype(np.float32) * so[None, :]).astype(np.float32)
        want = torch.from_numpy(want).to(xdt)
        assert torch.equal(y8.cpu(), want)
    else:
        ref = accd * sxo[:, None].astype(np.float64) * so[None, :].astype(np.float64)
        l18 = (np.abs(O._decode_q(qxo, dt)).astype(np.float64) @ np.abs(O._decode_q(qo, dt)).astype(np.float64).T) \
            * sxo[:, None] * so[None, :]
        close(y8, ref, l18, f"w8a8 fp8 {xdt}")
    # the fused SiLU form (interleaved weight) == the plain form followed by SiluAndMul in x's dtype
    if N % 64 == 0:
        il = NQ.interleave_gate_up
        qw_t, sw_t = q_t(qo, dt), torch.from_numpy(so).cuda()
        plain = NQ.per_row_weight_only_matmul(x, qw_t, sw_t, None, TDT[dt])
        fused = NQ.per_row_weight_only_matmul(x, il(qw_t), il(sw_t), None, TDT[dt], silu=True)
        assert torch.equal(fused, torch.nn.functional.silu(plain[:, :N // 2]) * plain[:, N // 2:])


# ----------------------------------------------------------------------------- edge cases / API behaviour
def test_empty_and_leading_dims(NQ):
    w = torch.randn(64, 128, device="cuda", dtype=torch.bfloat16)
    qw, sw = NQ.per_row_quant(w, torch.int8)
    y = NQ.per_row_weight_only_matmul(torch.empty(0, 128, device="cuda", dtype=torch.bfloat16), qw, sw, None, torch.int8)
    assert y.shape == (0, 64)
    x3 = torch.randn(1, 5, 128, device="cuda", dtype=torch.bfloat16)     # the PPL harness feeds [1, T, K]
    for fn in (NQ.per_row_weight_only_matmul, NQ.per_row_matmul):
        y3 = fn(x3, qw, sw, None, torch.int8)
        assert y3.shape == (1, 5, 64) and y3.dtype == torch.bfloat16
        assert torch.equal(y3[0], fn(x3[0], qw, sw, None, torch.int8))


def test_errors_are_loud(NQ):
    from nanovllm import _qlinear as Q
    w = torch.randn(64, 96, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(AssertionError):
        NQ.per_group_quant(w, 64, torch.int8)          # K % group != 0 (reference asserts too)
    with pytest.raises(AssertionError):
        NQ.per_row_quant(w, torch.float16)
    with pytest.raises(Q.QLinearError):
        NQ.per_row_quant(w.double(), torch.int8)       # bf16, fp16 and fp32 inputs are implemented
    w104 = torch.randn(64, 104, device="cuda", dtype=torch.bfloat16)
    qw, sw = NQ.per_row_quant(w104, torch.int8)
    with pytest.raises(Q.QLinearError):                # K = 104 is not a multiple of 32
        NQ.per_row_weight_only_matmul(torch.randn(2, 104, device="cuda", dtype=torch.bfloat16), qw, sw, None, torch.int8)


def test_module_classes_match_golden(NQ, golden):
    """QGroupLinear / QTensorLinear built by from_linear on the same weights the reference used."""
    from nanovllm.layers.linear import ReplicatedLinear
    lin = ReplicatedLinear(256, 64)
    lin.weight.data = bf16_t(golden["module/w"])
    x = bf16_t(golden["module/x"])
    l1 = O.abs_product_sum(golden["module/x"], O.bf16_to_f32(golden["module/w"]))
    y = NQ.QGroupLinear.from_linear(lin, torch.int8, 128)(x)
    assert_gemm_close(bits16(y), O.bf16_to_f32(golden["module/qgroup_int8_g128/y"]).astype(np.float64), l1, True)
    y = NQ.QTensorLinear.from_linear(lin, torch.float8_e4m3fn)(x)
    assert_gemm_close(bits16(y), O.bf16_to_f32(golden["module/qtensor_fp8/y"]).astype(np.float64), l1, True)
    y8 = NQ.QLinear.from_linear(lin, torch.int8)(x)
    q, s = O.per_row_quant(golden["module/w"], "int8")
    assert np.array_equal(bits16(y8), O.per_row_matmul(golden["module/x"], q, s, None, "int8"))


# ----------------------------------------------------------------------------- full-size properties
QWEN3_06B = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]


@pytest.mark.parametrize("nk", QWEN3_06B)
@pytest.mark.parametrize("M", [1, 16, 130, 256, 2048])
def test_full_size_properties(NQ, nk, M):
    """At the real Qwen3-0.6B layer shapes the oracle is too slow to run everywhere, so check
    size-independent properties: (1) int8 W8A8 equals an exact integer GEMM done by torch in int32 on
    the GPU... replaced by float64 on a row/column sample; (2) the result does not depend on M-tiling
    (row i of a batch == the same row run alone); (3) a one-hot activation row returns the weight
    column scaled exactly; (4) determinism."""
    N, K = nk
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    qw, sw = NQ.per_row_quant(w, torch.int8)
    y = NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)
    y_again = NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)
    assert torch.equal(y, y_again), "not deterministic"
    # (2) batch invariance up to f32 summation order (different M pick different tiles / split-K)
    rows = sorted(set([0, M // 2, M - 1]))
    for r in rows:
        y1 = NQ.per_row_weight_only_matmul(x[r:r + 1].contiguous(), qw, sw, None, torch.int8)
        d = (y1[0].float() - y[r].float()).abs()
        assert (d <= y[r].float().abs() * 2.0 ** -7 + 1e-3).all()
    # (1) sampled rows against float64 on the CPU oracle formula
    xs = bits16(x[rows])
    ref = O.w8a16_rowscale_matmul(xs, bits8(qw), sw.cpu().numpy(), None, "int8", return_f64=True)
    l1 = O.abs_product_sum(xs, O._decode_q(bits8(qw), "int8") * sw.cpu().numpy()[:, None])
    assert_gemm_close(bits16(y[rows]), ref, l1, what=f"sampled rows {nk} M={M}")
    # (3) one-hot row: y[0, n] == bf16(q[n, j] * sw[n]) exactly
    j = K // 3
    e = torch.zeros(1, K, device="cuda", dtype=torch.bfloat16)
    e[0, j] = 1.0
    yh = NQ.per_row_weight_only_matmul(e, qw, sw, None, torch.int8)
    expect = (qw[:, j].float() * sw).to(torch.bfloat16)
    assert torch.equal(yh[0], expect)
    # W8A8 int8: exact against integer arithmetic on sampled rows
    y8 = NQ.per_row_matmul(x, qw, sw, None, torch.int8)
    yo = O.per_row_matmul(xs, bits8(qw), sw.cpu().numpy(), None, "int8")
    assert np.array_equal(bits16(y8[rows]), yo)
    # FP8 weight + activation (fp8 MFMA, f32 accumulation) and fp8 per-group g128 at the same shapes
    fp8 = torch.float8_e4m3fn
    qf, sf = NQ.per_row_quant(w, fp8)
    yf = NQ.per_row_matmul(x, NQ.tile_weight(qf), sf, None, fp8)
    assert torch.equal(yf, NQ.per_row_matmul(x, NQ.tile_weight(qf), sf, None, fp8)), "not deterministic"
    reff, l1f = fp8_w8a8_exact(xs, bits8(qf), sf.cpu().numpy())
    assert_gemm_close(bits16(yf[rows]), reff, l1f, what=f"fp8 W8A8 {nk} M={M}")
    qg, sg = NQ.per_group_quant(w, 128, fp8)
    yg = NQ.per_group_matmul(x, NQ.tile_weight(qg), sg, None, 128, fp8)
    wg = (O._decode_q(bits8(qg), "fp8").reshape(N, K // 128, 128) * sg.cpu().numpy()[:, :, None]).reshape(N, K)
    refg = O.per_group_matmul(xs, bits8(qg), sg.cpu().numpy(), None, 128, "fp8", return_f64=True)
    assert_gemm_close(bits16(yg[rows]), refg, O.abs_product_sum(xs, wg), what=f"fp8 per-group {nk} M={M}")


# The layer shapes of the larger BASELINE configurations (Qwen3-4B: per-group g128; Qwen3-8B at TP=8): they take
# the tile / split-K paths that the 0.6B shapes never reach (2048-k slices for K >= 8192, unsplit 64 x 128 tiles).
QWEN3_LARGE = [(6144, 2560), (2560, 4096), (19456, 2560), (2560, 9728), (768, 4096), (4096, 512), (3072, 4096),
               (4096, 1536)]


@pytest.mark.parametrize("nk", QWEN3_LARGE)
@pytest.mark.parametrize("M", [16, 130])
def test_large_model_shapes_sampled_rows(NQ, nk, M):
    """Per-row W8A16, per-group g128 W8A16 and int8 W8A8 at the 4B / 8B-TP8 layer shapes: sampled output rows
    against the oracle (same bars as everywhere: float GEMMs within the bf16 + f32-accumulation allowance, the
    int8 W8A8 GEMM bit-exact), tiled weights as the modules hold them."""
    N, K = nk
    g = torch.Generator(device="cuda").manual_seed(7 * N + K + M)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    rows = sorted(set([0, M // 2, M - 1]))
    xs = bits16(x[rows])
    qw, sw = NQ.per_row_quant(w, torch.int8)
    qt = NQ.tile_weight(qw)
    y = NQ.per_row_weight_only_matmul(x, qt, sw, None, torch.int8)
    assert torch.equal(y, NQ.per_row_weight_only_matmul(x, qw, sw, None, torch.int8)), "tiled != row-major"
    ref = O.w8a16_rowscale_matmul(xs, bits8(qw), sw.cpu().numpy(), None, "int8", return_f64=True)
    l1 = O.abs_product_sum(xs, O._decode_q(bits8(qw), "int8") * sw.cpu().numpy()[:, None])
    assert_gemm_close(bits16(y[rows]), ref, l1, what=f"per-row {nk} M={M}")
    y8 = NQ.per_row_matmul(x, qt, sw, None, torch.int8)
    assert np.array_equal(bits16(y8[rows]), O.per_row_matmul(xs, bits8(qw), sw.cpu().numpy(), None, "int8"))
    qg, sg = NQ.per_group_quant(w, 128, torch.int8)
    yg = NQ.per_group_matmul(x, NQ.tile_weight(qg), sg, None, 128, torch.int8)
    refg = O.per_group_matmul(xs, bits8(qg), sg.cpu().numpy(), None, 128, "int8", return_f64=True)
    wg = O._decode_q(bits8(qg), "int8").reshape(N, K // 128, 128) * sg.cpu().numpy()[:, :, None]
    assert_gemm_close(bits16(yg[rows]), refg, O.abs_product_sum(xs, wg.reshape(N, K)), what=f"per-group {nk} M={M}")
    # the same shapes in fp8: W8A8 on fp8 MFMA and per-group g128 weight-only
    fp8 = torch.float8_e4m3fn
    qf, sf = NQ.per_row_quant(w, fp8)
    yf = NQ.per_row_matmul(x, NQ.tile_weight(qf), sf, None, fp8)
    reff, l1f = fp8_w8a8_exact(xs, bits8(qf), sf.cpu().numpy())
    assert_gemm_close(bits16(yf[rows]), reff, l1f, what=f"fp8 W8A8 {nk} M={M}")
    qg, sg = NQ.per_group_quant(w, 128, fp8)
    yg = NQ.per_group_matmul(x, NQ.tile_weight(qg), sg, None, 128, fp8)
    wg = (O._decode_q(bits8(qg), "fp8").reshape(N, K // 128, 128) * sg.cpu().numpy()[:, :, None]).reshape(N, K)
    refg = O.per_group_matmul(xs, bits8(qg), sg.cpu().numpy(), None, 128, "fp8", return_f64=True)
    assert_gemm_close(bits16(yg[rows]), refg, O.abs_product_sum(xs, wg), what=f"fp8 per-group {nk} M={M}")


@pytest.mark.parametrize("tiled", [False, True])
def test_fp8_weight_widening_is_exact_for_every_byte(NQ, tiled):
    """The in-register e4m3fn -> bf16 widening of the W8A16 GEMM (one packed conversion per two weights) must
    reproduce every finite e4m3fn value exactly: with a one-hot activation row and unit scales the output IS the
    decoded weight.  (0x7F / 0xFF are the format's NaNs; a quantizer never emits them.)"""
    codes = [c for c in range(256) if c not in (0x7F, 0xFF)]
    N, K = 256, 64
    q = torch.zeros(N, K, dtype=torch.uint8)
    for j in range(K):
        q[:, j] = torch.tensor([codes[(n + 7 * j) % len(codes)] for n in range(N)], dtype=torch.uint8)
    qw = q.cuda().view(torch.float8_e4m3fn)
    sw = torch.ones(N, device="cuda", dtype=torch.float32)
    w = NQ.tile_weight(qw) if tiled else qw
    for j in (0, 1, 15, 16, 31, 33, 63):
        e = torch.zeros(3, K, device="cuda", dtype=torch.bfloat16)
        e[1, j] = 1.0
        y = NQ.per_row_weight_only_matmul(e, w, sw, None, torch.float8_e4m3fn)
        expect = qw[:, j].float().to(torch.bfloat16)
        assert torch.equal(y[1], expect), f"column {j}"
        assert torch.equal(qw[:, j].float(), expect.float()), "e4m3fn must fit bf16 exactly"
        assert y[0].abs().max().item() == 0 and y[2].abs().max().item() == 0


@pytest.mark.parametrize("dt", ["int8", "fp8"])
def test_one_round_tile_rule_all_variants(NQ, dt):
    """K = 1024 with 176..256 tiles of 32 x 128 selects the eight-wave tile that keeps the whole K resident; every
    arithmetic variant must take that path correctly (per-row W8A16, per-group, W8A8, fused SiLU)."""
    N, K, M = 4096, 1024, 200
    tdt = TDT[dt]
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    rows = [0, 97, M - 1]
    xs = bits16(x[rows])
    qw, sw = NQ.per_row_quant(w, tdt)
    y = NQ.per_row_weight_only_matmul(x, NQ.tile_weight(qw), sw, None, tdt)
    wd = O._decode_q(bits8(qw), dt) * sw.cpu().numpy()[:, None]
    ref = O.w8a16_rowscale_matmul(xs, bits8(qw), sw.cpu().numpy(), None, dt, return_f64=True)
    assert_gemm_close(bits16(y[rows]), ref, O.abs_product_sum(xs, wd), what=f"per-row {dt}")
    qg, sg = NQ.per_group_quant(w, 128, tdt)
    yg = NQ.per_group_matmul(x, NQ.tile_weight(qg), sg, None, 128, tdt)
    wg = (O._decode_q(bits8(qg), dt).reshape(N, K // 128, 128) * sg.cpu().numpy()[:, :, None]).reshape(N, K)
    refg = O.per_group_matmul(xs, bits8(qg), sg.cpu().numpy(), None, 128, dt, return_f64=True)
    assert_gemm_close(bits16(yg[rows]), refg, O.abs_product_sum(xs, wg), what=f"per-group {dt}")
    y8 = NQ.per_row_matmul(x, NQ.tile_weight(qw), sw, None, tdt)
    if dt == "int8":
        assert np.array_equal(bits16(y8[rows]), O.per_row_matmul(xs, bits8(qw), sw.cpu().numpy(), None, dt))
    else:
        ref8, l18 = fp8_w8a8_exact(xs, bits8(qw), sw.cpu().numpy())
        assert_gemm_close(bits16(y8[rows]), ref8, l18, what="fp8 W8A8")
    # fused SiLU epilogue on the same tile: bit-identical to GEMM + SiluAndMul
    qi, si = NQ.interleave_gate_up(qw), NQ.interleave_gate_up(sw)
    from nanovllm import ops
    fused = NQ.per_row_weight_only_matmul(x, NQ.tile_weight(qi), si, None, tdt, silu=True)
    assert torch.equal(fused, ops.silu_mul(y))


# ---- W8A8 with the activation quantization inside the GEMM (partial row maxima in, no quantized copy of x) ----
@pytest.mark.parametrize("dt", ["int8", "fp8"])
@pytest.mark.parametrize("M,N,K,parts", [(1, 64, 64, 1), (7, 130, 128, 2), (33, 1024, 2048, 8), (130, 1024, 3072, 96),
                                         (256, 4096, 1024, 4), (300, 192, 96, 3), (2048, 1024, 2048, 16)])
def test_w8a8_dynamic_quant_in_gemm_bit_identical(NQ, dt, M, N, K, parts):
    """ql_gemm_w8a8_dynq(x, partial row maxima) == ql_quant_per_row(x) + ql_gemm_w8a8, bit for bit, for partial
    maxima over any split of the columns (the helper kernel's equal ranges, and an uneven split with a part that
    is zero), incl. an all-zero row (the 1e-8 floor) and the tiled weight layout."""
    tdt = TDT[dt]
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    if M > 2:
        x[2] = 0
        x[1] *= 37.0
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    qw, sw = NQ.per_row_quant(w, tdt)
    want = NQ.per_row_matmul(x, qw, sw, None, tdt)
    pm = NQ.row_absmax_parts(x, parts)
    assert torch.equal(pm.max(dim=1).values, x.float().abs().max(dim=1).values)
    got = NQ.w8a8_dynq_matmul(x, pm, qw, sw, None, tdt)
    assert torch.equal(got, want)
    # an uneven split: the first 8 columns, the rest, and a part that saw nothing
    pm2 = torch.stack([x[:, :8].float().abs().amax(1), x[:, 8:].float().abs().amax(1), torch.zeros(M, device="cuda")], 1)
    assert torch.equal(NQ.w8a8_dynq_matmul(x, pm2.contiguous(), NQ.tile_weight(qw) if N % 32 == 0 else qw, sw, None, tdt), want)


@pytest.mark.parametrize("tile", [(1, 1, 4, 2), (1, 1, 8, 4), (1, 2, 4, 8), (2, 2, 4, 8), (2, 4, 2, 4), (2, 4, 1, 4), (4, 4, 1, 4),
                                  (4, 8, 1, 4), (8, 4, 1, 4), (1, 4, 2, 16)])
@pytest.mark.parametrize("split", [1, 3])
def test_w8a8_dynamic_quant_every_tile(NQ, tile, split):
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, N, K = 150, 256, 3072
    g = torch.Generator(device="cuda").manual_seed(sum(tile) + split)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    for tdt in (torch.int8, torch.float8_e4m3fn):
        qw, sw = NQ.per_row_quant(w, tdt)
        pm = NQ.row_absmax_parts(x, 6)
        Q.workspace.reserve("cuda", 64 << 20)
        assert lib.ql_gemm_force_config(*tile, split) == 0
        try:
            want = NQ.per_row_matmul(x, qw, sw, None, tdt)
            got = NQ.w8a8_dynq_matmul(x, pm, qw, sw, None, tdt)
        finally:
            lib.ql_gemm_force_config(0, 0, 0, 0, 0)
        assert torch.equal(got, want), (tile, split, tdt)


@pytest.mark.parametrize("dt", ["int8", "fp8"])
@pytest.mark.parametrize("M,tile", [(5, None), (64, None), (130, None), (256, None), (700, None), (2100, None),
                                    (70, (1, 2, 4, 8)), (70, (2, 2, 4, 8)), (70, (2, 4, 2, 4)), (70, (2, 4, 2, 16)),
                                    (70, (1, 4, 2, 16)), (200, (2, 4, 1, 4)), (200, (4, 4, 1, 4)), (200, (4, 8, 1, 4))])
def test_w8a8_mlp_without_a_quantization_launch(NQ, dt, M, tile):
    """gate_up (W8A8, SiLU epilogue) -> down (W8A8): the gate/up GEMM also emits the |max| of every output row per
    32-column strip, and the down GEMM quantizes its input while staging it.  Outputs of both GEMMs are bit-identical
    to the chain with a separate ql_quant_per_row launch; the emitted maxima are exact."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    tdt = TDT[dt]
    I, H = 384, 256                                   # intermediate (12 strips), hidden
    g = torch.Generator(device="cuda").manual_seed(M + (sum(tile) if tile else 0))
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16, generator=g)
    wgu = (torch.randn(2 * I, H, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    wd = (torch.randn(H, I, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    qgu, sgu = NQ.per_row_quant(wgu, tdt)
    qgu, sgu = NQ.interleave_gate_up(qgu), NQ.interleave_gate_up(sgu)
    qd, sd = NQ.per_row_quant(wd, tdt)
    qx, sx = NQ.per_row_quant(x, tdt)
    if tile:
        assert lib.ql_gemm_force_config(*tile, 1) == 0
    try:
        y_ref = NQ.w8a8_matmul(qx, sx, qgu, sgu, None, tdt, silu=True)
        y, rm = NQ.w8a8_matmul(qx, sx, qgu, sgu, None, tdt, silu=True, return_rowmax=True)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    assert torch.equal(y, y_ref)
    assert rm.shape == (M, 2 * I // 64)
    assert torch.equal(rm, y.float().abs().view(M, I // 32, 32).amax(-1))
    assert torch.equal(NQ.w8a8_dynq_matmul(y, rm, qd, sd, None, tdt), NQ.per_row_matmul(y, qd, sd, None, tdt))


@pytest.mark.parametrize("tile,I", [((4, 4, 1, 4), 96), ((4, 8, 1, 4), 96), ((4, 8, 1, 4), 1216), (None, 1216)])
def test_w8a8_silu_rowmax_tile_wider_than_the_columns_left(NQ, tile, I):
    """Row maxima from the SiLU epilogue when the tile's width does not divide N (N = 192 on 128- and 256-column tiles,
    N = 2432 = the Qwen3-4B gate/up shard under TP = 8 at prefill size): strips past the last one have no slot in
    rowmax_out and must not be written (they used to land in the next row's first slot)."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    M, H = 1100, 128
    g = torch.Generator(device="cuda").manual_seed(I)
    x = torch.randn(M, H, device="cuda", dtype=torch.bfloat16, generator=g)
    wgu = (torch.randn(2 * I, H, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    qgu, sgu = NQ.per_row_quant(wgu, torch.int8)
    qgu, sgu = NQ.interleave_gate_up(qgu), NQ.interleave_gate_up(sgu)
    qx, sx = NQ.per_row_quant(x, torch.int8)
    if tile:
        assert lib.ql_gemm_force_config(*tile, 1) == 0
    try:
        for _ in range(3):
            y, rm = NQ.w8a8_matmul(qx, sx, qgu, sgu, None, torch.int8, silu=True, return_rowmax=True)
            assert torch.equal(rm, y.float().abs().view(M, I // 32, 32).amax(-1)), (tile, I)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("tile", [(1, 1, 8, 4), (1, 1, 4, 4), (1, 2, 4, 8), (2, 2, 4, 8), (2, 4, 2, 4)])
@pytest.mark.parametrize("split", [1, 2, 5])
def test_w8a8_deferred_splitk_norm_bit_identical(NQ, tile, split):
    """The W8A8 form of the deferred split-K: raw slices (int32 sums for int8 x int8, f32 for fp8 x fp8) finished
    by the residual add + RMSNorm launch with the dual-scale epilogue, == ql_gemm_w8a8 (same tile, same slices)
    followed by ql_add_rmsnorm / ql_add_rmsnorm_quant, bit for bit; with and without a bias, M and N tails, a
    ragged last slice."""
    from nanovllm import _qlinear as Q, ops
    lib = Q.load()
    M, N, K = 70, 200, 1184
    rng = np.random.default_rng(33)
    x = bf16_t(rand_bf16(rng, (M, K)))
    w = bf16_t(rand_bf16(rng, (N, K), 0.05))
    bias = torch.nn.Parameter(bf16_t(rand_bf16(rng, (N,), 0.5)), requires_grad=False)
    res = bf16_t(rand_bf16(rng, (M, N)))
    nw = bf16_t(O.f32_to_bf16((1 + 0.1 * rng.standard_normal(N)).astype(np.float32)))
    try:
        assert lib.ql_gemm_force_config(tile[0], tile[1], tile[2], tile[3], split) == 0
        for tdt, b in ((torch.int8, None), (torch.int8, bias), (torch.float8_e4m3fn, None), (torch.float8_e4m3fn, bias)):
            mod = NQ.QLinear(tdt, w, b).tile()
            qx, sx = NQ.per_row_quant(x, tdt)
            ps = NQ.w8a8_partial_matmul(mod, qx, sx)
            assert ps.parts.shape == (lib.ql_gemm_partial_slices(M, N, K, Q.QL_PER_ROW), M, N)
            assert ps.parts.dtype == (torch.int32 if tdt == torch.int8 else torch.float32)
            y = mod.forward_quantized(qx, sx)
            out, new_res = ops.add_rmsnorm(ps, res, nw, 1e-6)
            ref_out, ref_res = ops.add_rmsnorm(y, res, nw, 1e-6)
            assert torch.equal(new_res, ref_res) and torch.equal(out, ref_out), (tile, split, tdt)
            for qd in (torch.int8, torch.float8_e4m3fn):
                q1, s1, r1 = ops.add_rmsnorm_quant(ps, res, nw, 1e-6, qd)
                q0, s0, r0 = ops.add_rmsnorm_quant(y, res, nw, 1e-6, qd)
                assert torch.equal(q1.view(torch.uint8), q0.view(torch.uint8)) and torch.equal(s1, s0) and torch.equal(r1, r0)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)


@pytest.mark.parametrize("dt", ["int8", "fp8"])
@pytest.mark.parametrize("nk", [(1024, 2048), (1024, 3072)])
@pytest.mark.parametrize("M", [1, 16, 130, 256, 512, 600])
def test_w8a8_deferred_splitk_module_route(NQ, dt, nk, M):
    """QLinear.forward_partial_quantized at the o_proj / down_proj shapes: a PartialSum where the plan has slices to
    defer (decode sizes), the plain output otherwise; either way the norm that follows agrees with the plain route
    (the two routes slice K differently: int8 sums are exact, so int8 is bit-identical; fp8 to the GEMM's rounding)."""
    from nanovllm import ops
    tdt = TDT[dt]
    N, K = nk
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16, generator=g)
    nw = torch.ones(N, device="cuda", dtype=torch.bfloat16)
    mod = NQ.QLinear(tdt, w, None).tile()
    qx, sx = NQ.per_row_quant(x, tdt)
    y = mod.forward_partial_quantized(qx, sx)
    from nanovllm import _qlinear as Q
    slices = Q.load().ql_gemm_partial_slices(M, N, K, Q.QL_PER_ROW)
    assert hasattr(y, "parts") == (M <= 512 and slices > 1)
    if M in (1, 16):
        assert hasattr(y, "parts")                       # small decode batches do defer at these shapes
    out, new_res = ops.add_rmsnorm(y, res, nw, 1e-6)
    ref_out, ref_res = ops.add_rmsnorm(mod.forward_quantized(qx, sx), res, nw, 1e-6)
    if dt == "int8":
        assert torch.equal(out, ref_out) and torch.equal(new_res, ref_res)
    else:
        d = (out.float() - ref_out.float()).abs()
        assert (d <= 2.0 ** -6 * ref_out.float().abs() + 2e-2).all()
