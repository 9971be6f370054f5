"""The numpy oracle against vectors produced by the reference's own torch code on CPU
(tests/golden/quant_golden.npz, generator: tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import qlinear_oracle as O

CASES = ["m1_n64_k128", "m3_n64_k256", "m16_n48_k512", "m130_n64_k256", "m5_n8_k3072"]
DTYPES = ["int8", "fp8"]


def _ulp_bf16(bits):
    """size of one bf16 ulp at the magnitude of each element (as f32)."""
    f = np.abs(O.bf16_to_f32(bits))
    e = np.floor(np.log2(np.maximum(f, 2.0 ** -126)))
    return (2.0 ** (e - 7)).astype(np.float64)


def assert_bf16_close(got_bits, ref_bits, exact64=None, what=""):
    """Both sides round the same real number to bf16 but sum in different orders/precisions, so they
    may land on neighbouring bf16 values when the exact value sits near a rounding tie: allow one
    ulp, and require bit equality for at least 97 % of the elements."""
    got = O.bf16_to_f32(got_bits).astype(np.float64)
    ref = O.bf16_to_f32(ref_bits).astype(np.float64)
    ulp = np.maximum(_ulp_bf16(got_bits), _ulp_bf16(ref_bits))
    bad = np.abs(got - ref) > ulp * 1.0001
    assert not bad.any(), f"{what}: {bad.sum()} elements differ by more than 1 bf16 ulp"
    same = (np.asarray(got_bits) == np.asarray(ref_bits)).mean()
    assert same >= 0.97, f"{what}: only {same:.3f} bit-identical"


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_per_tensor_quant_bit_exact(golden, case, dt):
    q, s = O.per_tensor_quant(golden[f"{case}/w"], dt)
    assert np.array_equal(q.view(np.uint8), golden[f"{case}/{dt}/tensor/qw"])
    assert np.float32(s) == golden[f"{case}/{dt}/tensor/sw"][0]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("g", [64, 128])
def test_per_group_quant_bit_exact(golden, case, dt, g):
    q, s = O.per_group_quant(golden[f"{case}/w"], g, dt)
    assert np.array_equal(q.view(np.uint8), golden[f"{case}/{dt}/g{g}/qw"])
    assert np.array_equal(s, golden[f"{case}/{dt}/g{g}/sw"])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_per_tensor_matmul(golden, case, dt):
    x, b = golden[f"{case}/x"], golden[f"{case}/b"]
    qw, sw = golden[f"{case}/{dt}/tensor/qw"], golden[f"{case}/{dt}/tensor/sw"][0]
    assert_bf16_close(O.per_tensor_matmul(x, qw, sw, None, dt), golden[f"{case}/{dt}/tensor/y"], what="y")
    assert_bf16_close(O.per_tensor_matmul(x, qw, sw, b, dt), golden[f"{case}/{dt}/tensor/y_bias"], what="y+b")


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("g", [64, 128])
def test_per_group_matmul(golden, case, dt, g):
    x, b = golden[f"{case}/x"], golden[f"{case}/b"]
    qw, sw = golden[f"{case}/{dt}/g{g}/qw"], golden[f"{case}/{dt}/g{g}/sw"]
    assert_bf16_close(O.per_group_matmul(x, qw, sw, None, g, dt), golden[f"{case}/{dt}/g{g}/y"], what="y")
    assert_bf16_close(O.per_group_matmul(x, qw, sw, b, g, dt), golden[f"{case}/{dt}/g{g}/y_bias"], what="y+b")


@pytest.mark.parametrize("case", CASES[:2])
@pytest.mark.parametrize("dt", DTYPES)
def test_fake_quant_bit_exact(golden, case, dt):
    w = golden[f"{case}/w"]
    assert np.array_equal(O.fake_per_row_quant(w, dt), golden[f"{case}/{dt}/fake_row/w"])
    assert np.array_equal(O.fake_per_tensor_quant(w, dt), golden[f"{case}/{dt}/fake_tensor/w"])
    assert np.array_equal(O.fake_per_group_quant(w, 128, dt), golden[f"{case}/{dt}/fake_g128/w"])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", DTYPES)
def test_per_row_weight_quant_matches_fake_row(golden, case, dt):
    """The Triton per-row quantizer has no golden output; its restatement must dequantise to the
    reference's fake-per-row weights except where 'x*(Q/max)' vs 'x/max*Q' and half-away vs
    half-even rounding legitimately differ (counted, must be rare), and the W8A16 GEMM built on it
    must reproduce ``F.linear(x, fake_per_row_quant(W))`` to bf16-weight-rounding accuracy."""
    w, x = golden[f"{case}/w"], golden[f"{case}/x"]
    q, s = O.per_row_quant(w, dt)
    deq = O._decode_q(q, dt) * s[:, None]
    fake = O.bf16_to_f32(O.fake_per_row_quant(w, dt))
    step = s[:, None] * (1.0 if dt == "int8" else np.maximum(np.abs(O._decode_q(q, dt)), 1.0) / 8)
    # The fake route rounds the step amax/127 to bf16 (2**-9) and the product to bf16 (2**-9), so the
    # two dequantised weights agree to 2**-7 relative - or differ by one quantisation step where
    # 'x*(Q/max)' vs 'x/max*Q' or half-away vs half-even rounding land on different integers.
    close = np.abs(deq - fake) <= np.abs(fake) * 2.0 ** -7 + 1e-30
    one_step = np.abs(deq - fake) <= step * 1.001 + np.abs(fake) * 2.0 ** -7
    assert one_step.all()
    assert close.mean() > 0.995, f"{(~close).sum()} of {close.size} weights off by a quantisation step"
    y = O.bf16_to_f32(O.w8a16_rowscale_matmul(x, q, s, None, dt)).astype(np.float64)
    y_ref = O.bf16_to_f32(golden[f"{case}/{dt}/fake_row/y"]).astype(np.float64)
    l1 = O.abs_product_sum(x, fake)
    # worst case of a 2**-8 relative perturbation of every weight, plus output rounding, plus the
    # rare one-step weights
    extra = O.abs_product_sum(x, np.where(close, 0.0, np.abs(deq - fake)))
    assert (np.abs(y - y_ref) <= 2.0 ** -7 * np.abs(y_ref) + 2.0 ** -8 * l1 + extra + 1e-30).all()
    assert np.abs(y - y_ref).max() / np.abs(y_ref).max() < 0.02  # reference self-test allows 0.08


def test_module_outputs(golden):
    w, x = golden["module/w"], golden["module/x"]
    q, s = O.per_group_quant(w, 128, "int8")
    assert_bf16_close(O.per_group_matmul(x, q, s, None, 128, "int8"), golden["module/qgroup_int8_g128/y"])
    q, s = O.per_tensor_quant(w, "fp8")
    assert_bf16_close(O.per_tensor_matmul(x, q, s, None, "fp8"), golden["module/qtensor_fp8/y"])


@pytest.mark.parametrize("tp", [1, 2, 4])
def test_tp_weight_layout(golden, tp):
    f = {k: golden[f"tp/full/{k}"] for k in ("q", "k", "v", "o", "gate", "up", "down")}
    for r in range(tp):
        assert np.array_equal(O.pack_qkv(f["q"], f["k"], f["v"], tp, r), golden[f"tp/{tp}/{r}/qkv"])
        assert np.array_equal(O.pack_gate_up(f["gate"], f["up"], tp, r), golden[f"tp/{tp}/{r}/gate_up"])
        assert np.array_equal(O.shard_cols(f["o"], tp, r), golden[f"tp/{tp}/{r}/o"])
        assert np.array_equal(O.shard_cols(f["down"], tp, r), golden[f"tp/{tp}/{r}/down"])


def test_number_formats_against_torch():
    """bf16 / e4m3fn conversions of the oracle against torch's CPU casts on every bit pattern / a
    dense f32 sample (torch is the implementation the reference relies on)."""
    import torch
    rng = np.random.default_rng(0)
    f = np.concatenate([
        rng.standard_normal(200000).astype(np.float32) * 100,
        rng.standard_normal(100000).astype(np.float32) * 1e-2,
        np.array([0, -0.0, 448, 449, 463.9, 464, 479.9, -448, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9,
                  2.0 ** -6, 0.0175, 1e-8, 240, 256], dtype=np.float32)])
    t = torch.from_numpy(f)
    assert np.array_equal(O.f32_to_bf16(f), t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    f8 = f[np.abs(f) < 470]
    t8 = torch.from_numpy(f8)
    assert np.array_equal(O.f32_to_fp8(f8), t8.to(torch.float8_e4m3fn).view(torch.uint8).numpy())
    allb = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy(allb).view(torch.float8_e4m3fn).float().numpy()
    got = O.fp8_to_f32(allb)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    assert np.array_equal(ref[~np.isnan(ref)], got[~np.isnan(got)])
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    refh = torch.from_numpy(allh.view(np.int16)).view(torch.bfloat16).float().numpy()
    goth = O.bf16_to_f32(allh)
    m = ~np.isnan(refh)
    assert np.array_equal(refh[m], goth[m])
