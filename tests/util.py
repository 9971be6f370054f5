"""Helpers shared by the tests: torch <-> bit-pattern numpy views and the f32-accumulation tolerance."""
import numpy as np
import torch

from oracle import qlinear_oracle as O

TDT = {"int8": torch.int8, "fp8": torch.float8_e4m3fn}


def bf16_t(bits, device="cuda"):
    """uint16 bit patterns -> torch bf16 tensor"""
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16).to(device)


def q_t(bits, dt, device="cuda"):
    """uint8/int8 storage -> torch int8 / float8_e4m3fn tensor"""
    u = torch.from_numpy(np.ascontiguousarray(bits).view(np.uint8))
    return (u.view(torch.int8) if dt == "int8" else u.view(torch.float8_e4m3fn)).to(device)


def bits16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    t = t.detach().cpu().contiguous()
    return (t.view(torch.uint8) if t.dtype != torch.int8 else t.view(torch.uint8)).numpy()


def rand_bf16(rng, shape, std=1.0):
    return O.f32_to_bf16((rng.standard_normal(shape) * std).astype(np.float32))


def assert_gemm_close(y_bits, ref64, l1, double_round=False, what=""):
    """Kernel output (bf16 of an f32-accumulated sum, order unspecified) against the exact value.
    Tolerance per element: half a bf16 ulp of the result (relative 2**-8; twice that when the
    reference rounds twice, i.e. bf16(bf16(v)+bias)) plus an f32 accumulation term 2**-18 * L1 where
    L1 = sum_k |x*w| (worst-case f32 summation error is K * 2**-24 * L1; K <= 16384 here)."""
    got = O.bf16_to_f32(y_bits).astype(np.float64)
    rel = (2.0 ** -7 if double_round else 2.0 ** -8) * 1.002
    tol = rel * np.abs(ref64) + 2.0 ** -18 * l1 + 1e-30
    bad = np.abs(got - ref64) > tol
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(
            f"{what}: {bad.sum()}/{bad.size} outside tolerance; first at {tuple(i)}: got {got[tuple(i)]!r} "
            f"ref {ref64[tuple(i)]!r} tol {tol[tuple(i)]!r}")


# ---- the narrowing casts of Triton's CPU interpreter (triton/runtime/interpreter.py, _convert_float) ----
# tests/golden/triton_golden.npz was produced by the reference's kernels under TRITON_INTERPRET=1, whose
# f32 -> bf16 cast truncates and whose f32 -> e4m3 cast adds the first dropped mantissa bit WITHOUT carrying
# into the exponent (the sum is OR-ed into place).  Compiled Triton rounds to nearest even, so these are
# properties of the vector generator, restated here only to compare pre-cast values bit for bit.
def interp_f32_to_bf16(f):
    return (np.ascontiguousarray(f, dtype=np.float32).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def interp_f32_to_e4m3(f):
    """Returns (bits, valid): ``valid`` marks the elements whose magnitude is an e4m3 NORMAL number or zero;
    the interpreter's handling of the subnormal range is left out (callers skip those few elements)."""
    f = np.ascontiguousarray(f, dtype=np.float32)
    u = f.view(np.uint32)
    sign = (u >> np.uint32(31)).astype(np.uint32)
    exp = ((u >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int64)
    sig = u & np.uint32(0x7FFFFF)
    e_out = np.clip(exp - 127 + 7, 0, 15).astype(np.uint32)
    m_out = ((sig >> np.uint32(20)) & np.uint32(7)) + ((sig & np.uint32(1 << 19)) > 0)
    bits = (sign << np.uint32(7)) | (e_out << np.uint32(3)) | m_out.astype(np.uint32)
    zero = (u & np.uint32(0x7FFFFFFF)) == 0
    bits = np.where(zero, sign << np.uint32(7), bits)
    valid = zero | (exp - 127 + 7 >= 1)
    return bits.astype(np.uint8), valid


# ----------------------------------------------------------------------------------------------------------------
# Per-group GEMM cases run twice, in two processes: on the group-aligned form of the kernels' main loop (128-k groups
# that the unrolled steps meet at fixed positions: the first MFMA of a group takes a zero C operand) and, with
# QL_GROUP_GENERIC=1 in the environment, on the generic form.  The two must agree bit for bit (where both fold a group's sum once).
GROUP_FORM_TILES = [((2, 4, 2, 4), 1), ((2, 4, 2, 4), 3), ((1, 2, 4, 4), 2), ((4, 8, 1, 4), 1), ((4, 8, 1, 4), 3),
                    ((4, 8, 1, 2), 1), ((4, 8, 1, 2), 2), ((4, 8, 1, 2), 3), ((1, 4, 2, 16), 1), ((1, 1, 8, 8), 1),
                    ((1, 4, 1, 32), 1), ((2, 2, 4, 8), 1), ((2, 4, 1, 4), -8), ((4, 8, 1, 2), -5), (None, 0)]


def group_form_outputs():
    """{case name: bf16 output bits} of per-group (g = 128) W8A16 GEMMs over GROUP_FORM_TILES, both weight dtypes,
    plain and SiLU epilogues, K with an even and an odd number of groups (a K slice of the 64-k-chunk tile may then
    start in the middle of a group, which sends that slice to the generic form)."""
    import torch
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    lib = Q.load()
    Q.workspace.reserve("cuda", 64 << 20)
    out = {}
    g = torch.Generator().manual_seed(128)
    try:
        for (M, N, K) in ((40, 320, 1152), (200, 256, 1024), (130, 512, 2560)):
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
            w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
            for dt in (torch.int8, torch.float8_e4m3fn):
                q, s = NQ.per_group_quant(w, 128, dt)
                qt = NQ.tile_weight(q)
                for tile, split in GROUP_FORM_TILES:
                    if tile is None:
                        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
                    elif lib.ql_gemm_force_config(*tile, split) != 0:
                        continue
                    name = f"{M}x{N}x{K} {dt} {tile} {split}"
                    try:
                        out[name] = NQ.per_group_matmul(x, qt, s, None, 128, dt).cpu()
                        out[name + " silu"] = NQ.per_group_matmul(x, qt, s, None, 128, dt, silu=True).cpu()
                    except Q.QLinearError:
                        pass                                  # (a split the tile's chunk does not allow for this K)
    finally:
        lib.ql_gemm_force_config(0, 0, 0, 0, 0)
    return out
