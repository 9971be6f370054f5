"""Generate golden vectors from the reference's own pure-torch quantization code.

Run ONLY in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does
------------
* Imports ``nanovllm.utils.quantization`` / ``nanovllm.layers.linear`` from the
  read-only reference checkout.  Two third-party modules the reference imports
  but that are not installed here (``flash_attn``, ``torchao.quantization``) are
  replaced by empty name-only modules in ``sys.modules``; nothing on the
  quantized-linear path calls into them.
* Runs the reference functions that are plain torch (CPU-runnable):
  ``per_tensor_quant``, ``per_group_quant``, ``per_tensor_matmul``,
  ``per_group_matmul``, ``fake_per_{tensor,row,group}_quant`` and the
  ``linear.py`` weight loaders, on seeded inputs.
* Writes inputs and outputs to ``tests/golden/*.npz`` (bf16 stored as uint16
  bit patterns, fp8-e4m3fn as uint8 bit patterns).

The Triton per-row kernels cannot run without a GPU driver, so they have no
vectors here; their oracle follows the kernel text and is cross-checked
against ``fake_per_row_quant`` outputs stored by this script.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    fa = types.ModuleType("flash_attn")
    for name in ("flash_attn_varlen_func", "flash_attn_with_kvcache", "flash_attn_func"):
        setattr(fa, name, None)
    sys.modules["flash_attn"] = fa
    ta = types.ModuleType("torchao")
    taq = types.ModuleType("torchao.quantization")
    for name in ("quantize_", "Int8DynamicActivationInt8WeightConfig",
                 "Float8DynamicActivationFloat8WeightConfig"):
        setattr(taq, name, None)
    ta.quantization = taq
    sys.modules["torchao"] = ta
    sys.modules["torchao.quantization"] = taq


def bits16(t):  # bf16 tensor -> uint16 bit pattern
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def bits8(t):  # int8 / fp8 tensor -> uint8 bit pattern
    if t.dtype == torch.int8:
        return t.contiguous().numpy().view(np.uint8).copy()
    return t.contiguous().view(torch.uint8).numpy().copy()


CASES = [  # name, M, N, K, w_std
    ("m1_n64_k128", 1, 64, 128, 1.0),
    ("m3_n64_k256", 3, 64, 256, 1.0),
    ("m16_n48_k512", 16, 48, 512, 0.02),
    ("m130_n64_k256", 130, 64, 256, 0.02),
    ("m5_n8_k3072", 5, 8, 3072, 0.02),   # an 8-row slice of a Qwen3-0.6B down_proj shape
]


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    _stub_modules()
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", world_size=1, rank=0)
    import torch.nn.functional as F
    from nanovllm.utils import quantization as Q
    from nanovllm.layers import linear as L

    dts = {"int8": torch.int8, "fp8": torch.float8_e4m3fn}
    out = {}
    for ci, (name, M, N, K, wstd) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + ci)
        x = torch.randn(M, K, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * wstd).to(torch.bfloat16)
        b = torch.randn(N, generator=g).to(torch.bfloat16)
        # make a few exact .5 ties and an all-zero row so rounding/clamp paths are exercised
        w[0, :] = 0
        if N > 2:
            w[1, 0] = w[1].abs().max()  # row max at a known place
        out[f"{name}/x"] = bits16(x)
        out[f"{name}/w"] = bits16(w)
        out[f"{name}/b"] = bits16(b)
        for dn, dt in dts.items():
            # ---- per tensor (quantization.py:166-191, 237-262)
            qw, sw = Q.per_tensor_quant(w, dt)
            out[f"{name}/{dn}/tensor/qw"] = bits8(qw)
            out[f"{name}/{dn}/tensor/sw"] = np.array([float(sw.float())], dtype=np.float32)
            out[f"{name}/{dn}/tensor/sw_dtype"] = np.array(str(sw.dtype))
            out[f"{name}/{dn}/tensor/y"] = bits16(Q.per_tensor_matmul(x, qw, sw, None, dt))
            out[f"{name}/{dn}/tensor/y_bias"] = bits16(Q.per_tensor_matmul(x, qw, sw, b, dt))
            # ---- per group (quantization.py:194-234, 265-307)
            for gs in (64, 128):
                qw, sw = Q.per_group_quant(w, gs, dt)
                out[f"{name}/{dn}/g{gs}/qw"] = bits8(qw)
                out[f"{name}/{dn}/g{gs}/sw"] = sw.numpy().astype(np.float32)
                out[f"{name}/{dn}/g{gs}/y"] = bits16(Q.per_group_matmul(x, qw, sw, None, gs, dt))
                out[f"{name}/{dn}/g{gs}/y_bias"] = bits16(Q.per_group_matmul(x, qw, sw, b, gs, dt))
            # ---- fake quant (quantization.py:404-445): W8A16 numerics
            fr = Q.fake_per_row_quant(w, dt)
            out[f"{name}/{dn}/fake_row/y"] = bits16(F.linear(x, fr))
            if ci < 2:
                out[f"{name}/{dn}/fake_row/w"] = bits16(fr)
                out[f"{name}/{dn}/fake_tensor/w"] = bits16(Q.fake_per_tensor_quant(w, dt))
                out[f"{name}/{dn}/fake_g128/w"] = bits16(Q.fake_per_group_quant(w, 128, dt))
        # 3-D activations as the PPL harness feeds them ([1, T, K]); leading dims must be restored
        if M == 3:
            x3 = x.view(1, M, K)
            qw, sw = Q.per_group_quant(w, 128, torch.int8)
            y3 = Q.per_group_matmul(x3, qw, sw, None, 128, torch.int8)
            out[f"{name}/int8/g128/y3_shape"] = np.array(list(y3.shape), dtype=np.int64)

    # ---- module classes built through from_linear on a reference ReplicatedLinear
    lin = L.ReplicatedLinear(256, 64)
    g = torch.Generator().manual_seed(7)
    wl = (torch.randn(64, 256, generator=g) * 0.05).to(torch.bfloat16)
    lin.weight.data = wl.clone()
    xl = torch.randn(4, 256, generator=g).to(torch.bfloat16)
    out["module/w"] = bits16(wl)
    out["module/x"] = bits16(xl)
    ql = Q.QGroupLinear.from_linear(lin, torch.int8, 128)
    out["module/qgroup_int8_g128/y"] = bits16(ql(xl))
    qt = Q.QTensorLinear.from_linear(lin, torch.float8_e4m3fn)
    out["module/qtensor_fp8/y"] = bits16(qt(xl))

    # ---- weight-loader sharding (linear.py:65-70, 87-93, 114-128, 142-147) with a faked TP world
    import torch.distributed as tdist
    real_rank, real_ws = tdist.get_rank, tdist.get_world_size
    hidden, head, nh, nkv, inter = 32, 8, 8, 4, 48
    def _ar(tag, r, c):  # distinct integer per (tensor, row, col): sharding errors cannot cancel
        return (tag * 100000 + torch.arange(r * c).reshape(r, c)).float()
    full = {
        "q": _ar(1, nh * head, hidden),
        "k": _ar(2, nkv * head, hidden),
        "v": _ar(3, nkv * head, hidden),
        "o": _ar(4, hidden, nh * head),
        "gate": _ar(5, inter, hidden),
        "up": _ar(6, inter, hidden),
        "down": _ar(7, hidden, inter),
    }
    for k_, v_ in full.items():
        out[f"tp/full/{k_}"] = v_.numpy().astype(np.int32)
    try:
        for tp in (1, 2, 4):
            for r in range(tp):
                tdist.get_rank = lambda r=r: r
                tdist.get_world_size = lambda tp=tp: tp
                L.dist.get_rank = tdist.get_rank
                L.dist.get_world_size = tdist.get_world_size
                qkv = L.QKVParallelLinear(hidden, head, nh, nkv)
                for sid in "qkv":
                    qkv.weight_loader(qkv.weight, full[sid], sid)
                gu = L.MergedColumnParallelLinear(hidden, [inter, inter])
                gu.weight_loader(gu.weight, full["gate"], 0)
                gu.weight_loader(gu.weight, full["up"], 1)
                o = L.RowParallelLinear(nh * head, hidden)
                o.weight_loader(o.weight, full["o"])
                dn_ = L.RowParallelLinear(inter, hidden)
                dn_.weight_loader(dn_.weight, full["down"])
                for nm, mod in (("qkv", qkv), ("gate_up", gu), ("o", o), ("down", dn_)):
                    out[f"tp/{tp}/{r}/{nm}"] = mod.weight.data.float().numpy().astype(np.int32)
    finally:
        tdist.get_rank, tdist.get_world_size = real_rank, real_ws
        L.dist.get_rank, L.dist.get_world_size = real_rank, real_ws

    out["meta/torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, "quant_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
