"""Golden vectors for the Triton half of the reference's quantized-linear path, produced by the
reference's OWN ``@triton.jit`` kernels running under Triton's CPU interpreter.

Run ONLY in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_triton.py

``TRITON_INTERPRET=1`` (set below, before triton is imported) makes ``kernel[grid](...)`` execute the
kernel body with numpy on CPU tensors, so no GPU driver is needed.  What is stored in
``tests/golden/triton_golden.npz``:

* ``per_row_quant(t, float8_e4m3fn)`` (``_per_row_quant_kernel``, quantization.py:13-51, launched as
  :147-163) on weights (incl. an all-zero row) and on activations: q bytes + f32 scales;
* ``per_row_matmul(x, qw, sw, None, float8_e4m3fn)`` end to end (quantization.py:310-346): activation
  quantization + ``_per_row_post_process_mm`` (:54-144) on fp8 operands;
* ``_per_row_post_process_mm`` launched exactly as ``per_row_matmul`` launches it (:324-345) on INT8
  operands.  The int8 branch of ``_per_row_quant_kernel`` calls ``libdevice.round`` (:41), which the
  interpreter cannot execute (it raises ``InterpreterError: cannot convert None ... to tensor``), so the
  int8 operands fed to the GEMM are quantized by this repository's oracle and stored with the outputs.
  That one call - roundf on ``x * (127 / row_max)`` - is the only arithmetic of the path with no
  reference-produced vector; the division / multiplication around it is the same code as the fp8 branch.
"""
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"          # must precede `import triton`
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

import make_golden as MG  # noqa: E402  (stand-in modules, bit helpers, CASES)

# the per-tensor / per-group cases plus shapes that cross the kernel's 128-row / 128-column tile edge
CASES = MG.CASES + [
    ("m130_n200_k384", 130, 200, 384, 0.05),
    ("m7_n130_k128", 7, 130, 128, 1.0),
]


def main():
    sys.path.insert(0, REF)
    MG._stub_modules()
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29613", world_size=1, rank=0)
    import triton
    from nanovllm.utils import quantization as Q
    from oracle import qlinear_oracle as O

    fp8 = torch.float8_e4m3fn
    out = {}
    for ci, (name, M, N, K, wstd) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + ci)
        x = torch.randn(M, K, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * wstd).to(torch.bfloat16)
        w[0, :] = 0                               # all-zero row: the 1e-8 floor of row_max
        if N > 2:
            w[1, 0] = w[1].abs().max()
        if M > 2:
            x[2, :] = 0                           # an all-zero activation row
        out[f"{name}/x"] = MG.bits16(x)
        out[f"{name}/w"] = MG.bits16(w)

        # ---- fp8: everything from the reference's kernels
        qw, sw = Q.per_row_quant(w, fp8)
        qx, sx = Q.per_row_quant(x, fp8)
        out[f"{name}/fp8/qw"], out[f"{name}/fp8/sw"] = MG.bits8(qw), sw.numpy().astype(np.float32)
        out[f"{name}/fp8/qx"], out[f"{name}/fp8/sx"] = MG.bits8(qx), sx.numpy().astype(np.float32)
        out[f"{name}/fp8/y"] = MG.bits16(Q.per_row_matmul(x, qw, sw, None, fp8))

        # ---- int8: operands from the oracle (libdevice.round is not interpretable), GEMM from the reference
        qw_o, sw_o = O.per_row_quant(MG.bits16(w), "int8")
        qx_o, sx_o = O.per_row_quant(MG.bits16(x), "int8")
        out[f"{name}/int8/qw"], out[f"{name}/int8/sw"] = qw_o.view(np.uint8), sw_o
        out[f"{name}/int8/qx"], out[f"{name}/int8/sx"] = qx_o.view(np.uint8), sx_o
        tqx, tqw = torch.from_numpy(qx_o.copy()), torch.from_numpy(qw_o.copy())
        tsx, tsw = torch.from_numpy(sx_o.copy()), torch.from_numpy(sw_o.copy())
        y = torch.empty((M, N), dtype=torch.bfloat16)
        grid = (triton.cdiv(N, 128), triton.cdiv(M, 128))       # quantization.py:330
        Q._per_row_post_process_mm[grid](tqx, tsx, tqw, tsw, None, y, M, N, K,
                                         BLK_M=128, BLK_N=128, BLK_K=128, FP8=False)   # :324, :331-345
        out[f"{name}/int8/y"] = MG.bits16(y)
        print(name, "done", flush=True)

    out["meta/torch_version"] = np.array(torch.__version__)
    out["meta/triton_version"] = np.array(triton.__version__)
    path = os.path.join(HERE, "triton_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
