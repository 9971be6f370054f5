import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from nanovllm.utils import quantization as NQ
from oracle import qlinear_oracle as O
for xdt in (torch.float16, torch.float32):
    x = torch.arange(1, 129, dtype=torch.float32).reshape(1, 128).to(xdt).cuda()
    q, s = NQ.per_row_quant(x, torch.int8)
    print(xdt, q[0, :24].tolist(), float(s))
    qo, so = O.per_row_quant(x.float().cpu().numpy(), "int8")
    print("   oracle", qo[0, :24].tolist(), float(so))
