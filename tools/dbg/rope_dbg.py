import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
from nanovllm.utils import quantization as NQ
from tests.test_gpu_parity import _rope_setup, _rope_both_ways
for M in (130, 192, 224, 256, 288, 512):
    Hq, Hkv, K = 16, 8, 1024
    x, w, bias, qn, kn, cs, pos, slots = _rope_setup(M, Hq, Hkv, K, M)
    mod = NQ.QLinear(torch.int8, w, None).tile()
    for rep in range(2):
        a, b = _rope_both_ways(NQ, mod, x, qn, kn, cs, pos, slots, Hq, Hkv, False)
        for nm, ta, tb in zip(("q", "k", "v", "kc", "vc"), a, b):
            if ta is None: continue
            d = (ta.float() - tb.float()).abs()
            nz = (d > 0).nonzero()
            print(M, rep, nm, "ndiff", nz.shape[0], "max", d.max().item(), "first", nz[:3].tolist(), flush=True)
