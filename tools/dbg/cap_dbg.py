import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29717")
dist.init_process_group("nccl", rank=0, world_size=1)
from nanovllm.layers.linear import tp_all_reduce
which = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "global"
x = torch.randn(16, 256, device="cuda", dtype=torch.bfloat16)
v = torch.randn(16, device="cuda")
def ar():
    y = x * 2
    tp_all_reduce(y)
    return y
def ag():
    out = torch.empty((1, 16), device="cuda")
    dist.all_gather_into_tensor(out, v.contiguous())
    return out
def ag_list():
    parts = [torch.empty_like(v)]
    dist.all_gather(parts, v)
    return parts[0]
def plain_ar():
    y = x * 2
    dist.all_reduce(y)
    return y
fn = dict(ar=ar, ag=ag, ag_list=ag_list, plain_ar=plain_ar)[which]
print("eager", which, flush=True)
fn(); torch.cuda.synchronize()
print("capture", which, mode, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=mode):
    r = fn()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed ok", float(r.float().sum()), flush=True)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
gs = []
for i in range(n):
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, g.pool(), capture_error_mode=mode):
        r = fn(); r = fn()
    gs.append(g2)
for g2 in gs: g2.replay()
torch.cuda.synchronize()
print("many ok", flush=True)
dist.destroy_process_group()
print("destroyed", flush=True)
