import torch, sys
sys.path.insert(0,'nano-vllm-hw3_amd')
from nanovllm import ops
x=torch.randn(16,2048,device='cuda',dtype=torch.bfloat16)
def run(n):
    for _ in range(n): ops.silu_mul(x)
run(3); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run(200)
g.replay(); torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print("tiny kernel in graph: us/kernel", s.elapsed_time(e)*1e3/200)
