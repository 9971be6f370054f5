"""What a plain streaming read reaches on this box: torch reductions and a copy over 2 GiB (diagnostic)."""
import torch
n = 1 << 29                      # 2 GiB of int32
x = torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps
for name, fn, byts in [("int32 sum", lambda: x.sum(), 4 * n), ("int32 max", lambda: x.max(), 4 * n),
                       ("bf16 view max", lambda: x.view(torch.bfloat16).max(), 4 * n),
                       ("copy", lambda: y.copy_(x), 8 * n)]:
    dt = t(fn)
    print(f"{name:14s} {byts / dt / 1e12:5.2f} TB/s", flush=True)
