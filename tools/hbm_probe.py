"""Dev tool: what HBM bandwidth do simple torch kernels reach on this box? (copy, read-only sum, fill)"""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 30            # 2 GiB of bf16
x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
b = x.numel() * 2
dt = t(lambda: y.copy_(x)); print(f"copy   : {2*b/dt/1e12:.2f} TB/s (read+write), {dt*1e3:.3f} ms")
dt = t(lambda: x.view(torch.int16).sum(dtype=torch.int64)); print(f"sum    : {b/dt/1e12:.2f} TB/s (read)")
dt = t(lambda: y.zero_()); print(f"fill   : {b/dt/1e12:.2f} TB/s (write)")
xf = x.view(torch.float32); yf = y.view(torch.float32)
dt = t(lambda: torch.add(xf, 1.0, out=yf)); print(f"add f32: {2*b/dt/1e12:.2f} TB/s (read+write)")
