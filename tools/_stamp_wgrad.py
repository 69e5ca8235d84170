import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from music_generator_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for (M, DP, D, H, N, steps, tiled, name) in [(1 << 20, 256, 256, 256, 1024, 128, 1, "time L1 (HBM streaming)"), (16384, 1024, 1024, 1024, 4096, 8, 0, "C4-like (L2 resident)")]:
    X = torch.randn(M, DP, device=dev).to(torch.bfloat16); Hs = torch.randn(M, H, device=dev).to(torch.bfloat16)
    dZ = (torch.randn(M, N, device=dev) * 0.1).to(torch.bfloat16)
    dW = torch.zeros(D, N, device=dev); dU = torch.zeros(H, N, device=dev); z = torch.zeros(64, device=dev)
    for _ in range(3):
        _lib.check(lib.dj_lstm_wgrad(1, M, steps, p(X), DP, D, p(Hs), H, p(dZ), N, M * 256 if tiled else 0, p(dW), p(dU), p(z), st()), "wgrad")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.dj_lstm_wgrad(1, M, steps, p(X), DP, D, p(Hs), H, p(dZ), N, M * 256 if tiled else 0, p(dW), p(dU), p(z), st())
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    fl = 2.0 * M * (DP + H) * N
    if not hasattr(lib, "dj_debug_read"):
        print(name, f"{ms:.3f} ms {fl/ms/1e9:.0f} TFLOP/s", flush=True)
        continue
    buf = np.zeros(8192, np.uint64)
    lib.dj_debug_read.restype = C.c_int
    assert lib.dj_debug_read(C.c_void_p(buf.ctypes.data)) == 0
    s = buf[:32 * 8].reshape(32, 8).astype(np.int64)
    d = np.diff(s[:, :7], axis=1)
    per = np.diff(s[:, 0])
    print(name, f"{ms:.3f} ms {fl/ms/1e9:.0f} TFLOP/s; stage period median {np.median(per):.0f} ticks; phases (vmcnt wait, barrier, reads0, mfma0+dma, reads1, mfma1+dma) median", [float(np.median(d[:, i])) for i in range(6)], flush=True)
