"""Dev tool: time the fused weight-gradient launch alone (time layer 1 shape by default) and, with a library built
with -DDJ_STAMPS (tools/build_variant.sh stamps -DDJ_STAMPS; DEEPJ_LIB=...), print the per-wave cycle stamps."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_generator_amd import _lib as L

def main(M=1 << 20, steps=128, DP=256, D=256, H=256, N=1024, reps=10):
    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(M, DP, device=dev, generator=g).bfloat16()
    Hs = torch.randn(M, H, device=dev, generator=g).bfloat16()
    dZ = (0.1 * torch.randn(N // 256, M, 256, device=dev, generator=g)).bfloat16()
    if os.environ.get("WG_ZERO"):      # zero operands: same instruction stream, less switching power (DVFS check)
        X.zero_(); Hs.zero_(); dZ.zero_()
    dW = torch.zeros(D, N, device=dev); dU = torch.zeros(H, N, device=dev)
    zeros = torch.zeros(64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        L.check(lib.dj_lstm_wgrad(1, M, steps, L.ptr(X), DP, D, L.ptr(Hs), H, L.ptr(dZ), N, M * 256, L.ptr(dW), L.ptr(dU),
                                  L.ptr(zeros), st), "wgrad")
    run(); torch.cuda.synchronize()
    dbg = getattr(lib, "dj_debug_wg_stamps", None) if hasattr(lib, "dj_debug_wg_stamps") else None
    buf = (ctypes.c_ulonglong * 64)()
    if dbg is not None:
        dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]; dbg(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * (D + H) * N
    print(f"wgrad M={M} DP={DP} H={H} N={N}: {ms:.3f} ms, {fl / ms / 1e9:.0f} TFLOP/s  (DEEPJ_WGRAD_PP={os.environ.get('DEEPJ_WGRAD_PP', '1')})")
    if dbg is not None:
        dbg(buf, 0)
        nkt = max(1, (M // 32) // 32) * reps    # stages per workgroup (32 row splits)
        names = ["-", "-", "-", "read+dma(+vmcnt g1)+lgkm", "barA", "mfma", "vmcnt g0", "barB"]
        for w in range(8):
            row = [buf[w * 8 + i] / nkt for i in range(8)]
            print(f"wave {w}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, row)) + f"   total {sum(row):6.0f}")

if __name__ == "__main__":
    main(M=int(os.environ.get("WG_M", 1 << 20)), reps=int(os.environ.get("WG_REPS", 10)))
