"""Quick single-GPU timing of the training step (dev tool; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from music_generator_amd.engine import DeepJConfig, Engine, Nadam, param_count, init_params_numpy

def run(dtype, B=64, T=128, N=128, steps=3, pin=0.2, pdr=0.5, micro=1, **kw):
    dev = torch.device("cuda:0")
    cfg = DeepJConfig(num_notes=N, time_steps=T, dtype=dtype, **kw)
    eng = Engine(cfg, B, T, device=dev, input_dropout=pin, dropout=pdr)
    P = torch.from_numpy(init_params_numpy(cfg)).to(dev)
    G = torch.zeros_like(P)
    opt = Nadam(P.numel(), dev)
    g = torch.Generator(device=dev).manual_seed(0)
    play = (torch.rand(B, T + 1, N, device=dev, generator=g) < 0.05).float()
    rep = play * (torch.rand(B, T + 1, N, device=dev, generator=g) < 0.2).float()
    vol = play * (0.2 + 0.8 * torch.rand(B, T + 1, N, device=dev, generator=g))
    roll = torch.stack([play, rep, vol], -1)
    notes, target = roll[:, :T].contiguous(), roll[:, 1:].contiguous()
    beat = torch.zeros(B, T, 16, device=dev); beat[:, torch.arange(T), torch.arange(T) % 16] = 1
    style = torch.zeros(B, T, 23, device=dev); style[torch.arange(B), :, torch.arange(B) % 23] = 1
    print(f"[{dtype}] workspace {eng.ws_bytes/2**30:.2f} GiB", flush=True)
    losses = []
    for it in range(steps + 1):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.time()
        for m in range(micro):                      # micro-batches: gradients accumulate, one optimizer step
            loss = eng.train_fwd_bwd(P, G, notes, target, beat, style, target, seed=it * micro + m, accumulate=m > 0)
        opt.step(P, G, grad_scale=1.0 / micro)
        losses.append(loss.clone())
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print(f"[{dtype}] B={B}x{micro} T={T} N={N}: {dt*1e3:.2f} ms/step, {micro*B*T*N/dt/1e6:.2f} M note-steps/s, "
          f"losses {[round(float(l[0]), 5) for l in losses]}", flush=True)
    return eng

if __name__ == "__main__":
    if sys.argv[1:2] == ["scaled"]:      # BASELINE configs[4] family: 3 x 1024 units per axis; B, T from argv
        B, T = int(sys.argv[2]), int(sys.argv[3])
        micro = int(sys.argv[4]) if len(sys.argv) > 4 else 1
        run("bf16", B=B, T=T, N=128, steps=2, micro=micro, time_axis_units=1024, note_axis_units=1024, time_axis_layers=3,
            note_axis_layers=3)
        sys.exit(0)
    for dt in sys.argv[1:] or ["f32", "bf16"]:
        run(dt)
