#!/usr/bin/env python3
"""Headline benchmark: DeepJ biaxial-LSTM training throughput in note-steps/sec.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full training step (teacher-forced forward + BPTT + Nadam, dropout on,
plus the RCCL gradient all-reduce when N > 1) over one synthetic batch of
B=64 x T=128 x N=128 per GPU (BASELINE.json configs[1]; weak scaling: global batch
64*N).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line.  With --gpus N > 1 and no launcher above it (no RANK in the environment) the process
starts its own ranks: `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
as a child, before any GPU call; output and exit code pass through.
`roofline` is measured live with HIP events recorded by the library around every launch
(dj_profile_*); `roofline.traffic` comes from rocprofv3 PMC passes of the same build when
--pmc-dir names them (tools/profile_round.sh), else from the committed passes (labelled);
`cpu_baseline` times the CPU oracle on a bounded sample (1 warm-up + 3 timed steps on 8 of the
64 sequences); `fp32_parity_mode` is the same step in the fp32 mode the 1e-3 parity gate runs
in; `generation` is BASELINE configs[3] (3 pieces, 1024 steps; bf16 and the certified fp32
mode); `scaled` is BASELINE configs[4] (3 x 1024 units per axis, batch 128 x 256 x 128 as two
micro-batches, 1 warm-up + 2 timed steps) with its own roofline.  `--config scaled` times
configs[4] alone.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                              # HBM3E, same guide


def _fused_xw(cfg, d, H):
    """Mirror of dj_api.hip fuse_xw at the bench shape (>= 128 sequence tiles): is x*W of a layer with input
    width d and H units computed inside the recurrent forward kernel?"""
    if H not in (128, 256):      # generic width (scaled model): bf16 folds x*W into every step's GEMM (dj_step.hip, round 4)
        return cfg.dtype == "bf16" and os.environ.get("DEEPJ_STEP_EPILOGUE", "1") != "0"
    return d <= (288 if (H == 128 and cfg.dtype == "bf16") else 2 * H)


def _fused_dx(cfg, d, H):
    """Mirror of dj_api.hip fuse_dx == 1: is the whole dX = dz W^T of this layer produced inside the BPTT
    kernel?  (Mode 2 -- only the last 3 columns of note layer 0 -- moves a negligible share and stays under
    gemm_dx in this accounting.)"""
    return cfg.dtype == "bf16" and H == 128 and d <= H


def category_flops(cfg, B, T, N):
    """ALGORITHMIC FLOPs (2*MAC) per training step of each MFMA kernel category
    (SURVEY.md 8d formulas, generalised to the config)."""
    Ht, Hn = cfg.time_axis_units, cfg.note_axis_units
    rows = B * T * N
    F = 1 + cfg.octave + 1 + cfg.octave_units + cfg.notes_per_bar
    t_in = [F] + [Ht] * (cfg.time_axis_layers - 1)
    n_in = [Ht + cfg.note_units] + [Hn] * (cfg.note_axis_layers - 1)
    xw_t, xw_n = sum(2 * rows * d * 4 * Ht for d in t_in), sum(2 * rows * d * 4 * Hn for d in n_in)
    xw = xw_t + xw_n
    rec_t = cfg.time_axis_layers * 2 * rows * Ht * 4 * Ht
    rec_n = cfg.note_axis_layers * 2 * rows * Hn * 4 * Hn
    # the forward recurrent kernel carries the input projection x*W of every layer with D <= 2H
    # (dj_api.hip fuse_xw); the others keep a separate GEMM launch (category gemm_xw)
    fx_t = sum(2 * rows * d * 4 * Ht for d in t_in if _fused_xw(cfg, d, Ht))
    fx_n = sum(2 * rows * d * 4 * Hn for d in n_in if _fused_xw(cfg, d, Hn))
    dx_t = sum(2 * rows * d * 4 * Ht for d in t_in if _fused_dx(cfg, d, Ht))
    dx_n = sum(2 * rows * d * 4 * Hn for d in n_in if _fused_dx(cfg, d, Hn))
    return {
        "gemm_xw": xw - fx_t - fx_n, "gemm_dx": xw - dx_t - dx_n, "gemm_dw": xw + rec_t + rec_n,
        "lstm_fwd_time": rec_t + fx_t, "lstm_bwd_time": rec_t + dx_t, "lstm_fwd_note": rec_n + fx_n,
        "lstm_bwd_note": rec_n + dx_n,
    }


def category_bytes(cfg, B, T, N, esize):
    """ALGORITHMIC HBM bytes per training step of the same categories: activations only (weights
    stay in L2), esize bytes per element, gate stash at s bytes per element (bf16 mode: 8-bit activated gates,
    s = 1; fp32 mode: z, s = 4).  Per row of an LSTM layer with input width D and H units:
    forward reads x (D) and writes the gate stash (4H at s), h (H), c (H); the unfused variant reads
    x W + b (4H) instead of x; backward reads the gate stash (4H at s), c (H), dh (H) and writes dz (4H);
    dW/dU read dz (4H), x (D), h (H); dX reads dz (4H) and writes dx (D)."""
    Ht, Hn = cfg.time_axis_units, cfg.note_axis_units
    rows = B * T * N
    F = 1 + cfg.octave + 1 + cfg.octave_units + cfg.notes_per_bar
    t_in = [F] + [Ht] * (cfg.time_axis_layers - 1)
    n_in = [Ht + cfg.note_units] + [Hn] * (cfg.note_axis_layers - 1)
    out = {k: 0 for k in ("gemm_xw", "gemm_dx", "gemm_dw", "lstm_fwd_time", "lstm_bwd_time", "lstm_fwd_note",
                          "lstm_bwd_note")}
    for axis, H, dims in (("time", Ht, t_in), ("note", Hn, n_in)):
        # 8-bit activated-gate codes: the persistent bf16 kernels (dj_lstm.hip GateEnc) and, since round 4, the generic-width
        # path with the cell epilogue (dj_common.h dj_gate_code*); fp32 keeps z
        coded = esize == 2 and (H in (128, 256) or _fused_xw(cfg, 1, H))
        s = 1 if coded else (4 if H in (128, 256) else esize)
        for d in dims:
            fused = _fused_xw(cfg, d, H)
            out["lstm_fwd_" + axis] += rows * (esize * ((d + 2 * H) if fused else 6 * H) + s * 4 * H)
            if not fused:
                out["gemm_xw"] += rows * esize * (d + 4 * H)
            out["lstm_bwd_" + axis] += rows * (esize * 6 * H + s * 4 * H)
            out["gemm_dw"] += rows * esize * (5 * H + d)
            if _fused_dx(cfg, d, H):
                out["lstm_bwd_" + axis] += rows * esize * d          # dx written by the BPTT kernel itself
            else:
                out["gemm_dx"] += rows * esize * (4 * H + d)
    return out


def stash_bytes_per_note_step(cfg, esize=2):
    """minimal BPTT stash per note-step if gates are kept -- sum over layers of 6H, plus the feature
    row F and the 64 conv outputs, written once and read once: (sum 6H + F + 64) x esize x 2 = 19,064 B for the
    reference model in bf16."""
    F = 1 + cfg.octave + 1 + cfg.octave_units + cfg.notes_per_bar
    elems = 6 * (cfg.time_axis_layers * cfg.time_axis_units + cfg.note_axis_layers * cfg.note_axis_units) + F + cfg.octave_units
    return elems * esize * 2


def launches_per_step(cfg):
    Lt, Ln = cfg.time_axis_layers, cfg.note_axis_layers
    F = 1 + cfg.octave + 1 + cfg.octave_units + cfg.notes_per_bar
    dims = [(d, cfg.time_axis_units) for d in [F] + [cfg.time_axis_units] * (Lt - 1)]
    dims += [(d, cfg.note_axis_units) for d in [cfg.time_axis_units + cfg.note_units] + [cfg.note_axis_units] * (Ln - 1)]
    unfused = sum(not _fused_xw(cfg, d, H) for d, H in dims)
    return {"gemm_xw": max(1, unfused), "gemm_dx": Lt + Ln, "gemm_dw": Lt + Ln, "lstm_fwd_time": Lt,
            "lstm_bwd_time": Lt, "lstm_fwd_note": Ln, "lstm_bwd_note": Ln}


def cpu_share():
    """CPU cores this process may really use: affinity mask capped by the cgroup quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, T, N, sample_b, pin, pdr, batch, warmup=1, steps=3):
    """CPU oracle (torch-CPU fp32 restatement of the reference: "port") on a bounded sample of the same
    workload: the first sample_b sequences of the `batch`-sequence synthetic batch, `warmup` untimed +
    `steps` timed training steps (forward + BPTT + Nadam, dropout on) at the full T and N.  note-steps/s of
    the oracle does not depend on the batch size in this range (3.3 k/s at 2 sequences, 3.7 k/s at 8 on 8
    cores), so the sample rate is the rate of the full batch."""
    from oracle import deepj_oracle as O
    from music_generator_amd.data import synthetic_batch
    ocfg = O.OracleConfig(num_notes=N, time_steps=T, time_axis_units=cfg.time_axis_units,
                          note_axis_units=cfg.note_axis_units, time_axis_layers=cfg.time_axis_layers,
                          note_axis_layers=cfg.note_axis_layers)
    cores = cpu_share()
    torch.set_num_threads(cores)
    params = O.init_params(ocfg, seed=1234)
    data = [a[:sample_b] for a in synthetic_batch(N, T, batch, seed=0)]
    st = O.NadamState()
    flat = O.flatten_params(ocfg, params)
    times, loss = [], None
    for i in range(warmup + steps):
        masks = O.make_masks(ocfg, sample_b, i, pin, pdr, T=T)
        t0 = time.time()
        loss, _, grads = O.loss_and_grads(ocfg, O.unflatten_params(ocfg, flat), data, masks)
        flat = O.nadam_step(flat, O.flatten_params(ocfg, grads), st)
        times.append(time.time() - t0)
    dt = sum(times[warmup:]) / steps
    return {"value": sample_b * T * N / dt, "unit": "note-steps/sec", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (torch-CPU fp32, {cores} threads): {warmup} warm-up + {steps} timed train steps "
                      f"(fwd+BPTT+Nadam, dropout on) on {sample_b} of the {batch} sequences at T={T}, N={N}: "
                      f"{dt:.1f} s per step; rate taken as the full batch's (note-steps/s is flat in the batch size)",
            "sample_sequences": sample_b, "warmup_steps": warmup, "timed_steps": steps,
            "s_per_step": [round(t, 2) for t in times], "loss": loss}


def make_step(eng, opt, P, G, batch, world, rank, dist, force_collective=False):
    """THE timed step: forward + BPTT, one all-reduce(sum) of the flat fp32 gradient over RCCL/xGMI when
    world > 1, Nadam with the 1/world scale folded in.  Factored out so that the 2-rank gloo test
    (tests/test_bench_cpu.py) and the world-size-1 RCCL test on the GPU (tests/test_dist_gpu.py, force_collective)
    run exactly this function."""
    notes, chosen, beat, style, target = batch

    def step(i):
        loss = eng.train_fwd_bwd(P, G, notes, chosen, beat, style, target, seed=i * world + rank)
        if world > 1 or force_collective:
            dist.all_reduce(G)                           # one flat fp32 buffer
        opt.step(P, G, grad_scale=1.0 / world)
        return loss
    return step


def fp32_parity_mode(cfg_kw, B, T, N, pin, pdr, dev, rank, steps=3, warmup=1):
    """The same training step in the fp32 mode the 1e-3 parity tests run in (v_mfma_f32_32x32x2_f32)."""
    from music_generator_amd.data import synthetic_batch
    from music_generator_amd.engine import DeepJConfig, Engine, Nadam, init_params_numpy
    cfg = DeepJConfig(dtype="f32", **cfg_kw)
    eng = Engine(cfg, B, T, device=dev, input_dropout=pin, dropout=pdr)
    P = torch.from_numpy(init_params_numpy(cfg, seed=1234)).to(dev)
    G = torch.zeros_like(P)
    step = make_step(eng, Nadam(P.numel(), dev), P, G,
                     [torch.from_numpy(a).to(dev) for a in synthetic_batch(N, T, B, seed=rank)], 1, 0, None)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"ms_per_step": round(dt * 1e3, 3), "value": round(B * T * N / dt, 1), "unit": "note-steps/sec",
           "steps": steps, "warmup": warmup, "dtype": "f32", "final_loss": round(float(loss.cpu()[0]), 5)}
    del eng, P, G
    torch.cuda.empty_cache()
    return out


def generation_bench(dtype, steps):
    """Secondary metric (BASELINE configs[3]): autoregressive sampling, 3 genre style vectors,
    N=48 notes, fused dj_generate_step per time step (time-axis window + incremental note axis)."""
    import contextlib
    import io
    from music_generator_amd import generate as Gn
    from music_generator_amd.dataset import compute_genre
    from music_generator_amd.model import build_models
    models = build_models(dtype=dtype, seed=5)
    styles = [compute_genre(i) for i in range(3)]
    np.random.seed(0)
    chunk = Gn.GEN_CHUNK                        # steps per device batch of the resident path
    steps = max(chunk, (steps + chunk - 1) // chunk * chunk)
    bars = (steps + chunk + 15) // 16
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        g = Gn.generate(models, bars, styles)
        for _ in range(chunk):                    # first chunk (graph capture) is warm-up
            next(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in g:                               # whole chunks only: yields == computed steps
            n += 1
            if n >= steps:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for _ in g:                               # let the generator finish: it reads the near-tie census at its end
            pass
    # ALGORITHMIC HBM bytes of one generated time step: the stateless 128-step window through the time axis
    # (generate.py:106-109) plus one pass over the fp32 weights; the note loop runs out of L2.  Per row and layer:
    # bf16 (x W fused into the weight-stationary cluster sweep): x in, h out = (D + H) elements; fp32 (separate
    # GEMM): x in, x W + b out and back in, h out = (D + 9 H); layer 1's input is a glue copy of layer 0's h: + 2 H
    K = Gn.__dict__
    rows = 3 * K["NUM_NOTES"] * K["SEQ_LEN"]
    F, Ht, e = 94, K["TIME_AXIS_UNITS"], (2 if dtype == "bf16" else 4)
    zx = 0 if dtype == "bf16" else 8 * Ht
    gen_bytes = rows * e * ((F + Ht + zx) + (Ht + Ht + zx) + 2 * Ht) + 1269476 * 4
    return {"metric": "gen notes/sec", "value": round(3 * 48 * n / dt, 1), "unit": "notes/sec",
            "ms_per_time_step": round(dt / n * 1e3, 3), "pieces": 3, "notes": 48, "window": 128, "steps": n,
            "dtype": dtype, "hbm_gbs": round(gen_bytes / (dt / n) / 1e9, 1), "hbm_bytes_per_step": gen_bytes,
            "bound": "latency (M = 144 rows, 128 + 48 dependent steps per generated step)",
            "near_tie_draws": Gn.last_run_stats["near_ties"], "draws": Gn.last_run_stats["draws"],
            "path": "dj_generate_prepare + dj_generate_step_prepared (hipGraph replay), NumPy MT19937 draws in reference order"}


def dominant_category(ms, keys):
    """The category the roofline is reported for: the largest one -- but `gemm_dw` / `gemm_dx` are 3-4 launches of
    DIFFERENT shapes each, for which "bytes per launch / average launch duration" is an average over unlike things, and
    since round 3 `gemm_dw` (4 launches) and `lstm_bwd_time` (2 launches of one shape) are within 1-2 % of each other:
    a recurrent-sweep category (one kernel, one shape per layer pair) within 5 % of the largest is preferred, so that
    the line names the same kernel from run to run."""
    top = max(keys, key=lambda k: ms.get(k, 0.0))
    sweeps = [k for k in keys if k.startswith("lstm_") and ms.get(k, 0.0) >= 0.95 * ms.get(top, 0.0)]
    return max(sweeps, key=lambda k: ms[k]) if sweeps else top


def read_profile(lib, steps):
    """Per-category milliseconds per step from the library's HIP-event records (dj_profile_read)."""
    from music_generator_amd import _lib
    out = {}
    for c in range(lib.dj_profile_category_count()):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(lib.dj_profile_read(c, C.byref(ms), C.byref(n)), "dj_profile_read")
        out[lib.dj_profile_category_name(c).decode()] = round(ms.value / steps, 4)
    return out


def scaled_record(dtype, micro, steps, warmup, dropout, dev, rank, world, dist, force_collective=False, shape=None,
                  return_params=False, pmc_dir=None):
    """BASELINE configs[4]: 3 x 1024 units per axis, batch 128 x 256 steps x 128 notes per GPU.  The batch runs as
    `micro` equal micro-batches through one workspace with gradient accumulation and ONE optimizer step -- EXACTLY the
    step on the batch of 128 (dj_train_fwd_bwd_mb): the pitch_bins table is that of the whole batch (dj_pitch_bins, one
    small launch per step inside the timed region) and every dropout mask is the full batch's mask of the micro-batch's
    rows.  Returns the JSON record (rank 0) or None."""
    from music_generator_amd import _lib
    from music_generator_amd.data import synthetic_batch
    from music_generator_amd.engine import DeepJConfig, Engine, Nadam, init_params_numpy
    B, T, N = shape or (128, 256, 128)              # (tests rehearse the same function at a small shape)
    pin, pdr = dropout or (0.2, 0.5)
    cfg = DeepJConfig(num_notes=N, time_steps=T, dtype=dtype, time_axis_units=1024, note_axis_units=1024,
                      time_axis_layers=3, note_axis_layers=3)
    eng = Engine(cfg, B // micro, T, device=dev, input_dropout=pin, dropout=pdr)
    P = torch.from_numpy(init_params_numpy(cfg, seed=1234)).to(dev)
    G = torch.zeros_like(P)
    opt = Nadam(P.numel(), dev)
    data = [torch.from_numpy(a).to(dev) for a in synthetic_batch(N, T, B, seed=rank)]
    k = B // micro
    parts = [[a[m * k:(m + 1) * k].contiguous() for a in data] for m in range(micro)]
    lib = _lib.load()

    def step(i):
        seed = i * world + rank
        bins = eng.pitch_bins(data[0], seed=seed)
        for m in range(micro):
            loss = eng.train_fwd_bwd(P, G, *parts[m], seed=seed, accumulate=m > 0, full_batch=B, batch_offset=m * k,
                                     bins_full=bins)
        if world > 1 or force_collective:
            dist.all_reduce(G)                           # 187.6 MB of fp32 gradient
        opt.step(P, G, grad_scale=1.0 / (micro * world))
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss.cpu()[0])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
    if not np.isfinite(final_loss):
        raise SystemExit(f"invalid run: final loss {final_loss}")
    # the per-category table from ONE more, untimed step: events around each of its ~9,000 launches would cost the
    # timed steps 1-2 %
    lib.dj_profile_enable(1)
    step(warmup + steps)
    torch.cuda.synchronize()
    kernels = {k_: round(v_, 3) for k_, v_ in read_profile(lib, 1).items()}
    lib.dj_profile_enable(0)
    ws_gib = round(eng.ws_bytes / 2 ** 30, 1)
    faults = eng.cluster_faults("scaled bench record")
    params_out = P.cpu().numpy() if return_params else None
    del eng, P, G, opt, data, parts
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    fl = category_flops(cfg, B, T, N)
    flops_step = 3 * (fl["gemm_xw"] + fl["lstm_fwd_time"] + fl["lstm_fwd_note"])
    tf = flops_step * world * steps / elapsed / 1e12
    dom = max(fl, key=lambda c_: kernels.get(c_, 0.0))
    dom_tf = fl[dom] / (kernels[dom] * 1e-3) / 1e12 if kernels.get(dom) else 0.0
    rates = {k_: round(fl[k_] / (kernels[k_] * 1e-3) / 1e12, 1) for k_ in fl if kernels.get(k_)}
    # HBM bytes per launch of the dominant category's kernel from rocprofv3 PMC passes of THIS build (tools/profile_scaled.sh)
    traffic = tsrc = None
    try:        # committed passes of an earlier run of this shape (labelled as such), so that the driver's line is not null
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_scaled.json")))
        if dtype == "bf16" and (B, T, N, micro) == (128, 256, 128, 2) and dom in pm:
            traffic = pm[dom]
            tsrc = "profiles/pmc_traffic_scaled.json (committed passes of an earlier run): " + pm["source"]
    except Exception:
        pass
    if pmc_dir:
        live = pmc_traffic_from_dir(pmc_dir, pmc_category)
        if dom in live:
            traffic = live[dom]
            tsrc = ("measured: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes under %s (this build; FETCH_SIZE x 2 per "
                    "MI355X_MICROARCH.md), average bytes per launch of the category's kernel" % pmc_dir)
    by = category_bytes(cfg, B // micro, T, N, 2 if dtype == "bf16" else 4)      # one micro-batch = what a launch sees
    lps = launches_per_step(cfg)
    return {
        "metric": "note-steps/sec (train)", "value": round(world * B * T * N * steps / elapsed, 1),
        "unit": "note-steps/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"scaled biaxial-LSTM train step (BASELINE configs[4]): 3x1024 time-axis + 3x1024 "
                               f"note-axis LSTM, batch {B}/GPU as {micro} exact micro-batches (pitch_bins and dropout masks of the full batch) x {T} steps x {N} notes, "
                               f"dropout {pin}/{pdr}, Nadam; random-init weights",
                   "global_batch": B * world, "seq_len": T, "num_notes": N, "parallelism": f"dp{world}",
                   "micro_batches": micro, "workspace_gib": ws_gib},
        "model_tflops_per_s": round(tf, 1), "final_loss": round(final_loss, 5),
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(dom_tf, 1), "peak": PEAK_TFLOPS[dtype],
                     "unit": "TFLOP/s", "frac": round(dom_tf / PEAK_TFLOPS[dtype], 4), "traffic": traffic,
                     "traffic_source": tsrc, "algorithmic_bytes_per_launch": by[dom] / lps[dom] if lps.get(dom) else None,
                     "whole_step_mfma_frac": round(tf / world / PEAK_TFLOPS[dtype], 4),
                     "flops_per_step": flops_step},
        "kernel_ms_per_step": kernels, "kernel_tflops": rates, "cpu_baseline": None, "cluster_faults": faults,
        **({"params": params_out} if return_params else {})}


def scaled_bench(args, dev, rank, world, dist):
    rec = scaled_record(args.dtype, args.micro, args.steps, args.warmup, args.dropout, dev, rank, world, dist,
                        pmc_dir=args.pmc_dir)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic_from_dir(pmc_dir, category_of):
    """HBM bytes per launch of every kernel category from rocprofv3 PMC passes found under `pmc_dir` (counter_collection
    CSVs of `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of THIS build, separate passes): read = 2 x FETCH_SIZE KB
    (gfx950 tallies 128-byte requests at 64 bytes), write = WRITE_SIZE KB -- MI355X_MICROARCH.md, HBM section."""
    import csv
    import glob
    from collections import defaultdict
    acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
    for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in acc:
                acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = defaultdict(lambda: [0.0, 0])
    for k, v in acc["FETCH_SIZE"].items():
        c = category_of(k)
        if c is None:
            continue
        w = acc["WRITE_SIZE"].get(k)
        per = 2.0 * 1024 * sum(v) / len(v) + (1024 * sum(w) / len(w) if w else 0.0)
        out[c][0] += per * len(v)
        out[c][1] += len(v)
    return {c: tot / n for c, (tot, n) in out.items() if n}


def pmc_category(name):
    """Kernel name -> bench category for the PMC summary (as tools/pmc_summary.py)."""
    if "lstm_bwd256_kernel" in name:                 # bf16 H = 256: the split-gate-math sweep (round 5)
        return "lstm_bwd_time"
    if "lstm_bwd_kernel" in name:
        return "lstm_bwd_time" if "Li256" in name else "lstm_bwd_note"
    if "lstm_fwd_cluster" in name:
        return "lstm_fwd_time"
    if "lstm_fwd" in name:
        return "lstm_fwd_time" if "Li256" in name else "lstm_fwd_note"
    if "lstm_wgrad" in name:
        return "gemm_dw"
    return None


def device_identity(dev):
    """A number that differs between two physical GPUs of this node: the device's UUID where torch exposes it, else
    its PCI address (domain : bus : device)."""
    import hashlib
    p = torch.cuda.get_device_properties(dev)
    ident = getattr(p, "uuid", None)
    ident = str(ident) if ident is not None else ""
    if not ident.strip("0-"):
        ident = "pci %s:%s:%s" % (getattr(p, "pci_domain_id", "?"), getattr(p, "pci_bus_id", "?"),
                                  getattr(p, "pci_device_id", "?"))
    return int.from_bytes(hashlib.sha256(ident.encode()).digest()[:7], "little"), ident


def multi_gpu_evidence(dist, dev, world, elapsed_local, steps):
    """What lets the reader of an N > 1 line check that the collective really spanned N GPUs: the number of
    DISTINCT physical devices among the ranks (all-gathered device identities; N in a real run, 1 in the one-GPU rehearsal)
    and every rank's own time per step before the closing barrier (the timed region ends at the slowest rank)."""
    ident, _ = device_identity(dev)
    mine = torch.tensor([float(ident & 0xFFFFFF), float((ident >> 24) & 0xFFFFFF), float(ident >> 48),
                         elapsed_local / steps * 1e3], dtype=torch.float64, device=dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = [p.cpu().tolist() for p in parts]
    ms = [r[3] for r in rows]
    return {"devices_distinct": len({tuple(r[:3]) for r in rows}),
            "rank_ms_per_step": {"min": round(min(ms), 3), "max": round(max(ms), 3),
                                 "all": [round(v, 3) for v in ms]}}


def self_launch(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start the ranks ourselves.  The parent has not touched the GPU
    (nothing above calls into HIP), so the ranks are fresh CHILD processes of `python -m torch.distributed.run`
    (subprocess, never exec); their stdout/stderr pass through, the child's exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    launcher = os.environ.get("DEEPJ_BENCH_LAUNCHER")          # tests: a stub in place of torch.distributed.run
    if launcher:
        cmd = [sys.executable, launcher] + cmd[1:]
    print("[bench] starting %d ranks: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--time-steps", type=int, default=128)
    ap.add_argument("--notes", type=int, default=128)
    ap.add_argument("--cpu-sample", type=int, default=8, help="sequences in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (after 1 warm-up)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode timing")
    ap.add_argument("--config", default="baseline", choices=["baseline", "scaled"],
                    help="baseline = BASELINE configs[1] (2x256 + 2x128, B64 T128 N128); scaled = configs[4] "
                         "(3x1024 per axis, B128 T256 N128 as micro-batches)")
    ap.add_argument("--micro", type=int, default=2, help="micro-batches of the scaled config's batch of 128")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dropout", type=float, nargs=2, default=None, metavar=("INPUT", "HIDDEN"),
                    help="override the reference's dropout rates 0.2 0.5 (experiments only; the metric uses the defaults)")
    ap.add_argument("--gen-steps", type=int, default=1024, help="generated time steps for the secondary metric (0 = skip)")
    ap.add_argument("--scaled-steps", type=int, default=2,
                    help="timed steps of the `scaled` sub-record (BASELINE configs[4], after 1 warm-up; 0 = skip)")
    ap.add_argument("--pmc-dir", default=None,
                    help="directory with rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS build "
                         "(counter_collection CSVs): roofline.traffic is computed from them instead of the committed "
                         "profiles/pmc_traffic.json")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        # the driver's plain form `python bench.py --gpus N ...`: no launcher above us -> be the launcher
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the DeepJ engine has no CPU path")
    # rehearsal switches (tests/test_dist_gpu.py runs the N = 2 flow on a one-GPU box): every rank on cuda:0, gloo
    # instead of RCCL (which refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    if os.environ.get("DEEPJ_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("DEEPJ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.config == "scaled":
        return scaled_bench(args, dev, rank, world, dist)

    from music_generator_amd import _lib
    from music_generator_amd.data import synthetic_batch
    from music_generator_amd.engine import DeepJConfig, Engine, Nadam, init_params_numpy

    B, T, N = args.batch, args.time_steps, args.notes
    pin, pdr = args.dropout or (0.2, 0.5)                # model.py:128 defaults
    cfg = DeepJConfig(num_notes=N, time_steps=T, dtype=args.dtype)
    lib = _lib.load()
    notes, chosen, beat, style, target = [torch.from_numpy(a).to(dev)
                                          for a in synthetic_batch(N, T, B, seed=rank)]
    fallback_faults = 0
    for attempt in (0, 1):
        # attempt 1 only after cluster faults in attempt 0 (an exchange wait of the weight-stationary forward sweep
        # expired: the device was not this job's alone): the same run on the per-tile kernels, as Model.fit falls back
        eng = Engine(cfg, B, T, device=dev, input_dropout=pin, dropout=pdr,
                     kernel_flags=_lib.KF_NO_CLUSTER if attempt else 0)
        P = torch.from_numpy(init_params_numpy(cfg, seed=1234)).to(dev)     # replicated weights
        G = torch.zeros_like(P)
        opt = Nadam(P.numel(), dev)
        step = make_step(eng, opt, P, G, (notes, chosen, beat, style, target), world, rank, dist)

        # HIP events around EVERY launch cost ~0.2 ms per step (1.5 %: measured, --no-profile A/B).  So the warm-up
        # steps carry them all (they name the dominant category), the timed region carries the events of the dominant
        # category's launches only -- its live average launch duration is what `roofline.achieved` is computed from --
        # and a few untimed steps behind the timed region fill the per-category table.
        if not args.no_profile:
            lib.dj_profile_enable(1)         # (also clears the categories of a first attempt)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dom_cat = None
        if not args.no_profile:
            warm = read_profile(lib, max(args.warmup, 1))
            fl_keys = list(category_flops(cfg, B, T, N))
            dom_cat = dominant_category(warm, fl_keys) if args.warmup > 0 else "lstm_bwd_time"
            names = [lib.dj_profile_category_name(c).decode() for c in range(lib.dj_profile_category_count())]
            lib.dj_profile_enable(2 + names.index(dom_cat))
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        torch.cuda.synchronize()
        elapsed_local = time.perf_counter() - t0         # this rank's own K steps (its last collective included)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        final_loss = float(loss.cpu()[0])
        dom_ms, post_steps = None, 0
        if not args.no_profile:
            dom_ms = read_profile(lib, args.steps)[dom_cat]
            post_steps = min(args.steps, 5)
            lib.dj_profile_enable(1)
            for i in range(post_steps):      # untimed: the per-category table
                step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            post = read_profile(lib, post_steps)
            lib.dj_profile_enable(0)
        faults = local_faults = eng.cluster_faults()
        if world > 1:
            f = torch.tensor([float(faults)], dtype=torch.float64, device=dev)
            dist.all_reduce(f)
            faults = int(f.cpu()[0])
        # describe a fault only when THIS rank's census of THIS run saw one (an older `last_fault` is not this run's)
        from music_generator_amd.engine import describe_fault_report
        what = describe_fault_report(eng.last_fault) if local_faults and eng.last_fault else "reported by another rank"
        if not faults or attempt == 1:
            break
        fallback_faults = faults
        if rank == 0:
            print(f"[bench] {faults} cluster faults in the timed run ({what}): repeating it on the per-tile kernels "
                  "(DJ_KF_NO_CLUSTER)", file=sys.stderr, flush=True)
        eng.close()
        del eng, step
    if faults or not np.isfinite(final_loss):
        raise SystemExit(f"invalid run: {faults} cluster faults ({what}), final loss {final_loss} -- no number is "
                         "reported (DEEPJ_CLUSTER=0 selects the per-tile kernel)")
    multi = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
        multi = multi_gpu_evidence(dist, dev, world, elapsed_local, args.steps)

    kernels = {}
    roof = roof_all = None
    if not args.no_profile:
        kernels = dict(post)
        kernels[dom_cat] = dom_ms                        # the dominant category: its timed-region value
        fl = category_flops(cfg, B, T, N)
        lps = launches_per_step(cfg)
        dom = dom_cat
        ms = dom_ms
        by = category_bytes(cfg, B, T, N, 2 if args.dtype == "bf16" else 4)
        sec = ms * 1e-3
        tflops = fl[dom] / sec / 1e12 if ms > 0 else 0.0
        gbs = by[dom] / sec / 1e9 if ms > 0 else 0.0
        traffic, tsrc, committed = None, None, None
        try:                                            # HBM bytes per launch of that kernel from the committed PMC passes
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if args.dtype == "bf16" and (B, T, N) == (64, 128, 128) and dom in pm:
                committed = traffic = pm[dom]
                tsrc = "profiles/pmc_traffic.json (committed passes of an earlier run): " + pm["source"]
        except Exception:
            pass
        if args.pmc_dir:                                # passes of THIS build, made in the same gpurun call
            live = pmc_traffic_from_dir(args.pmc_dir, pmc_category)
            if dom in live:
                traffic = live[dom]
                tsrc = ("measured: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes under %s (this build; FETCH_SIZE x 2 "
                        "per MI355X_MICROARCH.md), bytes per launch" % args.pmc_dir)
        # the roof that binds is the one with the larger lower-bound time for this kernel
        t_mfma, t_hbm = fl[dom] / (PEAK_TFLOPS[args.dtype] * 1e12), by[dom] / (PEAK_HBM_GBS * 1e9)
        if t_hbm >= t_mfma:
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4)}
        else:
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(tflops, 2), "peak": PEAK_TFLOPS[args.dtype],
                    "unit": "TFLOP/s", "frac": round(tflops / PEAK_TFLOPS[args.dtype], 4)}
        # the whole step against both roofs, as SURVEY.md 8(d) defines them: train FLOPs per note-step (7,301,376 for the
        # reference model) x note-steps/s per GPU / dense MFMA peak, and minimal-stash bytes per note-step (19,064 B) x
        # note-steps/s per GPU / HBM peak
        nsps = B * T * N * args.steps / elapsed
        cf_ = category_flops(cfg, B, T, N)
        # (+ the octave conv and the heads, which the MFMA categories above do not carry: 2*24*3*64 and 2*Hn*3 per note-step)
        flops_ns = 3 * ((cf_["gemm_xw"] + cf_["lstm_fwd_time"] + cf_["lstm_fwd_note"]) / (B * T * N)
                        + 2 * 2 * cfg.octave * cfg.note_units * cfg.octave_units + 2 * cfg.note_axis_units * 3)
        stash_ns = stash_bytes_per_note_step(cfg, 2 if args.dtype == "bf16" else 4)
        roof.update({"whole_step_mfma_frac": round(flops_ns * nsps / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
                     "whole_step_stash_hbm_frac": round(stash_ns * nsps / (PEAK_HBM_GBS * 1e9), 4),
                     "train_flops_per_note_step": int(flops_ns), "stash_bytes_per_note_step": stash_ns})
        roof.update({"traffic": traffic, "traffic_committed_pmc": committed, "traffic_source": tsrc,
                     "launches_per_step": lps[dom],
                     "avg_launch_ms": round(ms / lps[dom], 4), "bytes_per_launch": by[dom] / lps[dom],
                     "flops_per_launch": fl[dom] / lps[dom],
                     "mfma_tflops": round(tflops, 2), "mfma_frac": round(tflops / PEAK_TFLOPS[args.dtype], 4),
                     "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
        roof_all = {k: {"ms": kernels.get(k, 0.0),
                        "tflops": round(fl[k] / (kernels[k] * 1e-3) / 1e12, 1) if kernels.get(k) else None,
                        "gbs": round(by[k] / (kernels[k] * 1e-3) / 1e9, 1) if kernels.get(k) else None}
                    for k in fl}

    if rank == 0:
        value = world * B * T * N * args.steps / elapsed
        cf = category_flops(cfg, B, T, N)
        flops_step = 3 * (cf["gemm_xw"] + cf["lstm_fwd_time"] + cf["lstm_fwd_note"])
        out = {
            "metric": "note-steps/sec (train)", "value": round(value, 1), "unit": "note-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"biaxial-LSTM train step, batch {B}/GPU x {T} steps x {N} notes "
                                   f"(BASELINE configs[1]), 2x256 time-axis + 2x128 note-axis LSTM, dropout {pin}/{pdr}, "
                                   f"Nadam; random-init weights",
                       "global_batch": B * world, "seq_len": T, "num_notes": N, "parallelism": f"dp{world}",
                       **({"rehearsal": f"backend {backend}, all ranks on one device: not a measurement"}
                          if (backend != "nccl" or os.environ.get("DEEPJ_BENCH_ONE_DEVICE") == "1") else {}),
                       **({"kernels": f"per-tile forward kernels (DJ_KF_NO_CLUSTER) after {fallback_faults} cluster "
                                      "faults in a first timed run"} if fallback_faults else {})},
            "model_tflops_per_s": round(flops_step * world * args.steps / elapsed / 1e12, 2),
            "final_loss": round(final_loss, 5),
            **(multi or {}),
            "roofline": roof, "kernel_ms_per_step": kernels, "kernel_rates": roof_all,
            "kernel_ms_source": (None if args.no_profile else
                                 f"HIP events (dj_profile_*): '{dom_cat}' over the {args.steps} timed steps (the only "
                                 f"launches that carry events there: events around every launch cost ~0.2 ms per "
                                 f"step); the other categories over {post_steps} untimed steps behind the timed region"),
        }
        eng.close()
        del eng
        torch.cuda.empty_cache()
        def leg(fn):
            """A secondary record must not cost the headline its line: its failure is reported in its place."""
            try:
                return fn()
            except (Exception, SystemExit) as e:                 # noqa: BLE001
                print(f"[bench] secondary record failed: {e!r}", file=sys.stderr, flush=True)
                return {"error": repr(e)[:300]}

        if world == 1 and args.dtype != "f32" and not args.no_fp32:
            out["fp32_parity_mode"] = leg(lambda: fp32_parity_mode(dict(num_notes=N, time_steps=T), B, T, N, pin, pdr,
                                                                   dev, rank))
        if world == 1 and args.gen_steps > 0:
            out["generation"] = leg(lambda: generation_bench(args.dtype, args.gen_steps))
            if args.dtype != "f32" and not args.no_fp32 and "error" not in out["generation"]:
                # the mode in which the sampled notes are certified against the fp32 oracle (DESIGN.md "Sampling parity")
                g32 = leg(lambda: generation_bench("f32", min(args.gen_steps, 256)))
                out["generation"]["fp32_parity_mode"] = g32 if "error" in g32 else {
                    k: g32[k] for k in ("value", "unit", "ms_per_time_step", "steps", "near_tie_draws", "draws")}
        if world == 1 and args.scaled_steps > 0 and args.dtype == "bf16" and (B, T, N) == (64, 128, 128):
            # BASELINE configs[4] next to the headline, so that its number is driver-visible: 1 warm-up + 2 timed steps
            free, _ = torch.cuda.mem_get_info(dev)
            if free > 230 * 2 ** 30:
                def scaled():
                    rec = scaled_record(args.dtype, args.micro, args.scaled_steps, 1, args.dropout, dev, rank, 1, None)
                    return {k: rec[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                                "model_tflops_per_s", "final_loss", "roofline", "kernel_ms_per_step",
                                                "kernel_tflops")}
                out["scaled"] = leg(scaled)
            else:
                out["scaled"] = {"skipped": "needs a 218 GiB workspace; %.0f GiB free" % (free / 2 ** 30)}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, T, N, min(args.cpu_sample, B), pin, pdr, B, steps=args.cpu_steps)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
