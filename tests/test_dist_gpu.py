"""RCCL meets the training step on the device: torch.distributed backend "nccl" (= RCCL on ROCm) initialised at
world size 1 on the one GPU of the test box, in a FRESH child process, driving (a) bench.py's own timed step
(make_step with its all-reduce forced) and (b) the collective branch of Model._train_step (DEEPJ_DIST_WORLD1=1) for 3
steps each on the production kernel selection (bf16, B16 x T16 x N128: 64 time-axis tiles -> the weight-stationary
cluster kernel): parameters and losses equal the non-distributed path, exactly one collective per step, no cluster
fault with RCCL's kernels on the same device; (c) the same with DEEPJ_DDP_EXACT=1 (dist.all_gather of the pitch_bins
parts + dj_train_fwd_bwd_mb); (d) bench.py's `scaled` record (BASELINE configs[4]) with its 187.6 MB all-reduce forced.  This is the N = 1 end of the path the driver launches at N = 2, 4, 8
(no multi-GPU node is available to the builder: the multi-rank arithmetic is covered by the 2-rank gloo tests)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench
from music_generator_amd.data import synthetic_batch
from music_generator_amd.engine import DeepJConfig, Engine, Nadam, init_params_numpy
from music_generator_amd.model import build_models

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
calls = {{"n": 0}}
_orig = dist.all_reduce
def counted(*a, **k):
    calls["n"] += 1
    return _orig(*a, **k)
dist.all_reduce = counted
gathers = {{"n": 0}}
_orig_gather = dist.all_gather
def counted_gather(*a, **k):
    gathers["n"] += 1
    return _orig_gather(*a, **k)
dist.all_gather = counted_gather

B, T, N, STEPS = 16, 16, 128, 3
cfg = DeepJConfig(num_notes=N, time_steps=T, dtype="bf16")
batch = [torch.from_numpy(a).to(dev) for a in synthetic_batch(N, T, B, seed=0)]
out = {{}}

def run_bench_step(force):
    eng = Engine(cfg, B, T, device=dev, input_dropout=0.2, dropout=0.5, fuse_xw_min_tiles=1)
    P = torch.from_numpy(init_params_numpy(cfg, seed=1234)).to(dev)
    G = torch.zeros_like(P)
    step = bench.make_step(eng, Nadam(P.numel(), dev), P, G, batch, 1, 0, dist, force_collective=force)
    losses = [float(step(i).cpu()[0]) for i in range(STEPS)]
    torch.cuda.synchronize()
    return P.cpu().numpy(), losses, eng.cluster_faults()

calls["n"] = 0
p1, l1, f1 = run_bench_step(True)
out["bench_collectives"] = calls["n"]
p0, l0, f0 = run_bench_step(False)
out["bench_faults"] = [f1, f0]
out["bench_loss"] = [l1, l0]
pinit = init_params_numpy(cfg, seed=1234)
out["bench_param_maxdiff"] = float(np.abs(p1 - p0).max())
out["bench_param_moved"] = float(np.abs(p1 - pinit).max())
out["bench_update_rel_l2"] = float(np.linalg.norm(p1 - p0) / np.linalg.norm(p1 - pinit))

def run_model(dist_branch, exact=False):
    if dist_branch:
        os.environ["DEEPJ_DIST_WORLD1"] = "1"
    else:
        os.environ.pop("DEEPJ_DIST_WORLD1", None)
    if exact:
        os.environ["DEEPJ_DDP_EXACT"] = "1"
    else:
        os.environ.pop("DEEPJ_DDP_EXACT", None)
    m = build_models(time_steps=T, config=cfg, seed=5)[0]
    x = [b.cpu().numpy() for b in batch]
    losses = [m.train_on_batch([x[0], x[1], x[2], x[3]], [x[4]]) for _ in range(STEPS)]
    eng = m._s.engine(B, T, train=True)
    return np.concatenate([w.ravel() for w in m.get_weights()]), losses, eng.cluster_faults(), m._s.kernel_flags

winit = np.concatenate([w.ravel() for w in build_models(time_steps=T, config=cfg, seed=5)[0].get_weights()])

calls["n"] = 0
w1, ml1, mf1, k1 = run_model(True)
out["model_collectives"] = calls["n"]
w0, ml0, mf0, k0 = run_model(False)
out["model_faults"] = [mf1, mf0, k1, k0]
out["model_loss"] = [ml1, ml0]
out["model_param_maxdiff"] = float(np.abs(w1 - w0).max())
out["model_update_rel_l2"] = float(np.linalg.norm(w1 - w0) / np.linalg.norm(w1 - winit))

# exact mode (DEEPJ_DDP_EXACT=1) over RCCL: dist.all_gather of the pitch_bins parts + dj_train_fwd_bwd_mb, then the one
# all-reduce -- at world size 1 the global batch IS the shard, so the result must be the non-distributed step's
calls["n"] = 0; gathers["n"] = 0
we, mle, mfe, ke = run_model(True, exact=True)
out["exact_collectives"] = [calls["n"], gathers["n"]]
out["exact_faults"] = [mfe, ke]
out["exact_loss"] = [mle, ml0]
out["exact_param_maxdiff"] = float(np.abs(we - w0).max())
out["exact_update_rel_l2"] = float(np.linalg.norm(we - w0) / np.linalg.norm(we - winit))
os.environ.pop("DEEPJ_DDP_EXACT", None); os.environ.pop("DEEPJ_DIST_WORLD1", None)

# BASELINE configs[4] (bench.py --config scaled: 3 x 1024 per axis, B128 x T256 x N128 as two exact micro-batches through
# the 218 GiB workspace) with its 187.6 MB gradient all-reduce forced through RCCL: 2 timed steps + the profiled one
free, _ = torch.cuda.mem_get_info(dev)
if free > 235 * 2 ** 30:
    calls["n"] = 0
    r1 = bench.scaled_record("bf16", 2, 2, 0, None, dev, 0, 1, dist, force_collective=True, return_params=True)
    out["scaled_collectives"] = calls["n"]
    r0 = bench.scaled_record("bf16", 2, 2, 0, None, dev, 0, 1, dist, force_collective=False, return_params=True)
    ps1, ps0 = r1.pop("params"), r0.pop("params")
    from music_generator_amd.engine import DeepJConfig as _DC
    pinit_s = init_params_numpy(_DC(num_notes=128, time_steps=256, dtype="bf16", time_axis_units=1024, note_axis_units=1024,
                                    time_axis_layers=3, note_axis_layers=3), seed=1234)
    out["scaled_faults"] = [r1["cluster_faults"], r0["cluster_faults"]]
    out["scaled_loss"] = [r1["final_loss"], r0["final_loss"]]
    out["scaled_ms"] = [r1["ms_per_step"], r0["ms_per_step"]]
    out["scaled_param_maxdiff"] = float(np.abs(ps1 - ps0).max())
    out["scaled_update_rel_l2"] = float(np.linalg.norm(ps1 - ps0) / np.linalg.norm(ps1 - pinit_s))
else:
    out["scaled_skipped"] = "needs a 218 GiB workspace; %.0f GiB free" % (free / 2 ** 30)
from music_generator_amd import engine as _E
out["fault_log"] = [[e["what"], e["expired"], e["misplaced"]] for e in _E.FAULT_LOG]
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
'''


def test_rccl_world1_drives_bench_step_and_model_step(gpu_device, tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_DIST_WORLD1"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    o = json.loads(line[len("RESULT "):])
    print("rccl world-1:", {k: v for k, v in o.items() if "loss" not in k})
    steps = 3
    # bench.py's step: one all-reduce per step, same parameters as without it (an all-reduce over one rank is the
    # identity; fp32 atomics reorder the gradient sums between runs), the cluster kernel never faulted beside RCCL
    assert o["bench_collectives"] == steps
    assert o["bench_faults"] == [0, 0]
    # (Nadam divides by sqrt(v): where a gradient is rounding noise the two runs may step in different directions, so
    # single parameters differ by up to ~lr; the UPDATE as a whole agrees)
    assert o["bench_param_moved"] > 1e-3 and o["bench_param_maxdiff"] < 3 * 2e-3 and o["bench_update_rel_l2"] < 5e-3, o
    assert all(abs(a - b) < 1e-4 * abs(b) for a, b in zip(*o["bench_loss"])), o["bench_loss"]
    # Model._train_step: [gradient | weight, weight * loss, faults] in ONE collective per step
    assert o["model_collectives"] == steps
    assert o["model_faults"] == [0, 0, 0, 0]
    assert o["model_param_maxdiff"] < 3 * 2e-3 and o["model_update_rel_l2"] < 5e-3, o
    assert all(abs(a - b) < 1e-4 * abs(b) for a, b in zip(*o["model_loss"])), o["model_loss"]
    # exact mode over RCCL: one all_gather + one all_reduce per step, the non-distributed step's numbers, no fault
    assert o["exact_collectives"] == [steps, steps] and o["exact_faults"] == [0, 0]
    assert o["exact_param_maxdiff"] < 3 * 2e-3 and o["exact_update_rel_l2"] < 5e-3, o
    assert all(abs(a - b) < 1e-4 * abs(b) for a, b in zip(*o["exact_loss"])), o["exact_loss"]
    # the scaled config's step with its 187.6 MB all-reduce through RCCL (2 timed + 1 profiled step)
    if "scaled_skipped" not in o:
        assert o["scaled_collectives"] == 3 and o["scaled_faults"] == [0, 0]
        assert abs(o["scaled_loss"][0] - o["scaled_loss"][1]) < 1e-3 * abs(o["scaled_loss"][1]), o["scaled_loss"]
        assert o["scaled_param_maxdiff"] < 3 * 2e-3 and o["scaled_update_rel_l2"] < 2e-2, o
    assert o["fault_log"] == []


WORKER2 = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from music_generator_amd.data import synthetic_batch
from music_generator_amd.engine import DeepJConfig
from music_generator_amd.model import build_models
from music_generator_amd._lib import KF_NO_CLUSTER
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", world_size=2, rank=int(sys.argv[1]))
rank = dist.get_rank()
try:
    t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); assert float(t[0]) == 2.0
except Exception as e:
    print("RESULT " + json.dumps({{"skip": "gloo cannot all-reduce device tensors here: %s" % str(e)[:120]}}), flush=True)
    sys.exit(0)
B, T, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = DeepJConfig(num_notes=N, time_steps=T, dtype="bf16")
a = synthetic_batch(N, T, B, seed=0)
x, y = [a[0], a[1], a[2], a[3]], [a[4]]
m = build_models(time_steps=T, config=cfg, seed=5)[0]
np.random.seed(0)
hist = m.fit(x, y, epochs=3, batch_size=B, verbose=0, shuffle=False)
w = np.concatenate([v.ravel() for v in m.get_weights()])
from music_generator_amd import engine as E
print("RESULT " + json.dumps({{"rank": rank, "loss": hist.history["loss"], "digest": float(np.dot(w, np.cos(np.arange(w.size) * 0.37))),
                              "wnorm": float(np.linalg.norm(w)), "fell_back": bool(m._s.kernel_flags & KF_NO_CLUSTER),
                              "fault_log": [[e["what"], e["expired"], e["misplaced"]] for e in E.FAULT_LOG]}}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _two_ranks(tmp_path, shape, env_common=None, env_rank=None):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_common or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_DIST_WORLD1"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r)] + [str(v) for v in shape],
                              env=dict(env, **((env_rank or {}).get(r, {}))), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1000:], se[-3000:])
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    if "skip" in res[0]:
        pytest.skip(res[0]["skip"])
    return sorted(res, key=lambda r: r["rank"])


def test_two_ranks_share_one_gpu(gpu_device, tmp_path):
    """Data parallel with TWO ranks on the ONE GPU of the test box (gloo carries the collective; RCCL refuses two ranks on
    one device): both processes run the real bf16 training step at the BASELINE per-rank shape (64 sequences each) at the
    same time on the same device.  Two full-chip grids of spin-waiting cluster workgroups from two processes cannot both
    be co-resident -- that is what the bounded waits exist for, and whether one expires would depend on the dispatch
    order of the day -- so this rehearsal runs the per-tile kernels (DEEPJ_CLUSTER=0) and is deterministic: replicas
    bit-identical, loss falling, nothing fell back, nothing in the fault log.  The joint fallback itself is the next test."""
    a, b = _two_ranks(tmp_path, (128, 16, 128), env_common={"DEEPJ_CLUSTER": "0"})
    print("two ranks on one GPU:", a, b)
    assert a["digest"] == b["digest"] and a["wnorm"] == b["wnorm"]          # replicas stay bit-identical
    assert a["loss"] == b["loss"] and all(np.isfinite(a["loss"])) and a["loss"][-1] < a["loss"][0]
    assert not a["fell_back"] and not b["fell_back"] and a["fault_log"] == [] and b["fault_log"] == []


@pytest.mark.fault_injection
def test_two_ranks_fall_back_together_on_an_injected_fault(gpu_device, tmp_path):
    """The fault census travels in the step's ONE all-reduce: with a placement fault injected into rank 1 ONLY
    (DEEPJ_DEBUG_CLUSTER_FAULT=1 in that process's environment; both ranks' cluster grids are small enough to be
    co-resident: 6 tiles -> 64 workgroups each), BOTH ranks see the count, both switch to the per-tile kernels and repeat
    the step, and the replicas stay bit-identical.  Rank 0 recorded no fault of its own."""
    a, b = _two_ranks(tmp_path, (8, 8, 48), env_rank={1: {"DEEPJ_DEBUG_CLUSTER_FAULT": "1"}})
    print("injected fault in rank 1:", a, b)
    assert a["fell_back"] and b["fell_back"]
    assert a["digest"] == b["digest"] and a["wnorm"] == b["wnorm"]
    assert a["loss"] == b["loss"] and all(np.isfinite(a["loss"])) and a["loss"][-1] < a["loss"][0]
    assert a["fault_log"] == [] and len(b["fault_log"]) == 1 and b["fault_log"][0][2] > 0      # misplaced, in rank 1 only


def test_bench_two_rank_flow_on_one_gpu(gpu_device):
    """`python bench.py --gpus 2` exactly as the driver types it, on the one-GPU test box: bench.py starts its own two ranks
    (torch.distributed.run as a child), both rehearse on cuda:0 with gloo as the collective backend
    (DEEPJ_BENCH_ONE_DEVICE / DEEPJ_BENCH_BACKEND: RCCL refuses two ranks on one device), warm up, barrier, time K steps of
    make_step with its all-reduce, take the MAX over ranks, and rank 0 prints ONE JSON line for the whole job.  The flow of
    the N > 1 bench on real kernels; the number itself is labelled a rehearsal."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DEEPJ_BENCH_ONE_DEVICE="1", DEEPJ_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_BENCH_LAUNCHER"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and "rehearsal" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 64 * 128 * 128 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]
    assert np.isfinite(d["final_loss"]) and d["cpu_baseline"] is None and "scaled" not in d
    # multi-GPU evidence of the line: the ranks' device identities were all-gathered -- both ranks of this
    # REHEARSAL sit on the one test GPU, so exactly 1 distinct device (a real N-GPU run prints N) -- and every rank's own
    # time per step is there, none of them above the job's (the timed region ends at the slowest rank's barrier)
    assert d["devices_distinct"] == 1, d.get("devices_distinct")
    rk = d["rank_ms_per_step"]
    assert len(rk["all"]) == 2 and 0 < rk["min"] <= rk["max"] <= d["ms_per_step"] * 1.001, (rk, d["ms_per_step"])


WORKER3 = r'''
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from music_generator_amd.data import synthetic_batch
from music_generator_amd.engine import DeepJConfig
from music_generator_amd.model import build_models
torch.cuda.set_device(0)
world = int(sys.argv[2])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", world_size=world, rank=int(sys.argv[1]))
    try:
        t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); assert float(t[0]) == 2.0
    except Exception as e:
        print("RESULT " + json.dumps({{"skip": "gloo cannot all-reduce device tensors here: %s" % str(e)[:120]}}), flush=True)
        sys.exit(0)
B, T, N = 6, 5, 20                                       # N % 12 != 0 too; batches of 6 and (7th sample) 1
cfg = DeepJConfig(num_notes=N, time_steps=T)             # fp32
a = synthetic_batch(N, T, 7, seed=2)
m = build_models(time_steps=T, config=cfg, seed=5)[0]    # dropout 0.2 / 0.5 on
np.random.seed(0)
hist = m.fit([a[0], a[1], a[2], a[3]], [a[4]], epochs=2, batch_size=B, verbose=0, shuffle=False)
w = np.concatenate([v.ravel() for v in m.get_weights()])
np.save(sys.argv[3], w)
print("RESULT " + json.dumps({{"loss": hist.history["loss"]}}), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def test_exact_data_parallel_equals_single_process_on_the_gpu(gpu_device, tmp_path):
    """DEEPJ_DDP_EXACT=1 on the HIP kernels: two ranks (both on the one GPU, gloo as the collective) fit 7 samples at
    global batch 6 -- shards of 3 + 3, then 1 + 0 -- and end at the weights ONE process reaches on the whole batches
    with the same model seed: dj_pitch_bins with the rank's batch offset, the all-gathered global table,
    dj_train_fwd_bwd_mb with the global batch's dropout masks (fp32; N = 20 exercises the N % 12 != 0 reshape)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker3.py"
    script.write_text(WORKER3.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DEEPJ_DDP_EXACT="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_DIST_WORLD1"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, str(script), "0", "1", str(tmp_path / "w_one.npy")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, (one.stdout[-1000:], one.stderr[-3000:])
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path / ("w_r%d.npy" % r))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1000:], se[-3000:])
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    if "skip" in res[0]:
        pytest.skip(res[0]["skip"])
    l_one = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])["loss"]
    w1, wa, wb = (np.load(tmp_path / n) for n in ("w_one.npy", "w_r0.npy", "w_r1.npy"))
    np.testing.assert_array_equal(wa, wb)
    assert res[0]["loss"] == res[1]["loss"]
    np.testing.assert_allclose(res[0]["loss"], l_one, rtol=1e-5)
    np.testing.assert_allclose(wa, w1, rtol=1e-4, atol=2e-5)
