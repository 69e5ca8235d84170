"""RCCL meets the training step on the device: torch.distributed backend "nccl" (= RCCL on ROCm) initialised at
world size 1 on the one GPU of the test box, in a FRESH child process, driving (a) bench.py's own timed step
(make_step with its all-reduce forced) and (b) the collective branch of Model._train_step (DEEPJ_DIST_WORLD1=1) for 3
steps each on the production kernel selection (bf16, B16 x T16 x N128: 64 time-axis tiles -> the weight-stationary
cluster kernel): parameters and losses equal the non-distributed path, exactly one collective per step, no cluster
fault with RCCL's kernels on the same device.  This is the N = 1 end of the path the driver launches at N = 2, 4, 8
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

def run_model(dist_branch):
    if dist_branch:
        os.environ["DEEPJ_DIST_WORLD1"] = "1"
    else:
        os.environ.pop("DEEPJ_DIST_WORLD1", None)
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
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
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
B, T, N = 128, 16, 128                                  # global batch 128 -> 64 per rank: 256 time-axis tiles per rank, a full-chip cluster grid each
cfg = DeepJConfig(num_notes=N, time_steps=T, dtype="bf16")
a = synthetic_batch(N, T, B, seed=0)
x, y = [a[0], a[1], a[2], a[3]], [a[4]]
m = build_models(time_steps=T, config=cfg, seed=5)[0]
np.random.seed(0)
hist = m.fit(x, y, epochs=3, batch_size=B, verbose=0, shuffle=False)
w = np.concatenate([v.ravel() for v in m.get_weights()])
print("RESULT " + json.dumps({{"rank": rank, "loss": hist.history["loss"], "digest": float(np.dot(w, np.cos(np.arange(w.size) * 0.37))),
                              "wnorm": float(np.linalg.norm(w)), "fell_back": bool(m._s.kernel_flags & KF_NO_CLUSTER)}}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_share_one_gpu(gpu_device, tmp_path):
    """Data parallel with TWO ranks on the ONE GPU of the test box (gloo carries the collective; RCCL refuses two ranks on
    one device): both processes run the real bf16 training step -- cluster kernels included -- at the same time on the
    same device, which is exactly the situation the cluster kernels' run-time checks exist for (another process's
    kernels keep members from being co-resident).  Whatever happens -- clean steps, or expired waits counted, summed over
    the ranks in the step's one all-reduce and answered by both ranks falling back to the per-tile kernel together --
    the run must end with a finite, falling loss and IDENTICAL weights on both ranks."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_DIST_WORLD1"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1000:], se[-3000:])
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    if "skip" in res[0]:
        pytest.skip(res[0]["skip"])
    print("two ranks on one GPU:", res)
    a, b = sorted(res, key=lambda r: r["rank"])
    assert a["digest"] == b["digest"] and a["wnorm"] == b["wnorm"]          # replicas stay bit-identical
    assert a["loss"] == b["loss"] and all(np.isfinite(a["loss"])) and a["loss"][-1] < a["loss"][0]
    assert a["fell_back"] == b["fell_back"]                                 # the census travels in the all-reduce


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
