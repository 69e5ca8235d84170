"""bench.py's OWN timed step function (make_step: forward + BPTT, one all-reduce(sum) of the flat gradient,
Nadam with the 1/world scale) run by two gloo ranks on the CPU with the oracle behind the Engine / Nadam
interfaces: both ranks must end with bit-identical parameters, equal to a single-process run that averages the
two ranks' gradients.  This is the N > 1 path the driver launches on RCCL (no 8-GPU node is available to the
builder); plus the byte/flop accounting helpers of the roofline block."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench
from oracle_backend import OracleBackend
from music_generator_amd.engine import DeepJConfig
from music_generator_amd.data import synthetic_batch
torch.set_num_threads(2)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = DeepJConfig(num_notes=12, time_steps=4, time_axis_units=128, note_axis_units=128)
be = OracleBackend()
eng = be.engine(cfg, 2, 4, 0.2, 0.5)
P = be.tensor(be.init_params(cfg, 1234))
G = torch.zeros_like(P)
opt = be.optimizer(P.numel())
batch = [be.tensor(a) for a in synthetic_batch(12, 4, 2, seed=rank)]
step = bench.make_step(eng, opt, P, G, batch, world, rank, dist)
losses = [float(step(i)[0]) for i in range(3)]
np.save(os.path.join({out!r}, "p%d.npy" % rank), P.numpy())
np.save(os.path.join({out!r}, "l%d.npy" % rank), np.array(losses))
dist.destroy_process_group()
'''


def test_bench_step_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    np.testing.assert_array_equal(p0, p1)                           # replicas stay bit-identical
    # single-process reference: per-rank batches (seed = rank) and dropout seeds (i * world + rank), mean gradient
    from music_generator_amd.data import synthetic_batch
    from oracle import deepj_oracle as O
    ocfg = O.OracleConfig(num_notes=12, time_steps=4, time_axis_units=128, note_axis_units=128)
    flat = O.flatten_params(ocfg, O.init_params(ocfg, 1234))
    st = O.NadamState()
    batches = [synthetic_batch(12, 4, 2, seed=r_) for r_ in range(2)]
    ref_losses = [[], []]
    for i in range(3):
        gs = []
        for r_ in range(2):
            masks = O.make_masks(ocfg, 2, i * 2 + r_, 0.2, 0.5, T=4)
            l, _, g = O.loss_and_grads(ocfg, O.unflatten_params(ocfg, flat), batches[r_], masks)
            gs.append(O.flatten_params(ocfg, g))
            ref_losses[r_].append(l)
        flat = O.nadam_step(flat, ((gs[0] + gs[1]) * np.float32(0.5)).astype(np.float32), st)
    np.testing.assert_allclose(p0, flat, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(np.load(tmp_path / "l0.npy"), ref_losses[0], rtol=1e-5)
    np.testing.assert_allclose(np.load(tmp_path / "l1.npy"), ref_losses[1], rtol=1e-5)


def test_roofline_accounting_matches_survey():
    """2,433,792 forward FLOPs per note-step (train = 3x); the byte model counts the bf16 gate
    stash at one byte per element."""
    sys.path.insert(0, ROOT)
    import bench
    from music_generator_amd.engine import DeepJConfig
    cfg = DeepJConfig(num_notes=128, time_steps=128, dtype="bf16")
    fl = bench.category_flops(cfg, 1, 1, 1)
    fwd = fl["gemm_xw"] + fl["lstm_fwd_time"] + fl["lstm_fwd_note"]
    assert fwd == 2433792 - 9216 - 768                      # conv and heads are elementwise categories
    by = bench.category_bytes(cfg, 1, 1, 1, 2)
    assert by["lstm_bwd_time"] == 2 * (6 * 256 * 2 + 4 * 256)           # stash 4H x 1 B + (c, dh, dz) 6H x 2 B
    assert by["lstm_fwd_time"] == (94 + 2 * 256) * 2 + 1024 + (256 + 2 * 256) * 2 + 1024
    by32 = bench.category_bytes(DeepJConfig(num_notes=128, dtype="f32"), 1, 1, 1, 4)
    assert by32["lstm_bwd_time"] == 2 * 10 * 256 * 4


def test_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher above it (the driver's plain command form) must start the N ranks
    itself -- as a CHILD `python -m torch.distributed.run ... bench.py <same args>` (subprocess, not exec, before any GPU
    call) -- pass their output through and exit with the child's code.  Checked with a stub in place of the launcher."""
    stub = tmp_path / "stub_launcher.py"
    stub.write_text("import json, sys\n"
                    "print(json.dumps({'stub_argv': sys.argv[1:]}), flush=True)\n"
                    "sys.exit(7)\n")
    env = dict(os.environ, DEEPJ_BENCH_LAUNCHER=str(stub))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, r.stderr[-2000:]
    import json
    argv = json.loads(r.stdout.strip().splitlines()[-1])["stub_argv"]
    assert argv[:2] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_bench_self_launch_through_the_real_launcher_fails_loudly_without_gpus():
    """The same path through the real torch.distributed.run on this GPU-less machine: two ranks start, each refuses to
    run without a HIP device, and the parent reports the failure (non-zero exit, no JSON line) instead of hanging or
    printing a number."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DEEPJ_BENCH_LAUNCHER"):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a HIP device" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout


def test_pmc_traffic_from_rocprof_passes(tmp_path):
    """bench.py --pmc-dir: HBM bytes per launch from the two rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    counter_collection CSVs anywhere under the directory): read = 2 x FETCH_SIZE KB (gfx950 tallies 128-byte requests at
    64 bytes, MI355X_MICROARCH.md), write = WRITE_SIZE KB, averaged per launch and summed per bench category."""
    import bench
    bwd = "_ZN12_GLOBAL__N_115lstm_bwd_kernelIDF16bLi256ELb0ELi0EEEvPKNS_6StashTIT_E4typeE"
    note = "_ZN12_GLOBAL__N_115lstm_bwd_kernelIDF16bLi128ELb0ELi1EEEvPKNS_6StashTIT_E4typeE"
    other = "void at::native::vectorized_elementwise_kernel"
    hdr = "Correlation_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    f = tmp_path / "fetch" / "runc"
    w = tmp_path / "write" / "runc"
    f.mkdir(parents=True); w.mkdir(parents=True)
    (f / "1_counter_collection.csv").write_text(hdr + "".join(
        '%d,"%s",FETCH_SIZE,%f\n' % (i, k, v) for i, (k, v) in enumerate([(bwd, 1000.0), (bwd, 1200.0), (note, 500.0), (other, 9.0)])))
    (w / "2_counter_collection.csv").write_text(hdr + "".join(
        '%d,"%s",WRITE_SIZE,%f\n' % (i, k, v) for i, (k, v) in enumerate([(bwd, 2000.0), (bwd, 2000.0), (note, 100.0)])))
    got = bench.pmc_traffic_from_dir(str(tmp_path), bench.pmc_category)
    assert set(got) == {"lstm_bwd_time", "lstm_bwd_note"}
    assert got["lstm_bwd_time"] == 2 * 1024 * 1100.0 + 1024 * 2000.0          # per launch: 2 x mean fetch + mean write
    assert got["lstm_bwd_note"] == 2 * 1024 * 500.0 + 1024 * 100.0
    assert bench.pmc_category("lstm_fwd_cluster_kernelILb0ELi16EE") == "lstm_fwd_time"
    assert bench.pmc_category("lstm_wgrad_bf16_kernel") == "gemm_dw" and bench.pmc_category(other) is None


def test_dominant_category_prefers_a_sweep_within_five_percent():
    """bench.dominant_category: the roofline names the largest category, but a recurrent sweep (one kernel, one shape)
    within 5 % of it is preferred over the multi-shape GEMM categories, so that near-ties do not flip the line."""
    import bench
    keys = ["gemm_dw", "gemm_dx", "lstm_bwd_time", "lstm_fwd_time"]
    assert bench.dominant_category({"gemm_dw": 2.85, "gemm_dx": 1.7, "lstm_bwd_time": 2.82, "lstm_fwd_time": 2.5}, keys) == "lstm_bwd_time"
    assert bench.dominant_category({"gemm_dw": 3.5, "gemm_dx": 1.7, "lstm_bwd_time": 2.82, "lstm_fwd_time": 2.5}, keys) == "gemm_dw"
    assert bench.dominant_category({"gemm_dw": 2.0, "gemm_dx": 1.7, "lstm_bwd_time": 2.82, "lstm_fwd_time": 2.9}, keys) == "lstm_fwd_time"
