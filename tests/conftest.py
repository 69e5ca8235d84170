import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fault_injection: the test injects cluster faults on purpose (census not asserted)")
    # the CPU oracle is torch-CPU: give it the cores this process may really use (affinity mask capped by the cgroup
    # quota).  os.cpu_count() on a GPU box is the whole host: 8x oversubscription made oracle steps 10x slower.
    try:
        import torch
        n = len(os.sched_getaffinity(0))
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
        except Exception:
            pass
        torch.set_num_threads(max(1, min(n, 32)))
    except Exception:
        pass


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class _DjEnv:
    """Library-level DEEPJ_* switches are read once per process (include/deepj_hip.h DJ_KF_*): a test that flips one
    mid-process goes through here, which re-reads them after every change (dj_env_reload) and once more at teardown."""

    def __init__(self, monkeypatch):
        self.mp = monkeypatch

    def _reload(self):
        from music_generator_amd import _lib
        _lib.load().dj_env_reload()

    def set(self, name, value):
        self.mp.setenv(name, str(value))
        self._reload()

    def unset(self, name):
        self.mp.delenv(name, raising=False)
        self._reload()


@pytest.fixture
def djenv(monkeypatch):
    env = _DjEnv(monkeypatch)
    yield env
    monkeypatch.undo()
    env._reload()


# ---- cluster-exchange census.  The weight-stationary cluster kernels count expired waits
# and misplaced clusters and every host path that sees a count records it in engine.FAULT_LOG; training falls back to
# the per-tile kernels and goes on, so a green test would not by itself say the waits held.  Every GPU test that does
# not inject faults on purpose (marker `fault_injection`) therefore FAILS if, while it ran, the host observed a fault,
# step-wise generation repeated a step, or a live engine holds fault counts nobody read.  The stall census (the longest
# gap between two polls of any exchange wait: a wave that was off the device) is collected from every engine and
# printed at the end of the session -- evidence about the box even when nothing expired.
_CENSUS = {"live_readings": 0, "stalled_waits": 0, "max_poll_gap_cycles": 0, "backward_clock_polls": 0,
           "faults_in_injection_tests": 0, "worst": None}


def _census_live_engines():
    """Engines still alive after a test.  Most die with the test's locals: Engine.__del__ does NO HIP work (round 5) and
    hands their workspaces to engine.drain_pending(), which files their census under E.CENSUS -- and any count nobody
    read under E.FAULT_LOG -- here, at a safe point."""
    import gc
    from music_generator_amd import engine as E
    gc.collect()
    E.drain_pending()
    unread = []
    for eng in list(E._LIVE):
        try:
            rep = eng.cluster_fault_report()
        except Exception:
            continue
        _CENSUS["live_readings"] += 1
        _CENSUS["stalled_waits"] = max(_CENSUS["stalled_waits"], rep["stalled_waits"])
        _CENSUS["backward_clock_polls"] = max(_CENSUS["backward_clock_polls"], rep["backward_clock_polls"])
        if rep["max_poll_gap_cycles"] > _CENSUS["max_poll_gap_cycles"]:
            _CENSUS["max_poll_gap_cycles"] = rep["max_poll_gap_cycles"]
        if rep["expired"] or rep["misplaced"]:
            unread.append(E.describe_fault_report(rep))
    return unread


@pytest.fixture(autouse=True)
def cluster_fault_census(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from music_generator_amd import engine as E
    from music_generator_amd import generate as Gn
    n0, r0 = len(E.FAULT_LOG), Gn.repeated_steps
    yield
    unread = _census_live_engines()
    new = E.FAULT_LOG[n0:]
    if request.node.get_closest_marker("fault_injection") is not None:
        _CENSUS["faults_in_injection_tests"] += len(new)
        for eng in list(E._LIVE):                      # leave no injected count behind for the next test
            try:
                eng.cluster_faults("after an injection test")
            except Exception:
                pass
        del E.FAULT_LOG[n0:]
        return
    if new and _CENSUS["worst"] is None:
        _CENSUS["worst"] = {k: v for k, v in new[0].items()}
    assert not new, "cluster faults observed during the test: " + "; ".join(
        "%s: %s" % (e["what"], E.describe_fault_report(e)) for e in new)
    assert Gn.repeated_steps == r0, "step-wise generation repeated %d steps" % (Gn.repeated_steps - r0)
    assert not unread, "fault counts left unread in a live engine: " + "; ".join(unread)


def pytest_terminal_summary(terminalreporter):
    try:
        from music_generator_amd import engine as E
    except Exception:
        return
    engines = E.CENSUS["engines"] + _CENSUS["live_readings"]
    if not engines:
        return
    E.drain_pending()
    stalled = max(E.CENSUS["stalled_waits"], _CENSUS["stalled_waits"])
    back = max(E.CENSUS["backward_clock_polls"], _CENSUS["backward_clock_polls"])
    gap = max(E.CENSUS["max_poll_gap_cycles"], _CENSUS["max_poll_gap_cycles"])
    rec = {"engines": engines, "stalled_waits": stalled, "max_poll_gap_cycles": gap, "backward_clock_polls": back,
           "faults_in_injection_tests": _CENSUS["faults_in_injection_tests"], "faults_elsewhere": _CENSUS["worst"]}
    terminalreporter.write_line(
        "cluster exchange census: %d engines, waits with polls > 2^20 cycles apart: %d, longest gap between two polls of a "
        "wait: %d shader cycles (~%.0f us at 2.1 GHz; 0 = below the 2^17-cycle recording threshold), polls behind a shader "
        "clock that had gone backwards (wave restored on another XCC; most in one workspace): %d, faults in injection "
        "tests: %d, faults elsewhere: %s" % (engines, stalled, gap, gap / 2100.0, back, rec["faults_in_injection_tests"],
                                            "NONE" if rec["faults_elsewhere"] is None else repr(rec["faults_elsewhere"])))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "cluster_census.json"), "w") as f:
            json.dump(rec, f, default=str)
