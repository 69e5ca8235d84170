import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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
