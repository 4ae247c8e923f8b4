import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_cores() -> int:
    """Cores this process may use: affinity capped by the cgroup CPU quota.  os.cpu_count() reports the whole host on the GPU
    boxes (hundreds of cores) while the container owns 16: ATen's default pool then oversubscribes them and every CPU-oracle
    step in a test runs an order of magnitude slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(host_cores())


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _collect_between_gpu_tests(request):
    """Trainers of earlier tests own HIP streams, events and captured graphs; Python frees them whenever the collector happens to run --
    possibly while a later test captures or replays a graph (hipGraphLaunch segfaulted in test_stream_sched_gpu when it ran right behind
    test_step_gpu, never alone).  Collect at the test boundary, with the device idle, instead."""
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    yield
