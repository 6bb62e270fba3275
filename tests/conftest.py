import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def host_cores() -> int:
    """CPU threads this process may really use: the cgroup quota if one is set, else the affinity mask, capped at 16.
    (os.cpu_count() reports the whole host -- 256 threads on a GPU box whose lease has 16 cores; sizing torch's pool by it
    oversubscribes the share and makes every CPU-oracle run crawl.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


@pytest.fixture(scope="session", autouse=True)
def _torch_threads():
    import torch

    torch.set_num_threads(host_cores())
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
