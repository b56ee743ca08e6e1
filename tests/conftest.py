import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def _cpu_share():
    """CPUs this process may actually use: the cgroup quota where there is one (the GPU boxes expose all 256 host CPUs to
    os.cpu_count() / sched_getaffinity but grant a 16-CPU quota: torch then starts 128 intra-op threads and the CPU oracle -- the
    float64 side of every parity test -- runs throttled and several times slower), else the affinity mask"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(_cpu_share())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
