import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_cpu_side():
    """The CPU oracle and the GPU-free host-logic library are built on demand (seconds)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liborb_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "vi_slam_amd", "libvslam_host.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vi_slam_amd", "csrc"),
                               "../libvslam_host.so"])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def kp_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)
