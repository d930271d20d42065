import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENE = os.path.join(ROOT, "real_time_path_tracing_with_spatiotemporal_filtering_amd", "scenes",
                     "CornellBox-Original-Merged.obj")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_policy: test_chain_gpu.py — run with the shipped kernel-selection thresholds")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.set_threads(min(8, os.cpu_count() or 1))
    # native frames of the faulting thread on SIGSEGV/SIGBUS/SIGABRT, then pytest's faulthandler (installed at
    # configure time, i.e. before this fixture) gets the signal: a crash in a thread without Python frames — an
    # oracle worker, a HIP runtime thread — is attributable from the log ("The round-2 abort")
    O.lib().oracle_install_crash_trace()
    return O


@pytest.fixture(scope="session")
def cornell(oracle):
    xyz, idx = oracle.load_obj(SCENE)
    return xyz, idx, oracle.flatten(xyz, idx)


@pytest.fixture(scope="session")
def hip_lib():
    """the product library; GPU tests fail (not skip) when it is missing"""
    # torch first: it ships its own ROCm runtime, and a process that initialises HIP through
    # librtpt_hip.so before torch does leaves torch unable to see the GPU ("No HIP GPUs are available");
    # in the other order both share one runtime (bench.py has the same order)
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    from real_time_path_tracing_with_spatiotemporal_filtering_amd import abi
    if not os.path.exists(abi.LIB_PATH):
        # a fresh checkout (the .so is git-ignored): build it the way __graft_entry__.build() does — hipcc
        # cross-compiles gfx950 without a GPU.  A failed build fails the tests; nothing falls back to the CPU.
        import shutil
        import subprocess
        if shutil.which("hipcc"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "real_time_path_tracing_with_spatiotemporal_filtering_amd", "csrc"), "-s"])
    abi.load()
    return abi


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)
