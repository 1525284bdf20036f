import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_only_run(config):
    """True for `-m "not gpu"` style runs on a machine without a GPU (counting devices does not initialise the runtime)."""
    if "gpu" in (config.option.markexpr or "") and "not gpu" not in config.option.markexpr:
        return False
    try:
        import torch
        return torch.cuda.device_count() == 0
    except Exception:
        return True


def pytest_cmdline_main(config):
    # The host-simulator tests run every lane as a host thread and spend their time in barrier wake-ups, not in compute: the CPU
    # suite is latency-bound and four xdist workers cut it from 7 to 2.5 minutes.  Only without a GPU (the GPU suite times
    # kernels and must own the device), only if xdist is there and the caller did not choose a worker count (-n 0 switches off).
    if getattr(config.option, "numprocesses", "absent") is None and _cpu_only_run(config) and not hasattr(config, "workerinput"):
        n = int(os.environ.get("FF_TEST_WORKERS", "4"))
        if n > 0:      # (what xdist's own hook, which has already run, does for -n N)
            config.option.numprocesses, config.option.dist, config.option.tx = n, "load", ["popen"] * n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not hasattr(config, "workerinput") and getattr(config.option, "numprocesses", None):
        # build the shared test libraries once, before the workers race for them
        try:
            from tests.hostsim import simlib
            simlib.build()
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {k: np.load(os.path.join(d, f"{k}.npz")) for k in
            ("g1_mcmc", "g2_slater", "g3_backflow", "g4_cnf", "g5_gsvmc", "g6_betavmc", "g7_3d")}
