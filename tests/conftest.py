import os
import sys

import pytest

# idle OpenMP workers of the oracle must sleep, not spin (a GPU box gives the container a CPU quota far below the CPUs it shows);
# has to be in the environment before libgomp is loaded
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# an abort() raised on one of the runtime's own threads (ROCr reporting a GPU memory fault, a runtime assertion) says which thread and
# from where (capi.hip: install_abort_backtrace); read when the library is loaded.  pytest.ini's --capture=sys keeps fd 2 uncaptured,
# so that text -- and ROCr's own message in front of it -- reaches the log of whoever runs the suite.
os.environ.setdefault("CPIR_ABORT_BACKTRACE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure; compiled with gcc on first use)"""
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def native():
    """the product's C ABI library; built on demand here so a fresh checkout can run the CPU suite"""
    from chalametpir_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native.load()


@pytest.fixture(scope="session")
def device(native):
    """cpir_device 0; GPU tests FAIL (not skip) if the HIP library cannot open a device"""
    import chalametpir_amd as cp

    # torch shares the process' HIP runtime (chalametpir_amd/_native.py); initialise its context first, as bench.py does,
    # instead of lazily in the middle of the suite
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    return cp.Device(0)


@pytest.fixture(scope="session")
def group_devices(device):
    """the device list of a GROUP of k shards on THIS box: distinct devices wherever the box has them (shard i on device i mod the number
    of visible devices), the one GPU listed k times on a one-GPU box.  The group tests therefore arm themselves on a multi-GPU node --
    peer copies between different devices, per-device streams and allocations, the root's sum kernel behind events of other devices --
    and pass as the one-device case elsewhere; they never skip."""
    import torch

    import chalametpir_amd as cp

    n_vis = max(1, torch.cuda.device_count())
    opened = {0: device}

    def pick(k):
        out = []
        for i in range(k):
            o = i % n_vis
            if o not in opened:
                opened[o] = cp.Device(o)
            out.append(opened[o])
        return out

    pick.visible = n_vis
    return pick


@pytest.fixture(autouse=True)
def _tuning_defaults_between_tests(request):
    """the library's tuning keys are process-wide: whatever a test (or a fixture of its module) flipped is put back to the defaults
    after it, so that no test depends on which tests ran before it.  (Only for tests that load the library anyway.)"""
    yield
    if "native" in request.fixturenames or "device" in request.fixturenames:
        import chalametpir_amd as cp

        cp.tuning_reset()
