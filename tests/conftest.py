import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {f[:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".npz")}


def free_port():
    """A TCP port the kernel just handed out (bind to 0), for torch.distributed rendezvous in multi-process tests."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.fixture(autouse=True)
def _free_big_tmp_files(tmp_path):
    """The multi-process tests hand their results over as torch.save files of 150-450 MB each (flat parameters, moments); pytest keeps every
    test's tmp_path until the session ends (and the last three sessions' after that): one run of the GPU suite left 68 GB in /tmp of a
    79-GB box, a second run in the same call died with "No space left on device" inside a worker's torch.save.  Files above 1 MB go
    when their test is over."""
    yield
    try:
        for root, _, files in os.walk(str(tmp_path)):
            for f in files:
                fp = os.path.join(root, f)
                if os.path.isfile(fp) and not os.path.islink(fp) and os.path.getsize(fp) > (1 << 20):
                    os.remove(fp)
    except Exception:          # noqa: BLE001  (cleanup only)
        pass
