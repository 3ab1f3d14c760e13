import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def ab_lib(dev):
    """Routes the test's ops calls to the -DLTXK_AB measurement build (libltxk_ab.so), whose launch-form switches are
    environment variables; the product library reads none (csrc/common.h)."""
    from mlx_video_amd import _lib
    with _lib.use_library(_lib.AB_LIB_PATH) as lib:
        yield lib


def _heartbeat():
    """On the GPU box a CPU-oracle comparison can run for minutes without output; a line per minute naming the test
    that is running tells a long comparison from a hang."""
    import threading
    import time
    t0 = time.time()

    def beat():
        while True:
            time.sleep(60)
            print(f"[progress] t={time.time() - t0:.0f}s running {os.environ.get('PYTEST_CURRENT_TEST', '?')}", file=sys.stderr, flush=True)

    threading.Thread(target=beat, daemon=True).start()


def pytest_sessionstart(session):
    if os.environ.get("GRAFT_REPO_ROOT"):
        _heartbeat()
    try:        # the CPU oracle: one thread per core this process may use (a GPU box reports the whole host's cores)
        import torch
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    except Exception:
        pass


def pytest_sessionfinish(session, exitstatus):
    try:
        import parity
        parity.dump()
    except Exception:
        pass
