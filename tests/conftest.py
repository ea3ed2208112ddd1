import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    # on the GPU box the prebuilt .so files travel with the snapshot; build() is a no-op when they are fresh
    try:
        g.build_oracle()
    except Exception as exc:  # pragma: no cover
        pytest.skip(f"cannot build the oracle: {exc}")
    return g


@pytest.fixture(scope="session")
def pkg(built):
    # torch ships its own copy of the HIP / ROCr runtime; when a test wants torch tensors on the GPU next to the library's handles
    # (device-side insert, torch views of HBM buffers), torch has to bring its runtime up first, as bench.py does
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:  # pragma: no cover
        pass
    import aircombat_selfplay_amd as m
    return m


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as O
    O.lib()
    return O
