import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


NATIVE = [os.path.join(ROOT, "mujoco_maze_amd", "csrc", "libmazestep.so"), os.path.join(ROOT, "oracle", "libmzo.so"),
          os.path.join(ROOT, "tests", "emu", "libantemu.so")]


@pytest.fixture(scope="session", autouse=True)
def _native_artefacts():
    """A fresh checkout has no binaries (they are git-ignored): build them once.  Existing ones are used as they are
    (the GPU box runs the prebuilt files that travelled with the snapshot)."""
    if not all(os.path.exists(p) for p in NATIVE):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()
