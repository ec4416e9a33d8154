import os
import sys

import pytest

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (_REPO, os.path.join(_REPO, "oracle"), os.path.join(_REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu():
    """SatOps bound to the host-side simulator build of the kernel sources (CPU tensors)."""
    from emu_util import emu_ops
    return emu_ops()


@pytest.fixture()
def emu_modules(emu):
    """Route the product nn.Modules through the simulator for the duration of one CPU test."""
    from emu_util import use_emu_ops
    undo = use_emu_ops()
    yield emu
    undo()


@pytest.fixture(scope="session")
def hip():
    """The product ops singleton (gfx950 library) — GPU tests only."""
    import torch
    assert torch.cuda.is_available(), "GPU test without a GPU"
    from stable_audio_tools_amd import ops
    assert ops.get_ops.__module__ == "stable_audio_tools_amd.ops", "a simulator substitution leaked into a GPU test"
    o = ops.get_ops()
    assert not o.simulator
    return o
