"""Loads the host-side simulator build of the kernel sources (tests/emu/libsat_emu.so) and binds the
C-ABI onto it.  TEST INFRASTRUCTURE ONLY — see tests/emu/hipemu.h."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
EMU_SO = os.path.join(_HERE, "emu", "libsat_emu.so")
_ops = None


def build_emu():
    csrc = os.path.join(_REPO, "stable_audio_tools_amd", "csrc")
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=csrc, check=True)


def emu_ops():
    global _ops
    if _ops is None:
        build_emu()
        from stable_audio_tools_amd import _lib, ops
        cdll = _lib.bind(ctypes.CDLL(EMU_SO))
        assert cdll.sat_is_simulator() == 1
        _ops = ops.SatOps(cdll)
    return _ops


def use_emu_ops():
    """Route the product nn.Modules through the simulator: replaces `stable_audio_tools_amd.ops.get_ops` (the one place the product
    resolves its kernel binding) with a function returning the simulator-bound SatOps.  Returns an undo callable.  Lives in the
    test tree on purpose — the product package has no such switch."""
    from stable_audio_tools_amd import ops
    original = ops.get_ops
    emu = emu_ops()
    ops.get_ops = lambda: emu
    return lambda: setattr(ops, "get_ops", original)
