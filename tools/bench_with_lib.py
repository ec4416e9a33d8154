"""EXPERIMENT helper: run bench.py against an alternative build of the library (tools/exp/*.so: same C-ABI, other compile flags).
    python tools/bench_with_lib.py tools/exp/libsat_amd_all_noslp.so --no-secondary --no-real-step ..."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_audio_tools_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
