"""TEST INFRASTRUCTURE ONLY — recipe that stages the REFERENCE itself for the GPU box.

The reference (Stability-AI/stable-audio-tools) is pure Python: there is nothing to compile, but its checkout lives at
/root/reference, which exists only in the build container.  BASELINE.json's north_star asks for "the reference timed on the
host cores of the same box in the same run" and for the drop-in (`generate_diffusion_cond` on the native modules) on the
GPU, so this recipe copies the importable part of the checkout —

    /root/reference/stable_audio_tools/**/*.py, *.json      (1.2 MB; no bytecode, no notebooks)

— where it lies, unmodified, into the git-ignored `oracle/_ref/stable_audio_tools/`.  `oracle/_ref/` is listed in
.gitignore (the reference's sources never enter this repository's history) but NOT in .gpurunignore, so it travels to
the GPU box with the working tree exactly like the built `libsat_amd.so`.  `oracle/refimport.py` falls back to it when
/root/reference is absent.  `__graft_entry__.build()` runs this whenever /root/reference is present.

Only the checkers use the staged tree: tests/ (drop-in, oracle pinning), bench.py's `cpu_baseline` leg
(kind "reference") — never the product package.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.environ.get("SAT_REFERENCE_ROOT", "/root/reference")
DST_ROOT = os.path.join(HERE, "_ref")
KEEP = (".py", ".json")


def stage(src_root=SRC_ROOT, dst_root=DST_ROOT, verbose=False):
    """Returns the number of files staged (0 when the checkout is absent: the GPU box uses what was staged here)."""
    src = os.path.join(src_root, "stable_audio_tools")
    if not os.path.isdir(src) or os.path.realpath(src_root) == os.path.realpath(dst_root):
        return 0
    dst = os.path.join(dst_root, "stable_audio_tools")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    n = 0
    for dirpath, dirnames, filenames in os.walk(src):
        dirnames[:] = [d for d in dirnames if d != "__pycache__"]
        rel = os.path.relpath(dirpath, src)
        out = os.path.join(dst, rel) if rel != "." else dst
        os.makedirs(out, exist_ok=True)
        for f in filenames:
            if f.endswith(KEEP):
                shutil.copyfile(os.path.join(dirpath, f), os.path.join(out, f))
                n += 1
    with open(os.path.join(dst_root, "STAGED_FROM"), "w") as fh:
        fh.write(f"{src_root}\n{n} files (*.py, *.json) copied unmodified by oracle/stage_ref.py — test infrastructure, git-ignored\n")
    if verbose:
        print(f"stage_ref: {n} files -> {dst}")
    return n


if __name__ == "__main__":
    sys.exit(0 if stage(verbose=True) else 1)
