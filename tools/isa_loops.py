"""Static view of the kernels' main loops (no GPU needed): compiles a csrc/*.hip file to gfx950 assembly and prints, per kernel and per
innermost loop that holds MFMAs, the instruction mix — all / MFMA / VALU / scalar / branches / LDS / VMEM — and instructions per MFMA.
Round 4's reading of the SQ counters ("one instruction per four cycles and wave") makes `all x 4 cycles x waves per SIMD` the issue
time of a loop iteration, to be compared with `MFMA x 32 cycles` (32 x 32 x 16 bf16) — profiles/EXPERIMENTS.md, last sections.

usage: python tools/isa_loops.py attention.hip [kernel-name-substring]     (files are looked up in stable_audio_tools_amd/csrc)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stable_audio_tools_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -fno-slp-vectorize -S --cuda-device-only".split()


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global", "buffer", "flat", "scratch")):
        return "vmem"
    return "other"


def loops_of(asm, pattern):
    lines = asm.split("\n")
    kernels, cur = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = [i, None]
        if l.startswith(".Lfunc_end") and cur:
            kernels[cur][1] = i
            cur = None
    for k, (a, b) in kernels.items():
        if pattern and pattern not in k:
            continue
        loops, curloop = collections.OrderedDict(), None
        for l in lines[a:b]:
            if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb"):
                if "Loop Header" in l and "=>" in l:
                    curloop = re.match(r"^\.L(BB\d+_\d+)", l).group(1)
                elif "in Loop: Header=" in l:
                    curloop = re.search(r"Header=(BB\d+_\d+)", l).group(1)
                else:
                    curloop = None
                continue
            t = l.strip()
            if not curloop or not t or t.startswith((".", ";")):
                continue
            c = loops.setdefault(curloop, collections.Counter())
            c["all"] += 1
            c[classify(t.split()[0])] += 1
        for lp, c in loops.items():
            if c["mfma"] >= 8:
                yield k, lp, c


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = sys.argv[1] if os.path.isabs(sys.argv[1]) else os.path.join(CSRC, sys.argv[1])
    pattern = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, src, "-o", out], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        asm = open(out).read()
    print(f"{'kernel':72s} {'loop':10s} {'all':>5s} {'mfma':>5s} {'valu':>5s} {'salu':>5s} {'br':>3s} {'lds':>4s} {'vmem':>4s} {'all/mfma':>8s}")
    for k, lp, c in loops_of(asm, pattern):
        print(f"{k[:72]:72s} {lp:10s} {c['all']:5d} {c['mfma']:5d} {c['valu']:5d} {c['salu']:5d} {c['branch']:3d} {c['lds']:4d} {c['vmem']:4d} {c['all'] / c['mfma']:8.1f}")
    print("(blocks of a loop that the compiler placed out of line — cold paths such as the K-tail staging — are counted with it)")


if __name__ == "__main__":
    main()
