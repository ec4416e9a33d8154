"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_WAIT_INST_ANY SQ_WAVE_CYCLES): sums over every dispatch of a kernel, then
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (four SIMDs per CU; the round-3/4 summaries' convention)
  valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA
  wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
Usage: pmc_mfma_busy.py <rocprof output dir> <description> -> JSON on stdout."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

root, desc = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))


def add(k, cn, v):
    if "sat_" not in k:
        return
    k = k.replace("void ", "")[:90]
    acc[k][cn] += float(v)
    cnt[k][cn] += 1


for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        add(row.get("Kernel_Name", ""), row["Counter_Name"], row["Counter_Value"])
for f in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.lower().startswith("counters_collection")]
    if not view:
        continue
    cols = [r[1] for r in db.execute(f"pragma table_info({view[0]})")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c][0]
    for k, cn, v in db.execute(f"select {kcol}, counter_name, value from {view[0]}"):
        add(k, cn, v)
out = {"source": desc, "convention": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), sums over all dispatches of the kernel", "kernels": {}}
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0.0)):
    busy, cu = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("SQ_BUSY_CU_CYCLES", 0.0)
    if cu <= 0:
        continue
    e = {"dispatches": cnt[k].get("SQ_BUSY_CU_CYCLES", 0), "busy_cu_cycles": cu, "mfma_busy": round(busy / (4 * cu), 4)}
    if d.get("SQ_INSTS_MFMA", 0) > 0:
        e["valu_per_mfma"] = round(d.get("SQ_INSTS_VALU", 0.0) / d["SQ_INSTS_MFMA"], 2)
    if d.get("SQ_WAVE_CYCLES", 0) > 0:
        e["wait_inst_frac"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 3)
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
