"""Summarise rocprofv3 --pmc output (CSV counter_collection files or rocpd .db) per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
pats = sys.argv[2:] or [""]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if any(p in k for p in pats):
            acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.lower().startswith("counters_collection")]
    if not view:
        continue
    cols = [r[1] for r in db.execute(f"pragma table_info({view[0]})")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c][0]
    for k, cn, v in db.execute(f"select {kcol}, counter_name, value from {view[0]}"):
        if any(p in k for p in pats):
            acc[k[:60]][cn].append(float(v))
for k, d in acc.items():
    print(k)
    for cn, vals in sorted(d.items()):
        print(f"   {cn:36s} n={len(vals):3d} mean={sum(vals) / len(vals):.4g}")
