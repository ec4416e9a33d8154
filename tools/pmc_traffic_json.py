"""FETCH_SIZE / WRITE_SIZE passes of `bench.py` (tools/collect_profiles.sh) -> per-kernel HBM bytes per launch as JSON.
usage: pmc_traffic_json.py <FETCH_SIZE dir> <WRITE_SIZE dir> <out.json> <source note>
gfx950 correction (MI355X_MICROARCH.md, HBM section; confirmed in profiles/r01g_pmc_traffic.json on a streaming kernel of known
traffic): FETCH_SIZE tallies 128-byte fabric read requests at 64 bytes -> x2; WRITE_SIZE needs none.  rocprofv3 reports both in KiB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(root):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            name = re.sub(r"^void ", "", k).split("<")[0].split("(")[0]
            acc[name].append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
unit = float(os.environ.get("PMC_UNIT_BYTES", "1024"))     # counter unit in bytes (rocprofv3 derived FETCH_SIZE / WRITE_SIZE: KiB)
doc = {"source": sys.argv[4] if len(sys.argv) > 4 else "", "unit_bytes": unit, "steps_profiled": int(os.environ.get("PMC_STEPS", "2")),
       "correction": "gfx950: FETCH_SIZE x2 (128-byte fabric reads tallied at 64 bytes); WRITE_SIZE as reported", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("sat_"):
        continue
    f = sum(fetch.get(k, [0.0])) / max(1, len(fetch.get(k, [])))
    w = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [])))
    doc["kernels"][k] = {"launches": len(fetch.get(k, [])), "fetch_bytes_per_launch_raw": f * unit, "write_bytes_per_launch": w * unit,
                         "hbm_bytes_per_launch": 2 * f * unit + w * unit}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in doc["kernels"].items()}))
