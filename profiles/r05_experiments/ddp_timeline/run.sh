#!/bin/bash
# round 5, call 5: fused k1 backward with LDS-DMA staging (A / B against the separate kernels), STFT mid instance, DDP timelines (1-rank RCCL)
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call5; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_stft_parity.py tests/test_vae_parity.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
tail -n 3 $OUT/tests.log
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph --no-dit-train"
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN >> $OUT/vae_fused.json 2>> $OUT/ab.err
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN --ops-set ru_k1_fused=0 >> $OUT/vae_separate.json 2>> $OUT/ab.err
done
timeout 300 python bench.py --steps 5 --warmup 2 $GEN --ddp-single-rank > $OUT/ddp_single_rank_vae.json 2>> $OUT/ab.err
timeout 300 python bench.py --workload dit_train --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --ddp-single-rank > $OUT/ddp_single_rank_dit.json 2>> $OUT/ab.err
timeout 300 python bench.py --workload dit_train --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dit_train_no_ddp.json 2>> $OUT/ab.err
python - <<PY
import json
for f in ("vae_fused","vae_separate","ddp_single_rank_vae","ddp_single_rank_dit","dit_train_no_ddp"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2))
        for k in (r["roofline"].get("all_conv_kernels") or [])[:5]: print("     ", k["kernel"], k.get("launches"), k.get("total_ms"), k.get("frac"))
        d=(r.get("config") or {}).get("ddp")
        if d and d.get("timeline"):
            tl=d["timeline"]; print("   ddp", {k:v for k,v in d.items() if k not in ("timeline","timeline_note")}); print("   backward_ms", tl["backward_ms"], "buckets", [(x["bucket"], x["from_hook"], x["ready_ms"], x["done_ms"]) for x in tl["buckets"]][:12])
PY
