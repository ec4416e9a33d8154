"""A/B: bias-gradient row sums inside the pipelined k7 weight-gradient kernel vs a separate sat_rowsum pass (SAT_WG_ROWSUM=0)."""
import json, os, sys
import torch
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops()
def timeit(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (c, t, dil) in [(128, 2097152, 1), (128, 2097152, 9), (256, 1048576, 3), (512, 262144, 9), (1024, 65536, 1)]:
    dy = torch.randn(1, c, t, device='cuda'); x = torch.randn(1, c, t, device='cuda')
    la = torch.randn(c, device='cuda') * 0.1; lb = torch.randn(c, device='cuda') * 0.1
    row = {"C": c, "T": t, "dil": dil}
    for mode in ("1", "0"):
        os.environ["SAT_WG_ROWSUM"] = mode
        row["fused_ms" if mode == "1" else "separate_ms"] = round(timeit(lambda: o.conv_wgrad7_bf16x3(dy, x, dil, 3 * dil, snake=(la, lb), dy_rowsum=True)), 3)
    row["no_rowsum_ms"] = round(timeit(lambda: o.conv_wgrad7_bf16x3(dy, x, dil, 3 * dil, snake=(la, lb))), 3)
    print(json.dumps(row), flush=True)
