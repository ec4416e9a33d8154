"""ResidualUnit forward at the top Oobleck level (C = 128): fused single launch vs k7q + k1 launches.  MI355X only."""
import json, sys, torch
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops()
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for (c, t, dil) in [(128, 2097152, 3), (128, 1048576, 1)]:
    x = torch.randn(1, c, t, device='cuda'); w7 = torch.randn(c, c, 7, device='cuda') / (7 * c) ** 0.5; w1 = torch.randn(c, c, 1, device='cuda') / c ** 0.5
    b1 = torch.randn(c, device='cuda'); b2 = torch.randn(c, device='cuda')
    s = [torch.randn(c, device='cuda') * 0.3 for _ in range(6)]
    w7q, w1q = o.pack_k7q(w7), o.pack_k7q(w1)
    w7p = o.pack_bf16x3(w7, 0, 1, q=True); w1p = o.pack_bf16x3(w1, 0, 1)
    row = {"C": c, "T": t, "dil": dil}
    def unfused(emit):
        h = o.conv1d_bf16x3(x, w7p, c, 7, 1, dil, 3 * dil, bias=b1, snake=(s[0], s[1]))
        return o.conv1d_bf16x3(h, w1p, c, 1, 1, 1, 0, bias=b2, snake=(s[2], s[3]), res=x, emit=({"snake": (s[4], s[5])} if emit else None))
    row["unfused_us"] = round(timeit(lambda: unfused(False)), 1)
    row["unfused_emit_us"] = round(timeit(lambda: unfused(True)), 1)
    row["k7q_only_us"] = round(timeit(lambda: o.conv1d_bf16x3(x, w7p, c, 7, 1, dil, 3 * dil, bias=b1, snake=(s[0], s[1]))), 1)
    for keep_h in (True, False):
        for emit in (False, True):
            row[f"fused_h{int(keep_h)}_e{int(emit)}_us"] = round(timeit(lambda: o.residual_unit_fwd(x, (s[0], s[1]), w7q, b1, (s[2], s[3]), w1q, b2, 7, dil, keep_h=keep_h,
                                                                                                       emit=({"snake": (s[4], s[5])} if emit else None))), 1)
    print(json.dumps(row), flush=True)
