import json, sys, os, torch
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops(); o.k7_planes = True; o.k7_planes_min_cin = 1
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (c, t, dil) in [(128, 2097152, 1), (256, 262144, 3), (1024, 8192, 1)]:
    x = torch.randn(1, c, t, device='cuda'); w = torch.randn(c, c, 7, device='cuda') / (7 * c) ** 0.5
    bias = torch.randn(c, device='cuda'); la = torch.randn(c, device='cuda') * 0.3; lb = torch.randn(c, device='cuda') * 0.3
    wq = o.pack_bf16x3(w, 0, 1, q=True)
    x2 = torch.randn(1, c, t, device='cuda')
    print(c, t, "variant", os.environ.get("SAT_K7Q_VARIANT"), "stagger", os.environ.get("SAT_K7Q_STAGGER"),
          "fwd", round(timeit(lambda: o.conv1d_bf16x3(x, wq, c, 7, 1, dil, 3 * dil, bias=bias, snake=(la, lb))), 1),
          "dgrad", round(timeit(lambda: o.conv1d_bf16x3(x, wq, c, 7, 1, dil, 3 * dil, dsnake=(x2, la, lb), res=x2)), 1), flush=True)
