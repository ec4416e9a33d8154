import torch, time, json, sys
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops()
dev = 'cuda'
torch.manual_seed(0)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, name) in [(2050, 1536, 6144, 'ff2'), (2050, 1536, 1536, 'out'), (2050, 4608, 1536, 'qkv'), (4100, 1536, 6144, 'ff2_b4')]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = o.gemm_bf16(a, b, bias=bias, res=res, epilogue=o.EPI_RES, out=out).float().clone()
    row = {'shape': name, 'plain_us': timeit(lambda: o.gemm_bf16(a, b, bias=bias, res=res, epilogue=o.EPI_RES, out=out))}
    for S in (2, 3, 4):
        y = o.gemm_bf16_splitk(a, b, S, bias=bias, res=res, out=out).float()
        row[f'S{S}_us'] = timeit(lambda: o.gemm_bf16_splitk(a, b, S, bias=bias, res=res, out=out))
        row[f'S{S}_err'] = float((y - ref).abs().max() / ref.abs().max())
    print(json.dumps(row))
