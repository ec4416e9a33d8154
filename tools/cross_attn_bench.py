"""Cross-attention (130 keys, GQA 24:12) forward and backward, bf16 planes: the short-key kernels (csrc/attention_cross.h) against the
general flash-style kernels on the same operands — microseconds per launch (HIP-graph replay of 20 launches: no host dispatch in the figure), achieved
TFLOP/s (4 N M d per head forward, 10 N M d backward: the algorithmic count, recomputation not credited) and the error against SDPA.
    python tools/cross_attn_bench.py            one JSON line per (shape, implementation)"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
torch.manual_seed(0)


def timeit(f, n=100):
    """microseconds per call of f: 20 calls captured into ONE HIP graph, the graph replayed n / 20 times — a launch of a few microseconds
    is otherwise timed together with the host's ~10 us of Python / ctypes dispatch per call (the first version of this tool did)."""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                f()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    reps = max(1, n // 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * reps) * 1e3


shapes = [(2, 24, 12, 1025, 130), (4, 24, 12, 1025, 130), (16, 24, 12, 1025, 130), (2, 24, 12, 6145, 130)]
for (b, h, hkv, nq, nk) in shapes:
    q = torch.randn(b, h, nq, 64, device='cuda').bfloat16()
    k = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    v = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    do = torch.randn(b, nq, h * 64, device='cuda').bfloat16()
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qr, kr.repeat_interleave(h // hkv, 1), vr.repeat_interleave(h // hkv, 1), scale=0.125)
    ref = ref.permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    ref.backward(do.float())
    for cross in (True, False):
        o.cross_kernels = cross
        out, lse, planes = o.attention(q, k, v, 0.125, return_planes=True)
        dq, dk, dv = o.attention_bwd(planes, out, do, lse, 0.125, hkv, nk)

        def rel(a, r):
            return float((a.float() - r).abs().max() / r.abs().max())
        errs = {"o": rel(out, ref.detach()), "dq": rel(dq, qr.grad), "dk": rel(dk, kr.grad), "dv": rel(dv, vr.grad)}
        fwd = timeit(lambda: o.attention_planes(planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0], nq, nk, 0.125))
        # the backward as the autograd node runs it (row dot + dO planes + the gradient kernels) and the gradient kernels alone
        bwd_all = timeit(lambda: o.attention_bwd(planes, out, do, lse, 0.125, hkv, nk), n=50)
        import ctypes
        dsum = torch.empty(b, h, nq, dtype=torch.float32, device='cuda')
        gp = o.attn_planes(do.view(b, nq, h, 64).permute(0, 2, 1, 3), row_major=True, transposed=True)
        qp, kp, vp = planes["q"], planes["k"], planes["v"]
        ptrs = [qp["rm"], kp["rm"], vp["rm"], kp["tr"], qp["tr"], gp["rm"], gp["tr"], (None, None)]
        arr = (ctypes.c_void_p * 16)(*[(x.data_ptr() if x is not None else None) for pair in ptrs for x in pair])
        dq2, dk2, dv2 = torch.empty_like(dq), torch.empty_like(dk), torch.empty_like(dv)
        if cross:
            nbytes = int(o.lib.sat_attention_cross_bwd_ws(b, h, hkv, nq, nk))
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device='cuda')

            def kern():
                o._chk(o.lib.sat_attention_cross_bwd(arr, lse.data_ptr(), dsum.data_ptr(), dq2.data_ptr(), dk2.data_ptr(), dv2.data_ptr(),
                                                     ws.data_ptr(), nbytes, b, h, hkv, nq, nk, qp["np"], kp["np"], 64, 0.125, o._stream(q)))
        else:
            def kern():
                o._chk(o.lib.sat_attention_bwd(arr, lse.data_ptr(), dsum.data_ptr(), dq2.data_ptr(), dk2.data_ptr(), dv2.data_ptr(), b, h, hkv,
                                               nq, nk, qp["np"], kp["np"], 64, 0.125, 1, o._stream(q)))
        bwd = timeit(kern)
        fl = 4.0 * b * h * nq * nk * 64
        print(json.dumps({"shape": [b, h, hkv, nq, nk], "kernels": "short-key" if cross else "general", "fwd_us": round(fwd, 2),
                          "fwd_tflops": round(fl / fwd * 1e-6, 1), "fwd_frac_of_2500": round(fl / fwd * 1e-6 / 2500, 4),
                          "bwd_kernels_us": round(bwd, 2), "bwd_frac_of_2500": round(2.5 * fl / bwd * 1e-6 / 2500, 4),
                          "bwd_node_us": round(bwd_all, 2), "rel_err": {k_: round(v_, 5) for k_, v_ in errs.items()}}), flush=True)
