"""Randomised shapes through the env-selected lean arms (attention forward + backward, projection GEMMs on tiles 4 / 7 / 8) against torch
references — the simulator by default (how the arms were checked at the end of round 4: 1150 attention and 3450 GEMM cases), `gpu` as the
third argument for the gfx950 library.   usage: python tools/fuzz_lean.py [seed] [seconds] [gpu]"""
import os, sys, random, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import torch.nn.functional as F
from golden_util import rel_err
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
if GPU:
    from stable_audio_tools_amd import ops as O
    ops = O.get_ops()
else:
    from emu_util import emu_ops
    ops = emu_ops()
DEV = "cuda" if GPU else "cpu"
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 0)
os.environ.update(SAT_ATTN_LEAN="1", SAT_ATTN_BWD_LEAN="1", SAT_GEMM_LEAN="1", SAT_LN_LEAN="1")
t_end=time.time()+float(sys.argv[2]) if len(sys.argv)>2 else time.time()+300
na=ng=0
while time.time()<t_end:
    # attention
    hkv=random.choice([1,2,3]); rep=random.choice([1,2]); h=hkv*rep; b=random.choice([1,2])
    nq=random.choice([1,2,31,32,33,63,64,65,100,127,128,129,200]); nk=random.choice([1,2,17,31,32,33,63,64,65,96,127,128,129,191,192,193,260])
    g=torch.Generator().manual_seed(random.randrange(1<<30))
    q=torch.randn(b,h,nq,64,generator=g).bfloat16().to(DEV); k=torch.randn(b,hkv,nk,64,generator=g).bfloat16().to(DEV); v=torch.randn(b,hkv,nk,64,generator=g).bfloat16().to(DEV)
    do=torch.randn(b,nq,h*64,generator=g).bfloat16().to(DEV)
    o,lse,pl=ops.attention(q,k,v,0.125,return_planes=True)
    dq,dk,dv=ops.attention_bwd(pl,o,do,lse,0.125,hkv,nk)
    qr,kr,vr=(t.float().detach().requires_grad_(True) for t in (q,k,v))
    ref=F.scaled_dot_product_attention(qr,kr.repeat_interleave(rep,1),vr.repeat_interleave(rep,1),scale=0.125).permute(0,2,1,3).reshape(b,nq,h*64)
    ref.backward(do.float())
    for name,a,r in (("o",o,ref.detach()),("dq",dq,qr.grad),("dk",dk,kr.grad),("dv",dv,vr.grad)):
        err=(a.float()-r).abs().max().item(); sc=max(r.abs().max().item(),1e-2)
        assert err<=3e-2*sc+8e-3, ("attn",(b,h,hkv,nq,nk),name,err,sc)
    na+=1
    # gemm
    m=random.choice([8,16,33,64,130,257,300,513]); n=8*random.randint(1,70); kk=8*random.randint(1,90)
    a=torch.randn(m,kk,generator=g).bfloat16().to(DEV); bb=torch.randn(n,kk,generator=g).bfloat16().to(DEV)
    ref=(a.float()@bb.float().t())
    bias=torch.randn(n,generator=g).to(DEV); res=torch.randn(m,n,generator=g).bfloat16().to(DEV)
    for tile in (4,7,8):
        ops.gemm_tile=tile
        try:
            sp=random.choice([1,1,2,3])
            assert rel_err(ops.gemm_bf16(a,bb,out_dtype=torch.float32,splits=sp),ref)<1e-5, ("gemm",tile,m,n,kk,sp)
            assert rel_err(ops.gemm_bf16(a,bb,bias=bias,res=res,epilogue=ops.EPI_RES).float(),ref+bias+res.float())<6e-3, ("gemm res",tile,m,n,kk)
            if n%16==0:
                c=ops.gemm_bf16(a,bb,bias=bias,epilogue=ops.EPI_SWIGLU,out_dtype=torch.float32)
                full=ref+bias
                assert rel_err(c,full[:,:n//2]*F.silu(full[:,n//2:]))<1e-5, ("gemm glu",tile,m,n,kk)
        finally:
            ops.gemm_tile=None
    ng+=1
print("fuzz ok: attention cases",na,"gemm cases",ng,"lean launches",[ops.lib.sat_lean_launches(i) for i in range(5)])
