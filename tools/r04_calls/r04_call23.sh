#!/bin/bash
set -u
R=$(pwd); mkdir -p gpurun_out; export PYTHONPATH=$R/qwen-image-finetune_amd:$R
python - <<'P'
import torch, sys
from qflux_amd import ops, _lib as L
BF=torch.bfloat16
def run(M,N,K,tiles=None):
    if tiles: assert L.lib.qfx_gemm_tune(tiles.encode(), None)==0
    g=torch.Generator().manual_seed(1)
    a=torch.randn(M,K,generator=g).to(BF).cuda(); b=(torch.randn(N,K,generator=g)*0.1).to(BF).cuda()
    out=ops.gemm(a,b); torch.cuda.synchronize()
    ref=(a.float()@b.float().t())
    d=(out.float()-ref).abs()
    bad=(d>0.05+0.02*ref.abs()).nonzero()
    rows=torch.unique(bad[:,0]%64).tolist() if bad.shape[0] else []
    print(M,N,K,tiles,'nbad',bad.shape[0],'rows%64',rows[:24])
for tiles in ("256x128","256x256"):
    for (M,N,K) in [(4096,2048,64),(4096,2048,128),(4096,2048,192),(2048,3072,64),(2048,3072,128),(2432,3072,3072),(4096,4096,64),(8192,4096,256)]:
        run(M,N,K,tiles)
P
