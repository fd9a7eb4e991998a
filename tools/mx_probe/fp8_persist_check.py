import os, sys, torch, ctypes as C
sys.path.insert(0, "qwen-image-finetune_amd")
from qflux_amd import ops, _lib as L
BF=torch.bfloat16; dev="cuda"
M,N,K=2432,3072,12288
x=torch.randn(M,K,device=dev).to(BF); w=(torch.randn(N,K,device=dev)*0.02).to(BF)
xq,xs=ops.quant_mxfp8(x); wq,ws=ops.quant_mxfp8(w)
y=torch.empty(M,N,dtype=BF,device=dev)
def mk():
    f=L.GemmFp8Args(); g=f.g
    g.A1,g.B1,g.lda1,g.ldb1,g.K1=xq.data_ptr(),wq.data_ptr(),K,K,K
    g.M,g.N=M,N; g.C,g.ldc=y.data_ptr(),N; g.rows_per_batch=M; g.epi=0
    f.sa,f.sb=xs.data_ptr(),ws.data_ptr()
    return f
f=mk()
st=torch.cuda.current_stream().cuda_stream
def t(fn,n=20):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
a=t(lambda: L.check(L.lib.qfx_gemm_mxfp8_grouped(C.byref(f),1,st),"g"))
ref=ops.mxfp8_dequant(xq,xs)@ops.mxfp8_dequant(wq,ws).t()
print("grouped persistent us",a, "TF", 2*M*N*K/a/1e6, "err", ((y.float()-ref).abs().max()/ref.abs().max()).item())
b=t(lambda: ops.gemm(x,w,out=y))
print("bf16 us",b,"TF",2*M*N*K/b/1e6)
