import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
dev="cuda"; lib=L.lib(); bf=torch.bfloat16
st=lambda: torch.cuda.current_stream().cuda_stream
def t(d, n=30):
    for _ in range(3): lib.fh_gemm(d,1,st())
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.fh_gemm(d,1,st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for M,N in ((15936,480),(15936,512),(65536,512),(2048,512)):
  for K in (32,128,480,960,1920,3840):
    X=torch.randn(M,K,device=dev).to(bf); W=torch.randn(N,K,device=dev).to(bf); C=torch.empty(M,N,device=dev,dtype=bf)
    d=L.GemmDesc(); d.X=L.operand(X.data_ptr(),0,K); d.Y=L.operand(W.data_ptr(),0,K); d.C=C.data_ptr()
    d.M,d.N,d.Kseg,d.ldc,d.groups,d.splitk=M,N,K,N,1,1; d.ep=L.epilogue()
    r=[]
    for path in (0,1):
        lib.fh_gemm_set_path(path); r.append(t(d))
    print(f"M={M} N={N} K={K:5d}: reg {r[0]:8.1f} us ({2.0*M*N*K/r[0]/1e6:6.0f} TF/s)  dma {r[1]:8.1f} us ({2.0*M*N*K/r[1]/1e6:6.0f} TF/s)", flush=True)
