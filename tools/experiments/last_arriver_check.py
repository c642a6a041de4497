import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from diffusionhandles_amd import _lib
L=_lib.lib(); dev=torch.device("cuda:0")
P=lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
part=torch.empty(48<<20,dtype=torch.float32,device=dev)
def gemm(A,lda,W,M,N,K,mode,geo,bias,R):
    C=torch.full((M,N),float("nan"),dtype=torch.float16,device=dev)
    _lib.check(L.dh_dbg_gemm(0,P(A),lda,P(W),M,N,K,mode,*geo,P(bias),P(None),0,1,P(R),N,P(C),N,0,P(part),part.numel(),_lib.stream_ptr()),"gemm")
    torch.cuda.synchronize(); return C
g=torch.Generator(device=dev).manual_seed(1)
bad=0
for (M,N,K,conv,H) in [(256,1280,1280*9,True,16),(64,1280,1280*9,True,8),(1024,640,640*9,True,32),(256,1280,5120,False,0),(256,1280,11520,False,0),(1024,1280,5760,False,0),(300,320,4096,False,0)]:
    if conv:
        Cin=K//9; A=torch.randn(M,Cin,generator=g,device=dev).half(); lda=Cin; geo=(H,H,Cin,H,H,1,0); mode=1
    else:
        A=torch.randn(M,K,generator=g,device=dev).half(); lda=K; geo=(0,0,0,0,0,1,0); mode=0
    W=(torch.randn(N,K,generator=g,device=dev)/K**0.5).half()
    bias=torch.randn(N,generator=g,device=dev); R=torch.randn(M,N,generator=g,device=dev).half()
    for rep in range(3):
        L.dh_dbg_gemm_stage(1); c0=gemm(A,lda,W,M,N,K,mode,geo,bias,R)
        L.dh_dbg_gemm_stage(33); c1=gemm(A,lda,W,M,N,K,mode,geo,bias,R)
        L.dh_dbg_gemm_stage(1)
        same=torch.equal(c0,c1)
        if not same: bad+=1
        print(M,N,K,conv,"identical" if same else f"DIFF max {float((c0.float()-c1.float()).abs().max())} nan {int(torch.isnan(c1.float()).sum())}")
print("bad",bad)
