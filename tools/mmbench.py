import torch, time, os
dev="cuda:0"
print("preferred:", torch.backends.cuda.preferred_blas_library())
for lib in ["default","hipblaslt","hipblas"]:
    if lib!="default":
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print(lib, "->", e); continue
    for (M,K,N) in [(512,64,128),(7168,30,128),(7168,128,30),(229376,128,30)]:
        A=torch.randn(M,K,device=dev); B=torch.randn(K,N,device=dev); bias=torch.randn(N,device=dev)
        for _ in range(5): A.mm(B)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(300): A.mm(B)
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        for _ in range(300): torch.addmm(bias,A,B)
        t3=time.perf_counter(); torch.cuda.synchronize()
        print("%-9s M=%6d K=%3d N=%3d  mm issue %.1f us (sync'd %.1f us)  addmm issue %.1f us" % (lib,M,K,N,(t1-t0)/300*1e6,(t2-t0)/300*1e6,(t3-t2)/300*1e6))
x=torch.randn(7168,128,device=dev)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(1000): y=x*2.0
t1=time.perf_counter(); torch.cuda.synchronize()
print("elementwise mul issue %.1f us" % ((t1-t0)/1000*1e6))
