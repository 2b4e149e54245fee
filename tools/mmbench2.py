"""GEMM shapes of the analytic SchNet passes at stacked-replica sizes: rocBLAS vs hipBLASLt kernel time."""
import sys
import time
import torch
dev = "cuda:0"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 458752
shapes = [("s@W2^T  [E,128]x[128,128]", (E, 128), (128, 128), False),
          ("Wb@W2   [E,128]x[128,128]", (E, 128), (128, 128), True),
          ("g@W1^T  [E,30]x[30,128]", (E, 30), (128, 30), False),
          ("ab@W1   [E,128]x[128,30]", (E, 128), (128, 30), True)]
for lib in ["hipblas", "hipblaslt"]:
    torch.backends.cuda.preferred_blas_library(lib)
    for name, sa, sw, plain in shapes:
        A = torch.randn(*sa, device=dev)
        W = torch.randn(*sw, device=dev)
        B = W if plain else W.t()
        for _ in range(3):
            A.mm(B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            A.mm(B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        fl = 2.0 * sa[0] * sa[1] * B.shape[1]
        print("%-9s %-28s %8.1f us  %6.1f TF  %6.2f TB/s" % (lib, name, dt * 1e6, fl / dt / 1e12,
              4.0 * (sa[0] * sa[1] + sa[0] * B.shape[1]) / dt / 1e12))
