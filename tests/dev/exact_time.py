"""Dev aid (GPU box): host-call time of the reference-order kernel (KBEST_FLAG_REFERENCE_ORDER, kbest_exact.hip) on a few batch shapes --
64x64 (the register form), integer 28x10 frames, 200x150 (the general form).  The numbers of NOTES 11.9."""
import time, numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
for (B,N,M,k) in [(256,64,64,200),(1024,64,64,200),(1000,28,10,200),(64,200,150,50)]:
    C = np.random.default_rng(1).random((B,N*M))
    if (N,M)==(28,10): C = np.random.default_rng(1).integers(0,6,(B,N*M)).astype(float)
    eng.kbest(C,N,M,k,reference_order=True)
    t=time.perf_counter(); out=eng.kbest(C,N,M,k,reference_order=True); dt=time.perf_counter()-t
    print(B,N,M,k,"%.1f ms"%(dt*1e3), "nf", int(out[0].sum()))
for (B,N,M,k) in [(1,200,150,50),(8,500,40,20),(2,1000,12,10)]:
    C = np.random.default_rng(2).random((B,N*M))
    eng.kbest(C,N,M,k,reference_order=True)
    t=time.perf_counter(); out=eng.kbest(C,N,M,k,reference_order=True); dt=time.perf_counter()-t
    print(B,N,M,k,"%.1f ms"%(dt*1e3), "nf", int(out[0].sum()))
