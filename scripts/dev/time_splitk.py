import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wdg_amd import ops
from wdg_amd._lib import lib
for (m,k,n) in ((2708,1433,7),(5201,2089,64),(2277,2325,64)):
    a=torch.randn(m,k,device="cuda"); b=torch.randn(k,n,device="cuda")
    print("plan", lib.wdg_gemm_splitk_plan(m,n,k))
    for rep in range(3):
        ops.gemm(a,b); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.gemm(a,b)
        e1.record(); torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(30): ops.gemm(a,b)
        torch.cuda.synchronize()
        print((m,k,n), "events us/call", e0.elapsed_time(e1)/30*1e3, "wall us/call", (time.perf_counter()-t0)/30*1e6)
