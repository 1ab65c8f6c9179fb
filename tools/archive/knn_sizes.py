import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import deltaconv_amd as dc
from deltaconv_amd.geometry import Graph
from deltaconv_amd.data import synthetic_batch
for B, N, k in [(32, 1024, 20), (32, 2048, 20), (32, 2048, 10), (8, 4096, 30), (16, 2048, 20)]:
    b = synthetic_batch(B, N, seed=3).to("cuda")
    info = dc.models.deltanet_base._ptr_info(b)
    for lanes in (8, 64):
        f = lambda: Graph.knn(b.pos, k, ptr_info=info, lanes_per_query=lanes)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); print(f"B={B} N={N} k={k} lanes={lanes}: {(time.perf_counter()-t0)/20*1e6:8.1f} us", flush=True)
