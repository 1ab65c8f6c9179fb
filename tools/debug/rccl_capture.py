import os, torch, torch.distributed as dist
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1024, device="cuda", dtype=torch.float64)
dist.all_reduce(x); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        dist.all_reduce(y)
        z = y + 1
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("RCCL all_reduce captured and replayed:", float(z[0]))
except Exception as e:
    print("capture of dist.all_reduce FAILED:", repr(e)[:400])
dist.destroy_process_group()
