"""Experiment: can a (1-rank) RCCL all-reduce be captured into a hipGraph with this torch build?
usage: python tools/try_graph_allreduce.py [thread_local|global|relaxed]"""
import os, sys, torch, torch.distributed as dist
mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "36123"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode=mode):
        y = x * 2
        dist.all_reduce(y)
        z = y + 1
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(mode, "OK", float(z[0]))
except Exception as e:
    print(mode, "FAILED:", str(e).splitlines()[0][:300])
dist.destroy_process_group()
