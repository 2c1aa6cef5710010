import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
streams = [torch.cuda.Stream(dev) for _ in range(3)]
local = [torch.zeros((135, 1920, 3), device=dev) for _ in range(3)]
slab = [torch.empty((1, 135, 1920, 3), device=dev) for _ in range(3)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(30):
    s = k % 3
    streams[s].wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(streams[s]):
        local[s].fill_(float(k))
        dist.gather(local[s], list(slab[s].unbind(0)), dst=0)
        slab[s].mul_(2.0)
torch.cuda.synchronize()
print("30 gathers on 3 side streams: %.2f ms, values %s" % ((time.perf_counter() - t0) * 1e3, [float(x[0, 0, 0, 0]) for x in slab]))
assert [float(x[0, 0, 0, 0]) for x in slab] == [54.0, 56.0, 58.0]
dist.barrier(device_ids=[0]); dist.destroy_process_group(); print("ok")
