"""Does a HIP-graph capture on the compute stream collide with the NCCL watchdog's event queries?
Loop: eager collective, synchronize, [optional sleep], capture a small graph.  Run under torch.distributed.run (1 rank)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from hypelcnn_amd.backend import HipBackend, Ref
be = HipBackend()
sleep = float(os.environ.get("SLEEP", "0"))
n = int(os.environ.get("N", "300"))
buf = be.zeros(1 << 20)
x = torch.ones(1024, device=be.device)
gathered = be.zeros(4096)
src = be.zeros(4096)
for i in range(n):
    mode = os.environ.get("MODE", "allreduce")
    for k in range(40):
        if mode == "allreduce":
            dist.all_reduce(x[: 64 + k])
        else:
            if mode == "allgather":
                out = gathered[: 961].view(1, 961)
                dist.all_gather(list(out.unbind(0)), src[:961])
            else:
                dist.all_gather_into_tensor(gathered[:961], src[:961])
        be.call("fill_f32", Ref(buf), 4096, float(k))
    torch.cuda.synchronize()
    if sleep:
        time.sleep(sleep)
    raw = os.environ.get("RAW") == "1"   # RAW=1: no settle step (what fails)
    be.settle_before_capture() if not raw else None
    gs = [be.capture([be.bind("fill_f32", (Ref(buf), 1 << 16, float(i))) for _ in range(6)], True) for _ in range(40)]
    for g in gs:
        g()
torch.cuda.synchronize()
print("CAPTURE_LOOP_OK", n, "sleep", sleep)
dist.destroy_process_group()
