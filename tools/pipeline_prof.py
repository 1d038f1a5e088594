"""Host / device cost of one BatchIterator.next_batch() of the bench input pipeline, component by component
(round 2: the per-sample augmentation draws cost 10 ms per batch on a 128-thread host until they moved to the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from hypelcnn_amd.backend import HipBackend
from hypelcnn_amd.common import common_nn_ops as cno
be = HipBackend()
it = bench.input_pipeline_iterator(be, 1024, 7, 145, 15, 0)
def T(f, n=20):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("next_batch ms", T(it.next_batch))
gen = torch.Generator(); gen.manual_seed(1)
info = it.augmentation_info
print("draw ms", T(lambda: cno.draw_augmentations(1024,145,info,gen)))
d = cno.draw_augmentations(1024,145,info,gen)
print("to(device) ms", T(lambda: {k:v.to(be.device) for k,v in d.items()}))
idx = torch.arange(1024, device=be.device)
print("apply ms", T(lambda: cno.apply_augmentations(be, it.arrays.data, idx, info, gen)))
print("threads", torch.get_num_threads())
torch.set_num_threads(4)
print("draw ms (4 threads)", T(lambda: cno.draw_augmentations(1024,145,info,gen)))
print("next_batch ms (4 threads)", T(it.next_batch))
