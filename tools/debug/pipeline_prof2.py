import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hypelcnn_amd.backend import HipBackend
be = HipBackend()
it = bench.input_pipeline_iterator(be, 1024, 7, 145, 15, 0)
for _ in range(3): it.next_batch()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(24): it.next_batch()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
