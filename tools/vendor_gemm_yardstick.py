#!/usr/bin/env python3
"""Yardstick only (never on the product path): what the vendor fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on
the plain-GEMM shapes of the H13 step, to put hypel_seg_gemm_f32's fraction of the fp32 MFMA peak in context."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(50176, 145, 120), (50176, 120, 240), (50176, 240, 480), (50176, 480, 480), (50176, 480, 240),
          (50176, 240, 120), (1024, 2940, 980), (1024, 405, 7105), (4 * 50176, 480, 480)]
for m, k, n in shapes:
    a = torch.rand(m, k, device="cuda"); b = torch.rand(k, n, device="cuda"); c = torch.empty(m, n, device="cuda")
    for _ in range(5):
        torch.mm(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        torch.mm(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"M={m:6d} K={k:4d} N={n:4d}  {us:8.1f} us  {2 * m * k * n / us / 1e6:6.1f} TF/s")
