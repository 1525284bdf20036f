#!/usr/bin/env python3
"""HBM rate of the stand-alone potential kernel (ff_potential_stream_kernel, 104 B per walker at 6 particles) by batch size: how much
of the gap to the 6.3 TB/s copy rate is launch ramp / tail of a 25 us launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fermiflow_amd import native
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (1 << 20, 1 << 22, 1 << 24):
    x = torch.randn(B, 6, 2, dtype=torch.float64, device=dev)
    native.potential(x, 2.0, True); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        native.potential(x, 2.0, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    y = torch.empty_like(x)
    y.copy_(x); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    msc = e0.elapsed_time(e1) / 10
    print(f"B = {B}: potential {ms * 1e3:.1f} us = {B * 104 / ms / 1e6:.0f} GB/s of algorithmic bytes; device copy of the same walkers "
          f"{msc * 1e3:.1f} us = {2 * B * 96 / msc / 1e6:.0f} GB/s (read + write)")
    del x, y
