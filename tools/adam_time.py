#!/usr/bin/env python
"""Time of the all-groups Adam launch at 1 M Gaussians (59 M floats), HIP events around 50 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import synth
from event_3dgs_amd.train_step import EventTrainer
dev = "cuda:0"
tr = EventTrainer(synth.make_scene(1_000_000, "trained", seed=0, device=dev), dev)
tr.flat_grad.normal_()
for _ in range(5):
    tr._adam(1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    tr._adam(2)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 50
print("adam %.4f ms  -> %.0f GB/s (28 B/param)" % (ms, tr.flat.numel() * 28 / ms / 1e6))
