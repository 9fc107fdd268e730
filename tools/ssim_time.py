#!/usr/bin/env python
"""HIP-event timing of the fused image loss (L1 + SSIM forward and gradient: e3dgs_image_loss) at a frame size.
Usage (GPU box): python tools/ssim_time.py [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from event_3dgs_amd import losses
W, H = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (1920, 1080)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
img, gt = torch.rand(3, H, W, device=dev, generator=g), torch.rand(3, H, W, device=dev, generator=g)
for gray in (True, False):
    for _ in range(3):
        out = losses.image_loss_raw(img, gt, gray, 0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = losses.image_loss_raw(img, gt, gray, 0.2)
    e1.record(); torch.cuda.synchronize()
    print("SSIM_TIME %s %dx%d: %.1f us per call (fwd + bwd + finalize), loss %.7f" % (
        "gray" if gray else "rgb", W, H, 1e3 * e0.elapsed_time(e1) / 20, float(out[0][0])))
