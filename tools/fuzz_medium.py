"""Randomised differential test at sizes where every list-building kernel runs MANY workgroups (chained scans over
hundreds of blocks, multi-block radix passes with dropped keys, 16- and 32-bit tile keys, segment-aligned last passes):
20 k .. 400 k Gaussians, frames from 300 x 200 to 2600 x 1500, trained-like and init-like scenes, cameras inside and
outside the cloud -- the single-view operator against the C oracle on a window of two tile rows (tests/helpers.py:
window_parity: radii, image, final_T, n_contrib bit-exact, instance counts, gradients per Gaussian).

Usage (GPU box, repo root):  python tools/fuzz_medium.py [cases] [first_seed]
TEST INFRASTRUCTURE: imports oracle/ (allowed for tests and tools run as tests, never for the product).
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_window_parity, window_parity   # noqa: E402
from event_3dgs_amd import synth                           # noqa: E402
from event_3dgs_amd.cameras import orbit_camera            # noqa: E402
from event_3dgs_amd.train_step import EventTrainer         # noqa: E402
from simple_knn._C import distCUDA2                        # noqa: E402

DEV = torch.device("cuda", 0)


def check(seed):
    r = np.random.default_rng(seed)
    N = int(r.choice([20_000, 60_000, 150_000, 400_000]))
    W, H = int(r.integers(300, 2600)), int(r.integers(200, 1500))
    kind = str(r.choice(["trained", "trained", "init"]))
    boost = float(np.exp(r.uniform(math.log(0.3), math.log(3.0))))
    radius = float(r.choice([1.0, 2.5, 4.0, 8.0]))
    bg = torch.tensor([float(r.choice([0.0, 0.3, 1.0]))] * 3, device=DEV)
    params = synth.make_scene(N, kind, seed=seed, device=DEV, dist2_fn=distCUDA2)
    params["scaling"] = params["scaling"] + math.log(boost)
    tr = EventTrainer(params, DEV, active_sh_degree=int(r.integers(0, 4)))
    cam = orbit_camera(int(r.integers(0, 16)), 16, W, H, device=DEV, radius=radius)
    rows = (H + 15) // 16
    r0 = int(r.integers(0, max(1, rows - 1)))
    what = "medium seed %d: N=%d %dx%d %s boost %.2f r=%.1f rows %d-%d" % (seed, N, W, H, kind, boost, radius, r0, r0 + 2)
    res = window_parity(tr, cam, bg, (r0, min(rows, r0 + 2)))
    what += " | visible %d, instances %d (oracle %d)" % (res["visible"], res["hip_instances"], res["oracle_instances"])
    try:
        # per-Gaussian bar 2e-2 instead of the tests' 1e-3: a splat hundreds of pixels wide (radius > 200: one or two per
        # scene at these sizes) sums ~1e5 pixel contributions in fp32 here and in double in the oracle, and the conic ->
        # scale / rotation chain amplifies that to ~1 % of its gradient (seeds 2012, 2062: ONE Gaussian of 400 000 above
        # 1e-3, its mean / opacity gradients at 1e-4); everything else about the window stays exact
        assert_window_parity(res, grad_pg=2e-2)
    except AssertionError as e:
        return what, [repr(e)[:300]]
    return what, []


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for s in range(first, first + cases):
        what, problems = check(s)
        if problems:
            bad += 1
            print("FAIL", what, "|", "; ".join(problems), flush=True)
        elif os.environ.get("FUZZ_VERBOSE"):
            print("ok  ", what, flush=True)
    print("fuzz_medium: %d cases, %d failing" % (cases, bad))
    sys.exit(1 if bad else 0)
