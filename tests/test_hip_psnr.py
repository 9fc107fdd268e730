"""GPU: "converge to the same PSNR" (north_star).

The same scene is trained twice from the same initial model with the same camera sequence and ground-truth frames:
  * on the HIP path (EventTrainer.step: fused three-view rasteriser, event-loss kernel, Adam kernel), and
  * on the CPU oracle (oracle/train_oracle.py: C-oracle rasteriser behind autograd, the reference's loss formulas and
    torch.optim.Adam in PyTorch) -- the stand-in for the reference, whose rasteriser source is absent.
Both models are then scored with the reference's evaluation protocol (eval.py:118-152: gray PSNR, utils/image_utils.py:
19-21, on the held-out views 5/25/45/65/85, which the sampler of train.py:116-131 never draws).  Criterion: the two
PSNRs agree within 0.1 dB.
"""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("fast_exp", [False, True])
def test_hip_trained_and_oracle_trained_reach_the_same_psnr(fast_exp):
    """fast_exp=True: the same criterion for the tolerance mode (E3DGS_FLAG_FAST_EXP: hardware exp in the compositing
    kernels) -- what lets bench.py report it as a second, labelled figure."""
    from event_3dgs_amd import fit, scene_io, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    from oracle import torch_oracle
    from oracle.train_oracle import OracleTrainer, camera_dict
    torch.set_num_threads(min(8, torch.get_num_threads()))      # tiny CPU tensors: a 256-thread team only adds latency
    N, W, H, K, ITERS = 2000, 128, 96, 100, 220
    bg = torch.ones(3, device=DEV)                                # white: ln(Y + 1e-8) of black pixels swamps the contrast
    bg_np = np.ones(3, np.float32)
    gt_params = synth.make_scene(N, "trained", seed=5, device=DEV)
    gt_tr = EventTrainer(gt_params, DEV)
    q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
    train, events = [], []
    for k in range(K):
        for lst, daz in ((train, 0.0), (events, 0.002)):
            c = orbit_camera(k, K, W, H, device=DEV, daz=daz)
            c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
            lst.append(c)
    # initial model: the ground truth with displaced centres, dimmer colours and flatter opacities
    g = torch.Generator().manual_seed(11)
    init = {k: v.clone() for k, v in gt_params.items()}
    init["xyz"] += 0.02 * torch.randn(N, 3, generator=g).to(DEV)
    init["features_dc"] += 0.5 * torch.randn(N, 1, 3, generator=g).to(DEV)
    init["opacity"] *= 0.7
    hip = EventTrainer(init, DEV, fast_exp=fast_exp)
    ora = OracleTrainer(init)
    cd_train = [camera_dict(c) for c in train]
    cd_event = [camera_dict(c) for c in events]
    img_train = [c.original_image.cpu() for c in train]
    img_event = [c.original_image.cpu() for c in events]
    with_gt = list(zip(cd_train, img_train))

    def score():
        p_hip = scene_io.evaluate_views(lambda cam: hip.render_raw(cam, bg)["color"], train)["psnr"]
        p_ora = torch_oracle.eval_gray_psnr(lambda cam: ora.render(cam, bg_np), with_gt)
        return p_hip, p_ora
    p0_hip, p0_ora = score()
    assert abs(p0_hip - p0_ora) <= 1e-3                          # same model, same images: the two PSNR codes agree
    rnd = random.Random(0)
    lh, lo = [], []
    for it in range(ITERS):
        i = fit.sample_index(K, "event", rnd.randint)
        assert i not in fit.HELD_OUT
        lh.append(float(hip.step(train[i], events[i], events[i + 1], train[i].original_image, events[i].original_image,
                                 events[i + 1].original_image, bg)[0]))
        lo.append(ora.step(cd_train[i], cd_event[i], cd_event[i + 1], img_train[i], img_event[i], img_event[i + 1], bg_np))
    p_hip, p_ora = score()
    print(f"gray PSNR on the held-out views: initial {p0_hip:.3f} dB; after {ITERS} iterations HIP-trained {p_hip:.3f} dB, "
          f"oracle-trained {p_ora:.3f} dB; first losses {lh[0]:.6f} / {lo[0]:.6f}, last {lh[-1]:.6f} / {lo[-1]:.6f}; "
          f"c {float(hip.c):.5f} / {float(ora.c):.5f}")
    assert abs(lh[0] - lo[0]) <= (1e-4 if fast_exp else 1e-5) * abs(lo[0])
    assert p_hip > p0_hip + 1.0 and p_ora > p0_ora + 1.0         # both actually trained
    assert abs(p_hip - p_ora) <= 0.1                             # north_star: PSNR within 0.1 dB of the reference


def test_cfg2_size_scene_hip_and_oracle_reach_the_same_psnr():
    """The same criterion at the size of BASELINE.json configs[1]: 200 k Gaussians at the pixel footprint of an 800 x 800
    frame, 300 (E3DGS_PSNR_ITERS=600: 600) event iterations with a random camera each (train.py:116-131).  The CPU oracle cannot composite 800 x 800
    frames 1 800 times inside a test, so both trainers see the CENTRAL 160 x 128 WINDOW of every 800 x 800 view -- a
    camera with the same centre and pixel pitch (tan(FoV / 2) scaled by 160 / 800 and 128 / 800) -- which keeps the
    Gaussian count, the per-tile list lengths and the splat sizes of the full frame.  Held-out views 5/25/45/65/85, gray
    PSNR (eval.py:118-152), HIP-trained against oracle-trained: within 0.1 dB."""
    import math
    from event_3dgs_amd import fit, scene_io, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    from oracle import torch_oracle
    from oracle.train_oracle import OracleTrainer, camera_dict
    torch.set_num_threads(min(16, torch.get_num_threads()))
    import os
    # (600 iterations is the figure the round-5 review asked for and DESIGN.md section 5 quotes; the suite runs 300 by default --
    # the oracle's iteration takes ~0.5 s on the host -- and E3DGS_PSNR_ITERS=600 runs the full length)
    N, FULL, W, H, K, ITERS = 200_000, 800, 160, 128, 100, int(os.environ.get("E3DGS_PSNR_ITERS", "300"))
    fovx_full = 0.6911112070083618
    fovx = 2.0 * math.atan(math.tan(fovx_full / 2.0) * W / FULL)
    bg = torch.ones(3, device=DEV)
    bg_np = np.ones(3, np.float32)
    gt_params = synth.make_scene(N, "trained", seed=6, device=DEV)
    gt_tr = EventTrainer(gt_params, DEV)
    q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
    train, events = [], []
    for k in range(K):
        for lst, daz in ((train, 0.0), (events, 0.002)):
            c = orbit_camera(k, K, W, H, device=DEV, daz=daz, fovx=fovx)
            c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
            lst.append(c)
    del gt_tr
    g = torch.Generator().manual_seed(12)
    init = {k: v.clone() for k, v in gt_params.items()}
    init["xyz"] += 0.004 * torch.randn(N, 3, generator=g).to(DEV)          # (a fifth of the mean neighbour spacing)
    init["features_dc"] += 0.5 * torch.randn(N, 1, 3, generator=g).to(DEV)
    init["opacity"] *= 0.7
    hip = EventTrainer(init, DEV)
    ora = OracleTrainer(init, parallel_views=True)
    cd_train = [camera_dict(c) for c in train]
    cd_event = [camera_dict(c) for c in events]
    img_train = [c.original_image.cpu() for c in train]
    img_event = [c.original_image.cpu() for c in events]
    with_gt = list(zip(cd_train, img_train))

    def score():
        p_hip = scene_io.evaluate_views(lambda cam: hip.render_raw(cam, bg)["color"], train)["psnr"]
        p_ora = torch_oracle.eval_gray_psnr(lambda cam: ora.render(cam, bg_np), with_gt)
        return p_hip, p_ora
    p0_hip, p0_ora = score()
    assert abs(p0_hip - p0_ora) <= 1e-3
    rnd = random.Random(1)
    lh, lo = [], []
    for it in range(ITERS):
        i = fit.sample_index(K, "event", rnd.randint)
        lh.append(float(hip.step(train[i], events[i], events[i + 1], train[i].original_image, events[i].original_image,
                                 events[i + 1].original_image, bg)[0]))
        lo.append(ora.step(cd_train[i], cd_event[i], cd_event[i + 1], img_train[i], img_event[i], img_event[i + 1], bg_np))
    p_hip, p_ora = score()
    print(f"cfg2-size scene, {W}x{H} window of {FULL}x{FULL}: gray PSNR on the held-out views initial {p0_hip:.3f} dB; after "
          f"{ITERS} iterations HIP-trained {p_hip:.3f} dB, oracle-trained {p_ora:.3f} dB; first losses {lh[0]:.6f} / "
          f"{lo[0]:.6f}, last {lh[-1]:.6f} / {lo[-1]:.6f}; c {float(hip.c):.5f} / {float(ora.c):.5f}")
    assert abs(lh[0] - lo[0]) <= 1e-5 * abs(lo[0])
    assert p_hip > p0_hip + 0.5 and p_ora > p0_ora + 0.5
    assert abs(p_hip - p_ora) <= 0.1
