"""GPU: the multi-rank training step (view-parallel DP) end to end.  The box has one GPU, so the two
ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); the code path is the one
`bench.py --gpus N` runs -- chunked asynchronous mean of the flat gradient buffer pipelined with the per-chunk
Adam launches -- and must equal the single-process emulation (mean of the two ranks' gradients, plain Adam)
bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, W, H, STEPS = 6000, 208, 144, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(rank, n=N):
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params = synth.make_scene(n, "trained", seed=0, device=DEV)
    cams = [orbit_camera(3 * rank, 32 if rank >= 5 else 16, W, H, device=DEV, daz=d) for d in (0.0, 0.004, 0.012)]
    bg = torch.zeros(3, device=DEV)
    gp = dict(params)
    gp["xyz"] = params["xyz"] + 0.01 * torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    gt = EventTrainer(gp, DEV)
    gts = [(torch.round(gt.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous() for c in cams]
    return params, cams, gts, bg


def _worker(rank, world, port, out, overlap, factorize, schedule="allreduce", n=N, steps=STEPS):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["E3DGS_FACTORIZE_SH"] = "1" if factorize else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _inputs(rank, n)
    tr = EventTrainer(params, DEV, overlap_features=overlap, dp_schedule=schedule)
    assert tr.world == world and tr.overlap_features == overlap and tr.factorize_sh == factorize
    assert tr.dp_schedule == schedule and tr.rank == rank
    for _ in range(steps):
        tr.step(*cams, *gts, bg)
    if schedule == "rs_ag":
        # every rank only advanced the moments of its own shard of the non-SH groups ...
        off, n = tr.seg["xyz"]
        stale = tr.exp_avg[off:off + n].clone()
        tr.sync_optimizer_state()                       # ... until they are gathered (export / densify / checkpoint do)
        assert not torch.equal(stale, tr.exp_avg[off:off + n])
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "m": tr.exp_avg.cpu(), "v": tr.exp_avg_sq.cpu(), "seg": dict(tr.seg)},
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,factorize", [(True, True), (False, True), (True, False), (False, False)])
def test_two_ranks_equal_single_process_emulation(tmp_path, overlap, factorize):
    """overlap: the SH-coefficient exchange + Adam run on a side stream under the next iteration's projection / sorts /
    binning (deferred colour kernel).  factorize: the ranks all-gather the per-view colour gradients and rebuild the
    mean SH gradient instead of averaging the SH gradient itself -- the same mathematical mean, summed in another
    order, so the SH coefficients agree with the emulation to fp32 rounding and everything else bit for bit."""
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, _free_port(), out, overlap, factorize), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k                    # replicas stay identical
    # single process: both ranks' gradients, their mean, plain (unchunked) Adam
    from event_3dgs_amd.train_step import EventTrainer
    assert not dist.is_initialized()
    pa, ca, ga, bg = _inputs(0)
    pb, cb, gb, _ = _inputs(1)
    ta, tb = EventTrainer(pa, DEV, overlap_features=False), EventTrainer(pb, DEV, overlap_features=False)
    for _ in range(STEPS):
        ta.compute_gradients(*ca, *ga, bg)
        tb.compute_gradients(*cb, *gb, bg)
        mean = (ta.flat_grad + tb.flat_grad).div_(2)           # what gloo's SUM followed by div_(world) computes
        ta.flat_grad.copy_(mean); tb.flat_grad.copy_(mean)
        ta.apply_update(); tb.apply_update()
    torch.cuda.synchronize()
    if not factorize:
        assert torch.equal(ta.flat.cpu(), r0["flat"])
        assert torch.equal(ta.exp_avg.cpu(), r0["m"])
        assert torch.equal(ta.exp_avg_sq.cpu(), r0["v"])
    else:
        f0, fn = r0["seg"]["features"]
        feat = slice(f0, f0 + fn)
        m_ref, m_got = ta.exp_avg.cpu(), r0["m"]
        assert float((m_got[feat] - m_ref[feat]).norm() / m_ref[feat].norm()) < 1e-5       # rounding only
        # after the first step the non-SH groups see slightly different SH coefficients -> tolerance there too
        assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
        assert float((r0["flat"] - ta.flat.cpu()).abs().max()) <= 0.05
    # and the update really used both ranks' views
    solo = EventTrainer(pa, DEV)
    for _ in range(STEPS):
        solo.step(*ca, *ga, bg)
    assert not torch.equal(solo.flat.cpu(), r0["flat"])


@pytest.mark.parametrize("overlap,factorize", [(False, False), (True, True)])
def test_reduce_scatter_allgather_schedule_equals_the_allreduce_schedule(tmp_path, overlap, factorize):
    """dp_schedule="rs_ag" (reduce-scatter of the non-SH gradients, Adam on the owned shard, all-gather of the
    updated parameters): replicas bit-identical, and -- two ranks over gloo sum two numbers either way -- bit-identical to
    the all-reduce schedule, parameters and (after sync_optimizer_state) moments."""
    res = {}
    for sched in ("allreduce", "rs_ag"):
        out = str(tmp_path / sched)
        mp.spawn(_worker, args=(2, _free_port(), out, overlap, factorize, sched), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
        for k in ("flat", "m", "v"):
            assert torch.equal(r0[k], r1[k]), (sched, k)
        res[sched] = r0
    for k in ("flat", "m", "v"):
        assert torch.equal(res["allreduce"][k], res["rs_ag"][k]), k


# ---------------------------------------------------------------------------------------------------------------
# World 4 and 8 on the one GPU of the box: what `bench.py --gpus 8` and an 8-GPU fit will execute -- the rs_ag shards of
# ceil((11 N + 1) / 8) elements with a padded last shard (N not divisible by 8), the 8-block factorised SH all-gather, the
# chunked all-reduce at world 8 -- so that the first real 8-GPU run exercises nothing new but the links.
N_BIG = 20003


@pytest.mark.parametrize("schedule,overlap,factorize", [("allreduce", True, True), ("rs_ag", True, True),
                                                         ("allreduce", False, False), ("rs_ag", False, False)])
@pytest.mark.parametrize("world", [4, 8])
def test_world_4_and_8_replicas_identical_and_equal_to_the_mean_of_the_ranks(tmp_path, world, schedule, overlap, factorize):
    """Every rank renders its own triplet of the 20 003-Gaussian scene; after two steps the replicas are bit-identical
    (parameters and, after sync_optimizer_state, both moments) and track the single-process emulation: the mean of the
    `world` gradient buffers, one Adam.  (gloo adds more than two numbers in its own order: fp32 rounding against the
    emulation's sum, not bit equality.)"""
    steps = 2
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), out, overlap, factorize, schedule, N_BIG, steps), nprocs=world, join=True)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in range(1, world):
        for k in ("flat", "m", "v"):
            assert torch.equal(rs[0][k], rs[r][k]), (r, k)
    from event_3dgs_amd.train_step import EventTrainer
    assert not dist.is_initialized()
    ins = [_inputs(r, N_BIG) for r in range(world)]
    bg = ins[0][3]
    ts = [EventTrainer(i[0], DEV, overlap_features=False) for i in ins]
    for _ in range(steps):
        for t, (_, cams, gts, _) in zip(ts, ins):
            t.compute_gradients(*cams, *gts, bg)
        mean = torch.stack([t.flat_grad for t in ts]).sum(0).div_(world)
        for t in ts:
            t.flat_grad.copy_(mean)
            t.apply_update()
    torch.cuda.synchronize()
    m_ref, m_got = ts[0].exp_avg.cpu(), rs[0]["m"]
    assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
    v_ref, v_got = ts[0].exp_avg_sq.cpu(), rs[0]["v"]
    assert float((v_got - v_ref).norm() / v_ref.norm()) < 1e-4
    assert float((rs[0]["flat"] - ts[0].flat.cpu()).abs().max()) <= 0.05
    # every rank's views went into the update: rank 0 alone ends elsewhere
    solo = EventTrainer(ins[0][0], DEV)
    for _ in range(steps):
        solo.step(*ins[0][1], *ins[0][2], bg)
    assert not torch.equal(solo.flat.cpu(), rs[0]["flat"])


def _fit_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd import fit, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    Wf, Hf, K = 112, 80, 32
    bg = torch.zeros(3, device=DEV)
    gt_params = synth.make_scene(2500, "trained", seed=9, device=DEV)
    gt_tr = EventTrainer(gt_params, DEV, overlap_features=False)
    q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
    train, events = [], []
    for k in range(K):
        for lst, daz in ((train, 0.0), (events, 0.003)):
            c = orbit_camera(k, K, Wf, Hf, device=DEV, daz=daz)
            c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
            lst.append(c)
    params = synth.make_scene(1200, "trained", seed=3, device=DEV)
    sizes = []
    tr = fit.fit_event_scene(params, train, events, bg, DEV, iterations=14, cameras_extent=4.4, densify_from_iter=3,
                             densification_interval=5, densify_grad_threshold=1e-9, start_sh_degree=3, seed=5,
                             on_iteration=lambda it, t, s: sizes.append(t.N))
    torch.cuda.synchronize()
    g = tr.export_groups()
    torch.save({"sizes": sizes, "xyz": g["xyz"][0].cpu(), "f_rest": g["f_rest"][0].cpu(), "m": g["scaling"][1].cpu()},
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_fit_with_densification_keeps_replicas_identical(tmp_path, world):
    """The event training loop on two / eight ranks: per-rank camera draws, averaged gradients, densification statistics
    combined over the ranks (sum / max, DensifyStats.sync) and an identically seeded split sampler -- after two
    densification steps the replicas hold the same number of Gaussians and bit-identical parameters and optimizer state."""
    out = str(tmp_path / "fit")
    mp.spawn(_fit_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rs = [torch.load(f"{out}.{r}") for r in range(world)]
    assert rs[0]["sizes"][0] == 1200 and rs[0]["sizes"][-1] != 1200          # densification really changed the model
    for r in range(1, world):
        assert rs[0]["sizes"] == rs[r]["sizes"]
        for k in ("xyz", "f_rest", "m"):
            assert torch.equal(rs[0][k], rs[r][k]), (r, k)


def _shared_pose_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _inputs(rank)
    if rank == 0:             # this rank's event camera `now` carries the pose of its training camera (a real dataset)
        cams = [cams[0], orbit_camera(0, 16, W, H, device=DEV, daz=0.0), cams[2]]
    # (rank 0 also collects densification statistics: the shared view's tiles run the second gradient chain -- which must
    # leave every exchanged gradient as it is)
    tr = EventTrainer(params, DEV, track_densification_stats=(rank == 0))
    tr.SHARE_STATS_MIN_INSTANCES = tr.SHARE_STATS_MIN_TILES = 0
    assert tr.factorize_sh
    for _ in range(STEPS):
        tr.step(*cams, *gts, bg)
    tr.sync_features()
    torch.cuda.synchronize()
    assert tr.shared_pose_iterations == (STEPS if rank == 0 else 0)
    assert rank != 0 or float(tr.viewspace_grad.abs().sum()) > 0
    torch.save({"flat": tr.flat.cpu(), "m": tr.exp_avg.cpu(), "v": tr.exp_avg_sq.cpu()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_of_them_with_a_shared_pose_triplet(tmp_path):
    """Rank 0 renders two views (its renders #1 and #2 share a pose), rank 1 three: the blocks of the colour-gradient
    all-gather keep one size (rank 0 pads a zero third view), the replicas stay bit-identical and track the emulation
    with three separate renders on both ranks."""
    out = str(tmp_path / "sp")
    mp.spawn(_shared_pose_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    pa, ca, ga, bg = _inputs(0)
    pb, cb, gb, _ = _inputs(1)
    ca = [ca[0], orbit_camera(0, 16, W, H, device=DEV, daz=0.0), ca[2]]
    ta, tb = EventTrainer(pa, DEV, overlap_features=False), EventTrainer(pb, DEV, overlap_features=False)
    ta.share_coincident_views = False
    for _ in range(STEPS):
        ta.compute_gradients(*ca, *ga, bg)
        tb.compute_gradients(*cb, *gb, bg)
        mean = (ta.flat_grad + tb.flat_grad).div_(2)
        ta.flat_grad.copy_(mean); tb.flat_grad.copy_(mean)
        ta.apply_update(); tb.apply_update()
    torch.cuda.synchronize()
    m_ref, m_got = ta.exp_avg.cpu(), r0["m"]
    assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
    assert float((r0["flat"] - ta.flat.cpu()).abs().max()) <= 0.05


def _mixed_size_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _inputs(rank)
    if rank == 1:             # this rank's intensity frame has another size than its event pair (utils/camera_utils.py:19-52)
        cams = [orbit_camera(3, 16, 160, 112, device=DEV), cams[1], cams[2]]
        gts = [EventTrainer(params, DEV).render_raw(cams[0], bg)["color"].clone(), gts[1], gts[2]]
    tr = EventTrainer(params, DEV)
    assert tr.factorize_sh and tr.overlap_features
    for _ in range(STEPS):
        tr.step(*cams, *gts, bg)
    tr.sync_features()
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "m": tr.exp_avg.cpu(), "v": tr.exp_avg_sq.cpu()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_of_them_with_a_mixed_size_triplet(tmp_path):
    """Rank 1's intensity frame is smaller than its event pair: it runs two multi-view passes and still hands the
    factorised exchange the three-view colour-gradient block every rank all-gathers -- no mismatched collectives, replicas
    bit-identical, and the update tracks the mean-of-gradients emulation."""
    out = str(tmp_path / "mx")
    mp.spawn(_mixed_size_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    pa, ca, ga, bg = _inputs(0)
    pb, cb, gb, _ = _inputs(1)
    cb = [orbit_camera(3, 16, 160, 112, device=DEV), cb[1], cb[2]]
    gb = [EventTrainer(pb, DEV).render_raw(cb[0], bg)["color"].clone(), gb[1], gb[2]]
    ta, tb = EventTrainer(pa, DEV, overlap_features=False), EventTrainer(pb, DEV, overlap_features=False)
    for _ in range(STEPS):
        ta.compute_gradients(*ca, *ga, bg)
        tb.compute_gradients(*cb, *gb, bg)
        mean = (ta.flat_grad + tb.flat_grad).div_(2)
        ta.flat_grad.copy_(mean); tb.flat_grad.copy_(mean)
        ta.apply_update(); tb.apply_update()
    torch.cuda.synchronize()
    m_ref, m_got = ta.exp_avg.cpu(), r0["m"]
    assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
    assert float((r0["flat"] - ta.flat.cpu()).abs().max()) <= 0.05


def _image_worker(rank, world, port, out, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd.train_step import EventTrainer
    params, cams, gts, bg = _inputs(rank)
    tr = EventTrainer(params, DEV)                       # defaults of a multi-rank run: overlap + factorised SH
    assert tr.overlap_features and tr.factorize_sh
    for _ in range(STEPS):
        tr.step_image(cams[0], gts[0], bg, mode=mode)
    tr.sync_features()
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "m": tr.exp_avg.cpu(), "v": tr.exp_avg_sq.cpu()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["gray", "rgb"])
def test_two_ranks_image_modes(tmp_path, mode):
    """The one-render iterations (train.py:213-223, 292-296) under view-parallel DP: one camera per rank, the same
    exchange as the event iteration with a single view in the colour-gradient all-gather.  Replicas stay bit-identical,
    track the mean-of-gradients emulation, and differ from a single-rank run."""
    out = str(tmp_path / "img")
    mp.spawn(_image_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    from event_3dgs_amd.train_step import EventTrainer
    pa, ca, ga, bg = _inputs(0)
    pb, cb, gb, _ = _inputs(1)
    ta, tb = EventTrainer(pa, DEV, overlap_features=False), EventTrainer(pb, DEV, overlap_features=False)
    for _ in range(STEPS):
        ta.compute_gradients_image(ca[0], ga[0], bg, mode=mode)
        tb.compute_gradients_image(cb[0], gb[0], bg, mode=mode)
        mean = (ta.flat_grad + tb.flat_grad).div_(2)
        ta.flat_grad.copy_(mean); tb.flat_grad.copy_(mean)
        ta.apply_update(); tb.apply_update()
    torch.cuda.synchronize()
    m_ref, m_got = ta.exp_avg.cpu(), r0["m"]
    assert float((m_got - m_ref).norm() / m_ref.norm()) < 1e-4
    solo = EventTrainer(pa, DEV)
    for _ in range(STEPS):
        solo.step_image(ca[0], ga[0], bg, mode=mode)
    assert not torch.equal(solo.flat.cpu(), r0["flat"])


# ---------------------------------------------------------------------------------------------------------------
# view-parallel DP changes the batch from 1 triplet to `world` triplets per optimizer step, so quality has to
# be compared at equal numbers of SAMPLES (camera triplets seen), not only at equal numbers of steps.
def _psnr_setup():
    from event_3dgs_amd import synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer
    Np, Wp, Hp, K = 2000, 128, 96, 100
    bg = torch.ones(3, device=DEV)                   # white: ln(Y + 1e-8) of black pixels swamps the contrast
    gt_params = synth.make_scene(Np, "trained", seed=5, device=DEV)
    gt_tr = EventTrainer(gt_params, DEV, overlap_features=False)
    q8 = lambda t: (torch.round(t.clamp(0, 1) * 255) / 255).contiguous()
    train, events = [], []
    for k in range(K):
        for lst, daz in ((train, 0.0), (events, 0.002)):
            c = orbit_camera(k, K, Wp, Hp, device=DEV, daz=daz)
            c.original_image = q8(gt_tr.render_raw(c, bg)["color"])
            lst.append(c)
    g = torch.Generator().manual_seed(11)
    init = {k: v.clone() for k, v in gt_params.items()}
    init["xyz"] += 0.02 * torch.randn(Np, 3, generator=g).to(DEV)
    init["features_dc"] += 0.5 * torch.randn(Np, 1, 3, generator=g).to(DEV)
    init["opacity"] *= 0.7
    return init, train, events, bg


def _psnr_after(iters):
    """Event-mode training (no densification) of the perturbed scene for `iters` optimizer steps -> (initial, final)
    gray PSNR on the held-out views (eval.py:118-152 protocol)."""
    from event_3dgs_amd import fit, scene_io
    from event_3dgs_amd.train_step import EventTrainer
    init, train, events, bg = _psnr_setup()
    p0 = scene_io.evaluate_views(lambda cam: EventTrainer(init, DEV, overlap_features=False).render_raw(cam, bg)["color"],
                                 train)["psnr"] if iters == 0 else None
    if iters == 0:
        return p0
    import random
    # (the single-rank loop draws its cameras with `rng`, by default the global random.randint as train.py:116 does:
    # seeded here, or the single-rank PSNRs move by +-0.2 dB from run to run and the envelope below turns flaky)
    tr = fit.fit_event_scene(init, train, events, bg, DEV, iterations=iters, cameras_extent=4.4,
                             densify_from_iter=10 ** 9, start_sh_degree=3, seed=5, white_background=True,
                             rng=random.Random(5).randint)
    return scene_io.evaluate_views(lambda cam: tr.render_raw(cam, bg)["color"], train)["psnr"]


def _psnr_worker(rank, world, port, iters, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _psnr_after(iters)
    if rank == 0:
        torch.save(float(p), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_quality_at_equal_steps_and_equal_samples(tmp_path):
    """Two ranks x K steps (2K camera triplets) against one rank x K steps and one rank x 2K steps (the same 2K
    triplets' worth of samples).  Measured on this scene (K = 110): initial 24.2 dB; one rank 30.4 dB (K) / 32.5 dB (2K);
    two ranks 31.6 dB.  Averaging two views per step lowers the gradient noise (better than one rank at equal steps)
    but Adam's step length does not grow with the batch, so at equal samples the two-rank run has taken half the steps
    and trails by ~1 dB in this short, learning-rate-limited regime (0.5 dB at twice the length).  The assertions pin
    that envelope: data parallelism must not cost quality per step, and must stay within 1.5 dB per sample."""
    K = 110
    out = str(tmp_path / "psnr")
    mp.spawn(_psnr_worker, args=(2, _free_port(), K, out), nprocs=2, join=True)
    p_dp = torch.load(out)
    assert not dist.is_initialized()
    p0, p_k, p_2k = _psnr_after(0), _psnr_after(K), _psnr_after(2 * K)
    print(f"held-out gray PSNR: initial {p0:.2f} dB; 1 rank x {K}: {p_k:.2f}; 1 rank x {2 * K}: {p_2k:.2f}; "
          f"2 ranks x {K}: {p_dp:.2f}")
    assert p_k > p0 + 3.0 and p_2k > p_k                      # the single-rank runs train
    assert p_dp >= p_k - 0.1                                   # equal steps: DP at least as good
    assert p_dp >= p_2k - 1.5                                  # equal samples: within the envelope described above
