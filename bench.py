#!/usr/bin/env python
"""bench.py -- headline benchmark: train iters/s of the reference's `--event` iteration
(train.py:97-332: three renders fwd+bwd, event + intensity loss, Adam) on synthetic data.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched through torch.distributed.run)

Prints ONE JSON line (rank 0).  Workload at N=1 is BASELINE.json configs[2]:
1M Gaussians, 1920x1080, event iteration, "trained-like" synthetic scene of SURVEY 8(d).
With N>1 every rank renders its own camera triplet of the same replicated model and the
59 floats/Gaussian of gradient are all-reduced over RCCL (view-parallel DP, weak scaling:
value = camera triplets (= reference iterations) processed per second by the whole job).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of these boxes only supports dmabuf IPC: RCCL between processes needs this (kept if already set)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

CONFIGS = {
    # name: (N gaussians, W, H, deblur)
    "cfg2_200k_800px": (200_000, 800, 800, False),
    "cfg3_1M_1080p_event": (1_000_000, 1920, 1080, False),
    "cfg4_1M_1080p_deblur": (1_000_000, 1920, 1080, True),
    "cfg5_2M_1080p_event": (2_000_000, 1920, 1080, False),
    "tiny": (20_000, 320, 240, False),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(slot, N, I, T, npx):
    """SURVEY 8(d) algorithmic bytes of each profiled stage; a launch covers all views of the iteration, so the
    caller passes N = Gaussians x views (splats), I = instances of all views, T / npx = tiles / pixels of all views."""
    tile_bits = max(1, math.ceil(math.log2(max(T, 2))))
    return {
        "preprocess": 132 * N,
        "sort_depth": 16 * N * 4,                       # 4 passes x (8 B read + 8 B write) on P pairs
        "scan_emit": 8 * N + 20 * N + 8 * I,            # gather+scan, emit (u32 tile id + u32 Gaussian id)
        # stable tile-id sort of (key, slot) pairs: 16-bit keys up to 65536 tiles; the first pass reads keys only (the
        # payload is the position), the last pass writes slots only (the ranges come out of its histogram)
        "sort_tile": (18 if T <= 65536 and 8 < tile_bits <= 16 else 16 * math.ceil(tile_bits / 8)) * I,
        "tile_ranges": 12 * T,                          # launch-order kernel: ranges in, order out
        "render_fwd": 40 * I + 20 * npx + 12,
        "render_bwd": 40 * I + 20 * npx + 36 * I,
        "geom_bwd": 160 * N,
    }[slot]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-substep", action="store_true",
                    help="skip the contrast-only two-render measurement after the timed region (profiling runs: keeps the "
                         "kernel statistics to the 3-view launches of the iteration)")
    ap.add_argument("--trained-steps", type=int, default=1000,
                    help="training iterations of the trained_random_camera leg before its timed window (0 = skip the leg)")
    ap.add_argument("--trained-timed", type=int, default=200, help="timed iterations of the trained_random_camera leg")
    ap.add_argument("--only-trained", action="store_true",
                    help="run ONLY the trained_random_camera leg and print its record (profiling runs: the kernel statistics "
                         "of the process then describe that leg; see profiles/collect.py --trained)")
    ap.add_argument("--cpu-rows", type=int, default=3, help="tile rows composited by the C-oracle sample")
    ap.add_argument("--torch-rows", type=int, default=1, help="tile rows composited by the PyTorch CPU baseline sample")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU); rank 0 of the children prints the line
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from event_3dgs_amd import _lib, synth
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer, FLOATS_PER_GAUSSIAN

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # test hooks (one-GPU boxes): E3DGS_BENCH_BACKEND=gloo + E3DGS_BENCH_DEVICE=0 run the N>1 code path with all
    # ranks sharing one device; the driver's runs use neither (one rank per GPU over RCCL)
    backend = os.environ.get("E3DGS_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("E3DGS_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- who is really in the communicator: every rank contributes (host, device UUID / PCI bus id) through the backend's own
    # all-gather; `rccl_ranks` in the JSON line is the number of DISTINCT devices that answered -- world ranks that share a GPU,
    # or a backend that is not RCCL, show up here instead of hiding behind get_world_size().  E3DGS_BENCH_STRICT=1: anything
    # but one RCCL rank per GPU is an error.
    rccl_ranks, distinct_devices = 1, 1
    if world > 1:
        import socket
        import zlib
        props = torch.cuda.get_device_properties(dev)
        ident = "%s|%s|%s" % (socket.gethostname(), getattr(props, "uuid", ""), getattr(props, "pci_bus_id", dev_index))
        mine = torch.tensor([zlib.crc32(ident.encode()) & 0x7FFFFFFF, dev_index], dtype=torch.int64, device=dev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        distinct_devices = len({(int(t[0]), int(t[1])) for t in everyone})
        rccl_ranks = distinct_devices if dist.get_backend() == "nccl" else 0
        if os.environ.get("E3DGS_BENCH_STRICT") == "1" and (dist.get_backend() != "nccl" or distinct_devices != world):
            raise SystemExit("E3DGS_BENCH_STRICT=1: %d ranks over backend %s on %d distinct devices -- expected one RCCL rank "
                             "per GPU" % (world, dist.get_backend(), distinct_devices))

    cfg_name = args.config or "cfg3_1M_1080p_event"
    N, W, H, deblur = CONFIGS[cfg_name]
    L = _lib.lib()

    # ---- scene, cameras, ground truth (render of the jittered scene) -- all resident in HBM
    params = synth.make_scene(N, "trained", seed=0, device=dev)
    K = 64
    k0 = 3 * rank                                            # every rank its own triplet
    cam_int = orbit_camera(k0, K, W, H, device=dev)
    cam_now = orbit_camera(k0, K, W, H, device=dev, daz=0.005)
    cam_next = orbit_camera(k0, K, W, H, device=dev, daz=0.015)
    bg = torch.zeros(3, device=dev)
    gt_params = dict(params)
    gt_params["xyz"] = params["xyz"] + 0.01 * torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    gt_tr = EventTrainer(gt_params, dev)
    # ground truth arrives as 8-bit images in the reference (PIL -> /255, utils/general_utils.py:21-27): quantise, so
    # that unchanged pixels of the two event frames compare EQUAL and rho = mean(D* != 0) < 1 as on real data
    gts = [(torch.round(gt_tr.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous()
           for c in (cam_int, cam_now, cam_next)]
    gt_blur = (0.5 * (gts[0] + gts[2])).contiguous() if deblur else None
    del gt_tr
    if args.only_trained:
        rec = measure_trained_random_camera(params, gt_params, dev, W, H, bg, L, deblur, args.trained_steps,
                                            args.trained_timed, K, trace_dir=os.environ.get("E3DGS_TRAINED_TRACE_DIR"))
        print(json.dumps({"trained_random_camera": rec, "config": {"workload": cfg_name, "gaussians": N, "width": W,
                                                                   "height": H}}))
        return
    trainer = EventTrainer(params, dev)

    def one_step():
        # (step_nocopy: the returned scalars are a view of one of two alternating blocks -- valid until the step after the
        # next one, and read once after the loops; step() would add a copy kernel per iteration for a value nobody reads)
        return trainer.step_nocopy(cam_int, cam_now, cam_next, gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)

    # Several ranks: the overlapped / factorised gradient exchange has only ever run over gloo (the development boxes
    # have one GPU).  If its first step raises on this backend, every rank falls back to the plain schedule (one
    # blocking mean of the gradient buffer per chunk, no side stream) instead of losing the run; the JSON says which.
    dp_schedule = "single rank" if world == 1 else "side-stream SH exchange + factorised SH gradient"
    dp_fallback = False
    warm = args.warmup
    if world > 1:
        err = None
        try:
            if os.environ.get("E3DGS_BENCH_FORCE_FALLBACK") == "1":          # test hook for the branch below
                raise RuntimeError("forced by E3DGS_BENCH_FORCE_FALLBACK")
            loss = one_step()
            torch.cuda.synchronize()
        except Exception as ex:           # noqa: BLE001 -- any failure of the first exchange
            err = "%s: %s" % (type(ex).__name__, str(ex)[:200])
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag[0]) and os.environ.get("E3DGS_BENCH_STRICT") == "1":
            # strict runs never trade a failing exchange for a slower schedule: they stop, with the first error on record
            raise SystemExit("E3DGS_BENCH_STRICT=1: the overlapped / factorised gradient exchange failed on rank %d: %s"
                             % (rank, err or "(another rank failed)"))
        if int(flag[0]):
            os.environ["E3DGS_FACTORIZE_SH"] = "0"
            os.environ["E3DGS_OVERLAP"] = "0"
            trainer = EventTrainer(params, dev)
            dp_schedule = "fallback (plain chunked all-reduce): " + (err or "another rank failed")
            dp_fallback = True
        else:
            warm = max(0, warm - 1)       # the probe step was the first warm-up step
    # Python's cyclic garbage collector is kept out of the measurements (as timeit does): a generation-2 pass of this
    # process takes 30-60 ms -- measured: it landed inside a 20-step window in two runs of three and turned 2.0 ms per
    # step into 4 -- and nothing the iteration allocates needs it (no reference cycles; tensors are freed by refcount).
    # Collected BEFORE the warm-up steps: between warm-up and timed region the GPU then idles only for the contract's
    # barrier + synchronize.  A 30-60 ms pause there lets the clocks drop and the timed iterations pay the ramp: same-box
    # A/B of a 20-step window (E3DGS_BENCH_GC_LATE=1 = collection between warm-up and timed region, as before round 4):
    # 2.53-2.55 ms per step against 2.43-2.44, render_bwd_kernel 0.863 against 0.823 ms.
    import gc
    gc.collect()
    gc.disable()
    for _ in range(warm):
        loss = one_step()
    torch.cuda.synchronize()
    if os.environ.get("E3DGS_BENCH_GC_LATE") == "1":      # (A/B hook: the old placement of the collection)
        gc.enable(); gc.collect(); gc.disable()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # inside the timed region only the dominant kernel (slot 6, render_bwd_kernel) is bracketed with HIP events:
    # every timed slot costs two event packets per launch on the queue
    # ... and only on every fourth timed iteration: the two event packets around the launch leave the GPU idle for ~6 us
    # each (kernel trace: the only two gaps of an iteration), i.e. bracketing every launch costs the iteration 0.5 %
    DOMINANT_SLOT = 6
    EVENT_EVERY = max(1, int(os.environ.get("E3DGS_BENCH_EVENT_EVERY", "4")))
    L.e3dgs_profile_enable(1 << DOMINANT_SLOT)          # (resets the counters)
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for it_ in range(args.steps):
        L.e3dgs_profile_select((1 << DOMINANT_SLOT) if it_ % EVENT_EVERY == 0 else 0)
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    device_allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0   # hipMalloc calls while timed
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- per-kernel durations from the HIP events recorded on the launch stream during the timed region
    import ctypes as C
    def read_slots():
        out = {}
        for slot in range(8):
            ms, n = C.c_double(0), C.c_int(0)
            L.e3dgs_profile_query(slot, C.byref(ms), C.byref(n))
            out[L.e3dgs_profile_slot_name(slot).decode()] = (ms.value, n.value)
        return out
    timed = read_slots()
    L.e3dgs_profile_enable(0)
    loss_val = float(loss[0].item())
    # per-stage table: the same steps again, outside the timed region, with every slot bracketed
    L.e3dgs_profile_enable(0xFF)
    # ... and the optimizer (EventTrainer.apply_update: SH-gradient rebuild + Adam kernels) with events on torch's stream
    opt_events = []
    real_apply = trainer.apply_update

    def timed_apply(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); real_apply(*a, **k); e1.record()
        opt_events.append((e0, e1))
    trainer.apply_update = timed_apply
    for _ in range(max(3, args.steps // 4)):
        one_step()
    torch.cuda.synchronize()
    trainer.apply_update = real_apply
    kern = read_slots()
    L.e3dgs_profile_enable(0)
    opt_ms = sum(a.elapsed_time(b) for a, b in opt_events) / max(len(opt_events), 1)
    dom_name = L.e3dgs_profile_slot_name(DOMINANT_SLOT).decode()
    if timed[dom_name][1]:
        kern[dom_name] = timed[dom_name]          # the roofline figure uses the timed-region measurement

    # workload statistics: one launch of every stage covers the three views of the iteration
    V = 3
    vis = int((trainer.last_radii > 0).sum())
    I1 = trainer.render_raw(cam_int, bg)["num_rendered"]
    I = sum(trainer.render_raw(c, bg)["num_rendered"] for c in (cam_int, cam_now, cam_next))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    npx = W * H

    stages = {}
    for name, (ms, n) in kern.items():
        if n:
            avg = ms / n
            b = algorithmic_bytes(name, N * V, I, T * V, npx * V)
            stages[name] = {"avg_ms": round(avg, 4), "launches": n, "alg_GB": round(b / 1e9, 4),
                            "alg_GBps": round(b / 1e9 / (avg / 1e3), 1)}
    if opt_events and world == 1:
        ob = 28 * (FLOATS_PER_GAUSSIAN * N + 1)          # torch.optim.Adam: read g, p, m, v; write p, m, v
        stages["optimizer"] = {"avg_ms": round(opt_ms, 4), "launches": len(opt_events), "alg_GB": round(ob / 1e9, 4),
                               "alg_GBps": round(ob / 1e9 / (opt_ms / 1e3), 1)}
    # ---- committed PMC profile of this workload (profiles/traffic.json = summary.json of profiles/collect.py): measured
    # HBM-side bytes per stage next to the algorithmic ones, instruction mix of the compositing kernels.  The profile
    # names the sources it was taken on; if they are not the sources this run was built from, its numbers are reported
    # as stale and the derived figures are dropped.
    prof, prof_stale = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and cfg_name == "cfg3_1M_1080p_event":
        try:
            prof = json.load(open(tpath))
            prof_stale = prof.get("source_fingerprint") != source_fingerprint()
        except Exception:
            prof = None
    # per-iteration view of the stages (a stage may be several launch groups per iteration)
    n_iter_prof = max(kern.get("render_fwd", (0, 1))[1], 1)
    for name, (ms, n) in kern.items():
        if name in stages:
            stages[name]["ms_per_iter"] = round(ms / n_iter_prof, 4) if name != dom_name or not timed[dom_name][1] else \
                round(timed[dom_name][0] / max(timed[dom_name][1], 1), 4)
    for st_ in stages.values():
        st_.setdefault("ms_per_iter", st_["avg_ms"])                 # (the optimizer: one event pair per iteration)
    if prof and not prof_stale:
        for name, rec in prof.get("stages", {}).items():
            if name in stages and rec.get("bytes_per_iteration") and stages[name].get("ms_per_iter"):
                stages[name]["pmc_GB_per_iter"] = round(rec["bytes_per_iteration"] / 1e9, 4)
                stages[name]["pmc_GBps"] = round(rec["bytes_per_iteration"] / 1e9 / (stages[name]["ms_per_iter"] / 1e3), 1)
    dominant = dom_name if dom_name in stages else None
    # ---- effective shader clock INSIDE the two compositing kernels (one extra iteration with the debug trace armed: every
    # tile's wave stamps s_memtime -- shader cycles -- and the 100 MHz wall clock at its first and last instruction)
    clocks = measure_kernel_clocks(trainer, (cam_int, cam_now, cam_next), bg, L, W, H) if world == 1 else None
    roofline = None
    if dominant:
        s = stages[dominant]
        traffic = None
        if prof and not prof_stale:
            traffic = prof.get("traffic_per_kernel", {}).get(dominant + "_kernel", {}).get("hbm_bytes_per_launch")
        # What bounds compositing is instruction ISSUE, not HBM: a wave64 VALU instruction occupies its SIMD-32
        # for 2 cycles (MI355X_MICROARCH.md), so N wave-instructions on 1024 SIMDs need at least 2 N / 1024 cycles -- at the
        # nominal 2.4 GHz (`frac_nominal`) and at the clock the kernel itself ran at (`frac_at_measured_clock`).  The counts
        # come from the committed PMC profile of THESE sources (null when it is stale); both floors are lower bounds on the
        # kernel's time by construction (frac <= 1; tests/test_profiles.py).
        def valu_model(kname, ms):
            sq = (prof or {}).get("sq", {}).get(kname, {}) if (prof and not prof_stale) else {}
            n = sq.get("SQ_INSTS_VALU")
            ghz = (clocks or {}).get(kname, {}).get("ghz_median")
            rec = {"avg_launch_ms": round(ms, 4), "valu_wave_instructions": n, "salu_wave_instructions": sq.get("SQ_INSTS_SALU"),
                   "branch_wave_instructions": sq.get("SQ_INSTS_BRANCH"), "lds_wave_instructions": sq.get("SQ_INSTS_LDS"),
                   "transcendental_wave_instructions": sq.get("SQ_INSTS_VALU_TRANS"), "clock": (clocks or {}).get(kname),
                   "floor_ms_nominal": None, "frac_nominal": None, "floor_ms_at_measured_clock": None,
                   "frac_at_measured_clock": None}
            if n:
                f_nom = 2.0 * n / 1024.0 / 2.4e9 * 1e3
                rec["floor_ms_nominal"], rec["frac_nominal"] = round(f_nom, 4), round(f_nom / ms, 3)
                if ghz:
                    f_clk = 2.0 * n / 1024.0 / (ghz * 1e9) * 1e3
                    rec["floor_ms_at_measured_clock"], rec["frac_at_measured_clock"] = round(f_clk, 4), round(f_clk / ms, 3)
                # the scalar unit is shared by the 4 SIMDs of a CU (one SALU / branch issue per cycle per CU)
                if sq.get("SQ_INSTS_SALU") and ghz:
                    sc = (sq["SQ_INSTS_SALU"] + sq.get("SQ_INSTS_BRANCH", 0)) / 256.0 / (ghz * 1e9) * 1e3
                    rec["scalar_unit_floor_ms_at_measured_clock"] = round(sc, 4)
                    rec["scalar_unit_frac_at_measured_clock"] = round(sc / ms, 3)
            return rec
        kernels = {}
        for kn, st_name in (("render_fwd_kernel", "render_fwd"), ("render_bwd_kernel", "render_bwd")):
            if st_name in stages:
                kernels[kn] = valu_model(kn, stages[st_name]["avg_ms"])
        dk = kernels.get(dominant + "_kernel", {})
        roofline = {"kernel": dominant + "_kernel", "bound": "valu", "achieved": s["alg_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(s["alg_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_stale": prof_stale, "profile_source_fingerprint": prof.get("source_fingerprint") if prof else None,
                    "avg_launch_ms": s["avg_ms"], "alg_bytes_per_launch": int(s["alg_GB"] * 1e9),
                    "views_per_launch": V, "frac_nominal": dk.get("frac_nominal"),
                    "frac_at_measured_clock": dk.get("frac_at_measured_clock"),
                    "effective_clock_ghz": (dk.get("clock") or {}).get("ghz_median"), "kernels": kernels,
                    "note": "one launch composites the 3 views of the iteration (HIP events on the launch stream, timed "
                            "region).  `frac` is the contract's figure: algorithmic bytes / time against the 8 TB/s HBM peak -- "
                            "small by construction, the kernel is bound by instruction issue (SURVEY 8d, DESIGN.md section 5).  "
                            "frac_nominal / frac_at_measured_clock: its VALU wave-instructions (PMC, committed profile) at 2 "
                            "cycles each on 1024 SIMDs, at 2.4 GHz / at the shader clock measured inside the kernel, over its "
                            f"duration; alpha evaluations/s = {256.0 * I / (s['avg_ms'] / 1e3) / 1e9:.1f} G/s"}

    # ---- the contrast-only sub-step north_star words the metric by: renders #2 and #3 forward,
    # differentialable_event_simu + L1 on the pair, backward through both renders; no intensity render, no optimizer.
    # Measured after the timed region (it does not enter `value`).
    contrast = None
    if world == 1 and not args.no_substep:
        from event_3dgs_amd import losses, rasterizer
        v = trainer.views
        two = [trainer._settings(c, bg) for c in (cam_now, cam_next)]
        g2 = {k: torch.empty_like(t) for k, t in trainer.grads.items()}
        out2 = dict(means3D=g2["xyz"], sh=g2["features"], opacities=g2["opacity"], scales=g2["scaling"], rots=g2["rotation"])
        lbuf = None

        def contrast_step():
            nonlocal lbuf
            raw2 = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], two,
                                            flags=trainer.FWD_FLAGS, pool=trainer._pool)
            im = raw2["color"]
            if lbuf is None:
                lbuf = (torch.empty(8, device=dev), torch.empty_like(im[0]), torch.empty_like(im), torch.empty(
                    L.e3dgs_event_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev))
            # the intensity slot gets its own target: its L1 is 0 and only the contrast term drives the two renders
            # (both renders enter the loss through one luminance: rank-1 pixel gradients, as EventTrainer hands them over)
            losses.event_loss_raw(gts[0], im[0], im[1], trainer.c, gts[0], gts[1], gts[2],
                                  out=(lbuf[0], lbuf[1], lbuf[2][0], lbuf[2][1], lbuf[3]), rank1=trainer.rank1)
            rasterizer.backward_multi(raw2, lbuf[2], out2, rank1={0: rasterizer.LUV_WEIGHTS, 1: rasterizer.LUV_WEIGHTS}
                                      if trainer.rank1 else None)
        for _ in range(2):
            contrast_step()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        reps = max(5, min(args.steps, 20))
        for _ in range(reps):
            contrast_step()
        torch.cuda.synchronize()
        cms = 1e3 * (time.perf_counter() - tc) / reps
        contrast = {"ms": round(cms, 3), "per_s": round(1e3 / cms, 1), "renders": 2,
                    "what": "renders #2,#3 fwd + log-contrast L1 (train.py:159-178) + backward of both; no optimizer"}

    # ---- the iteration as the reference's DATASETS shape it: the event camera `index` is read with the training camera's
    # extrinsics (scene/dataset_readers.py:157), so render #1 (train.py:144) and render #2 (:159) are the same view.
    # EventTrainer renders it once (compute_gradients: shared pose).  The headline above keeps three DISTINCT cameras
    # (three renders of work); this is the same step with cam_now at cam_int's pose.  After the timed region.
    shared_pose = None
    if world == 1 and not args.no_substep:
        cam_now_same = orbit_camera(k0, K, W, H, device=dev)           # a second camera object, the pose of cam_int
        gt_same = trainer.render_raw(cam_now_same, bg)["color"].clone()
        sp_step = lambda: trainer.step(cam_int, cam_now_same, cam_next, gts[0], gt_same, gts[2], bg, gt_blur=gt_blur)
        before = trainer.shared_pose_iterations
        for _ in range(3):
            sp_step()
        torch.cuda.synchronize()
        tsp = time.perf_counter()
        reps = max(5, min(args.steps, 20))
        for _ in range(reps):
            sp_step()
        torch.cuda.synchronize()
        sms = 1e3 * (time.perf_counter() - tsp) / reps
        shared_pose = {"ms": round(sms, 3), "per_s": round(1e3 / sms, 1), "renders": 2,
                       "taken": trainer.shared_pose_iterations - before == reps + 3,
                       "what": "the full event iteration (loss on three images, backward, Adam) when render #1 and render #2 "
                               "share a pose, as in the reference's datasets: that view is rendered once"}

    # ---- TOLERANCE MODE, beside the headline and never instead of it: the same iteration with E3DGS_FLAG_FAST_EXP (hardware
    # exp2 in the compositing kernels; tests/test_hip_parity.py::test_fast_exp_mode_is_a_counted_tolerance_mode) -- what
    # the bit-exact forward costs.  A fresh trainer on the same parameters, after the timed region.
    fast_exp = None
    if world == 1 and not args.no_substep:
        tf = EventTrainer(params, dev, fast_exp=True)
        f_step = lambda: tf.step(cam_int, cam_now, cam_next, gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)
        for _ in range(4):
            f_step()
        torch.cuda.synchronize()
        tfe = time.perf_counter()
        reps = max(5, min(args.steps, 20))
        for _ in range(reps):
            f_step()
        torch.cuda.synchronize()
        fms = 1e3 * (time.perf_counter() - tfe) / reps
        # parity evidence of the mode on this workload: image against the exact mode on the same parameters
        ex = trainer.render_raw(cam_int, bg)
        tf.flat.copy_(trainer.flat)
        fa = tf.render_raw(cam_int, bg)
        dd = (ex["color"] - fa["color"]).abs()
        fast_exp = {"ms": round(fms, 3), "per_s": round(1e3 / fms, 1), "default": False,
                    "image_max_abs_vs_exact": float(dd.max()), "values_off_by_1e-4": int((dd > 1e-4).sum()),
                    "values": int(dd.numel()), "radii_equal": bool(torch.equal(ex["radii"], fa["radii"])),
                    "instances_equal": ex["num_rendered"] == fa["num_rendered"],
                    "what": "the full event iteration with E3DGS_FLAG_FAST_EXP (v_exp_f32 instead of the bit-reproducible "
                            "polynomial in render_fwd / its decisions in render_bwd): a tolerance mode, default OFF"}
        del tf

    # ---- BASELINE configs[4] (the per-rank workload of the 8-GPU configuration: 2 M Gaussians, 1080p) is touched whenever
    # several ranks run: measured after the timed region, never part of `value` (the weak-scaling metric keeps cfg3 per rank)
    cfg5 = None
    if world > 1 and os.environ.get("E3DGS_BENCH_SKIP_CFG5") != "1":
        N5 = CONFIGS["cfg5_2M_1080p_event"][0]
        p5 = synth.make_scene(N5, "trained", seed=0, device=dev)
        t5 = EventTrainer(p5, dev)
        s5 = lambda: t5.step(cam_int, cam_now, cam_next, gts[0], gts[1], gts[2], bg)
        for _ in range(3):
            s5()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t5s = time.perf_counter()
        for _ in range(10):
            s5()
        torch.cuda.synchronize(); dist.barrier()
        tt = torch.tensor([(time.perf_counter() - t5s) / 10 * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cfg5 = {"workload": "cfg5_2M_1080p_event", "gaussians_per_rank": N5, "ms_per_step": round(float(tt.item()), 3),
                "iters_per_s_whole_job": round(world * 1e3 / float(tt.item()), 2),
                "what": "BASELINE configs[4]: every rank trains its own camera triplet of a 2 M-Gaussian replica, RCCL "
                        "gradient exchange as in the timed run; 10 steps after the timed region"}
        del t5, p5

    # ---- several ranks: what the exchange costs, so that one JSON line diagnoses an 8-GPU run.  After the timed region:
    #   exposed_comm_ms  = step time - time of the same step without any exchange (sync_grads=False: local Adam)
    #   allreduce_ms / allgather_ms = the two collectives of the default schedule alone, on their real buffers
    comm = None
    if world > 1:
        def timed_ms(fn, reps):
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / reps * 1e3], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        from event_3dgs_amd import parallel as par
        nonsh = torch.zeros(11 * N + 1, device=dev)
        colour = torch.zeros(9 * N + 9, device=dev)
        gathered = torch.zeros(world, colour.numel(), device=dev)
        ar_ms = timed_ms(lambda: par.allreduce_mean_(nonsh), 5)
        ag_ms = timed_ms(lambda: par.allgather_async_(gathered, colour).wait(), 5)
        sh = (nonsh.numel() + world - 1) // world
        stag, shard = torch.zeros(world * sh, device=dev), torch.zeros(sh, device=dev)
        rs_ms = timed_ms(lambda: (par.reduce_scatter_mean_async_(shard, stag).wait(),
                                  par.allgather_flat_async_(stag, shard).wait()), 5)
        local_ms = timed_ms(lambda: trainer.step(cam_int, cam_now, cam_next, gts[0], gts[1], gts[2], bg, gt_blur=gt_blur,
                                                 sync_grads=False), max(3, args.steps // 2))
        comm = {"exposed_comm_ms": round(1e3 * dt / args.steps - local_ms, 4), "local_step_ms": round(local_ms, 4),
                "allreduce_ms": round(ar_ms, 4), "allreduce_bytes": 4 * nonsh.numel(),
                "allgather_ms": round(ag_ms, 4), "allgather_bytes_per_rank": 4 * colour.numel(),
                "reduce_scatter_plus_allgather_ms": round(rs_ms, 4),
                "nonsh_schedule": trainer.dp_schedule, "sh_exchange": "factorised" if trainer.factorize_sh else "allreduce",
                "sh_exchange_bytes": par.sh_exchange_bytes(world, N),
                "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                       "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY",
                                                       "E3DGS_DP_SCHEDULE", "E3DGS_FACTORIZE_SH", "E3DGS_OVERLAP")}}
        del nonsh, colour, gathered, stag, shard

    # ---- the drop-in path (the reference's unmodified train.py on the two drop-in packages), cfg3 and cfg2
    dropin = None
    if world == 1 and not args.no_substep:
        dropin = {cfg_name: measure_dropin(params, (cam_int, cam_now, cam_next), gts, bg, dev, L)}
        if cfg_name == "cfg3_1M_1080p_event":
            N2, W2, H2, _ = CONFIGS["cfg2_200k_800px"]
            p2 = synth.make_scene(N2, "trained", seed=0, device=dev)
            cams2 = [orbit_camera(0, K, W2, H2, device=dev, daz=d) for d in (0.0, 0.005, 0.015)]
            t2 = EventTrainer(p2, dev)
            gts2 = [(torch.round(t2.render_raw(c_, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous() for c_ in cams2]
            dropin["cfg2_200k_800px"] = measure_dropin(p2, cams2, gts2, bg, dev, L)
            for _ in range(5):
                t2.step(*cams2, *gts2, bg)
            torch.cuda.synchronize(); tq = time.perf_counter()
            for _ in range(20):
                t2.step(*cams2, *gts2, bg)
            torch.cuda.synchronize()
            dropin["cfg2_200k_800px"]["fused_step_ms"] = round(1e3 * (time.perf_counter() - tq) / 20, 3)
            del t2, p2, gts2
        dropin[cfg_name]["fused_step_ms"] = round(1e3 * dt / args.steps, 3)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(trainer, (cam_int, cam_now, cam_next), bg, W, H, args.cpu_rows, args.torch_rows)

    # ---- how to read the headline (after the timed region, never part of `value`):
    #   sustained          >= 500 more iterations of the same trainer (the scene keeps training: its instance count drifts),
    #                      with the board's power / clocks sampled meanwhile (rocm-smi) -- a 20-step window is 47 ms;
    #   static_workload    the same iteration on FROZEN parameters (every learning rate 0: Adam runs, nothing moves), so that
    #                      two runs time the same lists;
    #   reference_binning  the iteration with the reference's rectangle binning (E3DGS_TILE_CULL=0: "identical tile
    #                      assignment" read literally; the default drops the instances that provably touch no pixel) and
    #                      its instance count -- what the exact tile culling is worth.
    # (LAST of the legs: it trains the benchmark's own trainer on, and the parity leg above wants the state the timed region left)
    reading = None
    if world == 1 and not args.no_substep:
        reading = {}
        n_sus = max(500, args.steps)
        samples, stop = [], [False]

        def sampler():
            import subprocess
            while not stop[0]:
                try:
                    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                                       text=True, timeout=10)
                    card = next(iter(json.loads(r.stdout).values()))
                    samples.append({k: v for k, v in card.items() if any(
                        t in k.lower() for t in ("power", "sclk", "mclk", "temperature (sensor junction)"))})
                except Exception:           # noqa: BLE001 -- no rocm-smi, no JSON: the leg still reports its timing
                    return
        import threading
        th = threading.Thread(target=sampler, daemon=True)
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        th.start()
        ts = time.perf_counter()
        for _ in range(n_sus):
            one_step()
        torch.cuda.synchronize()
        sus_ms = 1e3 * (time.perf_counter() - ts) / n_sus
        stop[0] = True
        th.join(timeout=15)
        # the stages at the END of the window: a slower sustained figure can be the clocks or the workload (the scene the
        # extra iterations train is not the scene the timed region saw: opacities move, pixels saturate later, walks deepen)
        L.e3dgs_profile_enable(0xFF)
        for _ in range(5):
            one_step()
        torch.cuda.synchronize()
        end_stages = {k: round(ms / n, 4) for k, (ms, n) in read_slots().items() if n}
        L.e3dgs_profile_enable(0)

        def num(v):
            try:
                return float(str(v).strip("()").lower().replace("mhz", "").replace("w", ""))
            except ValueError:
                return None
        power = [num(v) for smp in samples for k, v in smp.items() if "power" in k.lower() and num(v) is not None]
        sclk = [num(v) for smp in samples for k, v in smp.items() if "sclk" in k.lower() and num(v) is not None]
        reading["sustained"] = {"steps": n_sus, "ms_per_step": round(sus_ms, 3), "per_s": round(1e3 / sus_ms, 1),
                                "tile_instances_3views_after": int(sum(trainer.render_raw(c, bg)["num_rendered"] for c in
                                                                       (cam_int, cam_now, cam_next))),
                                "board_power_w_mean": round(sum(power) / len(power), 1) if power else None,
                                "board_power_w_max": max(power) if power else None,
                                "sclk_mhz_reported_mean": round(sum(sclk) / len(sclk), 1) if sclk else None,
                                "rocm_smi_samples": len(samples), "last_sample": samples[-1] if samples else None,
                                "stage_avg_ms_at_the_end": end_stages,
                                # why this leg is slower than the headline: the SAME triplet for 500 more iterations -- the scene
                                # overfits three views and the walks deepen (compare trained_random_camera.walk_statistics)
                                "walk_statistics_at_the_end": _walk_statistics(trainer, (cam_int, cam_now, cam_next), bg, W, H),
                                "stage_avg_ms_in_the_timed_region_s_table": {k: v["avg_ms"] for k, v in stages.items()},
                                "in_kernel_clock_ghz": {k: (v.get("clock") or {}).get("ghz_median") for k, v in
                                                        ((roofline or {}).get("kernels") or {}).items()}}
        zero = dict(position_lr_init=0.0, position_lr_final=0.0, feature_lr=0.0, opacity_lr=0.0, scaling_lr=0.0,
                    rotation_lr=0.0, c_lr=0.0)

        def leg(tr, reps):
            st = lambda: tr.step_nocopy(cam_int, cam_now, cam_next, gts[0], gts[1], gts[2], bg, gt_blur=gt_blur)
            for _ in range(4):
                st()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                st()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t1) / reps
        tz = EventTrainer(params, dev, **zero)
        z_ms = leg(tz, 100)
        moved = float((tz.flat - EventTrainer(params, dev, **zero).flat).abs().max())
        reading["static_workload"] = {"steps": 100, "ms_per_step": round(z_ms, 3), "per_s": round(1e3 / z_ms, 1),
                                      "parameters_moved_by": moved,
                                      "what": "the event iteration with every learning rate 0 (same kernels, same traffic, "
                                              "frozen lists): the figure to compare between runs / boxes / rounds"}
        del tz
        tr_ref = EventTrainer(params, dev, tile_cull=0, **zero)
        r_ms = leg(tr_ref, 40)
        I_ref1 = tr_ref.render_raw(cam_int, bg)["num_rendered"]
        I_ref = sum(tr_ref.render_raw(c, bg)["num_rendered"] for c in (cam_int, cam_now, cam_next))
        I_cull = sum(EventTrainer(params, dev, **zero).render_raw(c, bg)["num_rendered"] for c in (cam_int, cam_now, cam_next))
        reading["reference_binning_iteration"] = {
            "ms_per_step": round(r_ms, 3), "per_s": round(1e3 / r_ms, 1), "frozen_parameters": True,
            "tile_instances_reference_binning": I_ref1, "tile_instances_reference_binning_3views": I_ref,
            "tile_instances_exact_culling_3views": I_cull, "vs_static_workload": round(r_ms / z_ms, 3),
            "what": "E3DGS_TILE_CULL=0: every (splat, tile) pair of the reference's rectangles is emitted, sorted, walked; "
                    "images, radii and gradients are the default mode's (tests: bit-exact lists in this mode)"}
        del tr_ref
        if roofline and dominant:
            b_ref = algorithmic_bytes(dominant, N * V, I_ref, T * V, npx * V)
            roofline["frac_on_reference_instances"] = round(b_ref / 1e9 / (stages[dominant]["avg_ms"] / 1e3) / HBM_PEAK_GBS, 5)
            roofline["alg_bytes_on_reference_instances"] = int(b_ref)

    # ---- the state a trainer lives in (train.py:116-131 draws a random camera every iteration; the headline renders one
    # triplet of a scene that has never been trained): after the timed region, never part of `value`
    trained = None
    if world == 1 and not args.no_substep and args.trained_steps > 0:
        del trainer
        torch.cuda.empty_cache()
        trained = measure_trained_random_camera(params, gt_params, dev, W, H, bg, L, deblur, args.trained_steps,
                                                args.trained_timed, K)
        trainer = EventTrainer(params, dev)        # (the fields below read configuration flags off a trainer)

    # gradient exchange per iteration and rank: mean of the non-SH groups (+ of the SH gradient unless it is rebuilt
    # from the all-gathered per-view colour gradients, EventTrainer.factorize_sh)
    if world == 1:
        grad_ar_bytes = grad_ag_bytes = 0
    elif trainer.factorize_sh:
        grad_ar_bytes, grad_ag_bytes = 4 * (11 * N + 1), 4 * (9 * N + 9)      # 3 views x 3 colour channels (+ 3 centres)
    else:
        grad_ar_bytes, grad_ag_bytes = 4 * (FLOATS_PER_GAUSSIAN * N + 1), 0
    if rank == 0:
        iters_per_s = args.steps * world / dt
        out = {
            "metric": "train iters/s fwd+bwd @1M Gaussians 1080p; PSNR vs ref; HBM GB/s %peak",
            "value": round(iters_per_s, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name, "gaussians": N, "width": W, "height": H, "visible": vis,
                       "tile_instances": I1, "tile_instances_3views": I,
                       "tile_instances_reference_binning": ((reading or {}).get("reference_binning_iteration") or {}).get(
                           "tile_instances_reference_binning"),
                       "tiles": T, "renders_per_iter": 3, "deblur": deblur,
                       "parallelism": f"view-dp{world}", "grad_allreduce_bytes": grad_ar_bytes,
                       "sh_colour_allgather_bytes_per_rank": grad_ag_bytes,
                       "sh_exchange_on_side_stream": bool(trainer.overlap_features), "dp_schedule": dp_schedule,
                       "loss": round(loss_val, 6)},
            "roofline": roofline, "stages": stages, "contrast_only_substep": contrast,
            "sustained": (reading or {}).get("sustained"), "static_workload": (reading or {}).get("static_workload"),
            "reference_binning_iteration": (reading or {}).get("reference_binning_iteration"),
            "trained_random_camera": trained,
            "dropin_autograd_step": dropin, "shared_pose_iteration": shared_pose, "fast_exp_iteration": fast_exp,
            "cfg5_per_rank_workload": cfg5, "cpu_baseline": cpu_baseline,
            "device_allocs_in_timed_region": device_allocs,
            # ranks the communicator itself reports (1: no process group) and whether the factorised / overlapped
            # exchange had to be replaced by the plain schedule (a failing exchange must not hide in a slower number)
            "rccl_ranks": rccl_ranks, "distinct_devices_in_the_communicator": distinct_devices,
            "world_size": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "comm_backend": (dist.get_backend() if (world > 1 and dist.is_initialized()) else None),
            "dp_fallback": dp_fallback, "comm": comm,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def measure_trained_random_camera(params, gt_params, dev, W, H, bg, L, deblur, n_train, n_timed, K=64, trace_dir=None):
    """The iteration in the state a trainer actually lives in: K orbit cameras (each with its event pair and its 8-bit
    ground-truth frames, all resident in HBM), a fresh random camera per iteration with the reference's draw
    (train.py:116-131 = fit.sample_index), `n_train` training iterations, THEN `n_timed` timed ones (still training, still
    drawing).  Next to it, so that the gap to the headline can be read: (a) the same random draw on the UNTRAINED scene
    with every learning rate 0 (what random cameras alone cost: camera / pair-count caches, capacity keys, per-view instance
    counts that differ from the headline's triplet), (b) the trained end state, frozen, on the headline's fixed triplet (what
    training alone costs per instance).  Stage tables with all slots bracketed are taken after each timed window."""
    import ctypes as C
    import random
    import torch
    from event_3dgs_amd import fit
    from event_3dgs_amd.cameras import orbit_camera
    from event_3dgs_amd.train_step import EventTrainer

    cams = [tuple(orbit_camera(k, K, W, H, device=dev, daz=d) for d in (0.0, 0.005, 0.015)) for k in range(K)]
    gt_tr = EventTrainer(gt_params, dev)
    quant = lambda c: (torch.round(gt_tr.render_raw(c, bg)["color"].clamp(0, 1) * 255.0) / 255.0).contiguous()
    gts = [tuple(quant(c) for c in trip) for trip in cams]
    blur = [(0.5 * (g[0] + g[2])).contiguous() for g in gts] if deblur else None
    del gt_tr
    zero = dict(position_lr_init=0.0, position_lr_final=0.0, feature_lr=0.0, opacity_lr=0.0, scaling_lr=0.0,
                rotation_lr=0.0, c_lr=0.0)

    def read_slots():
        out = {}
        for slot in range(8):
            ms, n = C.c_double(0), C.c_int(0)
            L.e3dgs_profile_query(slot, C.byref(ms), C.byref(n))
            if n.value:
                out[L.e3dgs_profile_slot_name(slot).decode()] = round(ms.value / n.value, 4)
        return out

    marks = []
    mark_pos = torch.zeros(64, 3, device=dev)
    mark_mat = torch.eye(4, device=dev).contiguous()

    def mark(tag):
        """A launch of mark_visible_kernel -- which no iteration contains -- in stream order: profiles/collect_trained.py cuts
        the kernel trace of this process at these launches (the k-th one is marks[k])."""
        from event_3dgs_amd.rasterizer import _mark_visible
        _mark_visible(mark_pos, mark_mat, mark_mat)
        marks.append(tag)

    def window(tr, draw, n, n_stage=40, tag=""):
        """n timed iterations (wall clock, nothing bracketed), then n_stage with every stage bracketed."""
        inst = []

        def it():
            k = draw()
            tr.step_nocopy(*cams[k], *gts[k], bg, gt_blur=blur[k] if blur else None)
            inst.append(tr._instances_per_view * 3)      # (the count the host has just read for this iteration's pass)
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        del inst[:]
        r0 = tr.count_retries
        mark(tag + ":timed:begin")
        t0 = time.perf_counter()
        for _ in range(n):
            it()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        mark(tag + ":timed:end")
        retries = tr.count_retries - r0
        I_mean = sum(inst) / max(len(inst), 1)
        I_min, I_max = (min(inst), max(inst)) if inst else (0, 0)
        L.e3dgs_profile_enable(0xFF)
        del inst[:]
        for _ in range(n_stage):
            it()
        torch.cuda.synchronize()
        st = read_slots()
        L.e3dgs_profile_enable(0)
        I_stage = sum(inst) / max(len(inst), 1)
        per = {k: round(1e9 * st[k] / I_stage, 1) for k in ("render_fwd", "render_bwd") if k in st and I_stage}   # ps / instance
        return {"ms_per_step": round(ms, 3), "per_s": round(1e3 / ms, 1), "steps": n, "count_retries": retries,
                "tile_instances_3views_mean": int(I_mean), "tile_instances_3views_min": int(I_min),
                "tile_instances_3views_max": int(I_max), "stage_avg_ms": st,
                "tile_instances_3views_mean_in_the_stage_window": int(I_stage), "compositing_ps_per_instance": per}

    rng = random.Random(0)
    draw = lambda: fit.sample_index(K, "event", rng.randint)
    fixed = lambda: 0
    rec = {"cameras": K, "draw": "train.py:116-131 (fit.sample_index, event mode), python random.Random(0)",
           "training_steps_before_the_timed_window": n_train}
    # (a) random cameras alone: untrained scene, frozen
    tz = EventTrainer(params, dev, **zero)
    rec["untrained_frozen_random_camera"] = window(tz, draw, max(50, n_timed // 2), tag="untrained_frozen_random_camera")
    rec["untrained_frozen_fixed_triplet"] = window(tz, fixed, max(50, n_timed // 2), tag="untrained_frozen_fixed_triplet")
    walk_untrained = _walk_statistics(tz, cams[0], bg, W, H)
    del tz
    # (b) the leg itself
    tr = EventTrainer(params, dev)
    t0 = time.perf_counter()
    for _ in range(n_train):
        k = draw()
        tr.step_nocopy(*cams[k], *gts[k], bg, gt_blur=blur[k] if blur else None)
    torch.cuda.synchronize()
    rec["training_phase"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / max(n_train, 1), 3),
                             "count_retries": tr.count_retries, "shared_pose_iterations": tr.shared_pose_iterations}
    rec["trained_random_camera"] = window(tr, draw, n_timed, tag="trained_random_camera")
    rec["pair_count_cache_entries"] = len(tr._pair_counts._entries) if getattr(tr, "_pair_counts", None) is not None else None
    # (c) the end state, frozen, on the headline's triplet and on random cameras
    frozen = _freeze(tr)
    rec["trained_frozen_fixed_triplet"] = window(frozen, fixed, max(50, n_timed // 2), tag="trained_frozen_fixed_triplet")
    rec["trained_frozen_random_camera"] = window(frozen, draw, max(50, n_timed // 2), tag="trained_frozen_random_camera")
    rec["walk_statistics"] = {"untrained": walk_untrained, "trained": _walk_statistics(frozen, cams[0], bg, W, H)}
    if trace_dir:
        os.makedirs(trace_dir, exist_ok=True)
        json.dump(rec["walk_statistics"], open(os.path.join(trace_dir, "walk_statistics.json"), "w"), indent=1)
    rec["trace_markers"] = marks
    a, b = rec["trained_random_camera"], rec["untrained_frozen_fixed_triplet"]
    rec["vs_untrained_fixed_triplet"] = {
        "ms_per_step_ratio": round(a["ms_per_step"] / b["ms_per_step"], 3),
        "instances_ratio": round(a["tile_instances_3views_mean"] / max(b["tile_instances_3views_mean"], 1), 3),
        "what": "trained + random camera against the headline's state (untrained scene, one triplet), both by this function"}
    return rec


def _freeze(tr):
    """The trainer `tr` with every learning rate 0 from here on (Adam keeps running, nothing moves)."""
    for k in list(tr.lrs):
        tr.lrs[k] = 0.0
    tr.c_lr = 0.0
    tr.xyz_lr = lambda step: 0.0
    return tr


def _walk_statistics(tr, trip, bg, W, H):
    """Walk statistics of the compositing kernels on the state `tr` holds, per view of the triplet: tile list lengths,
    entries the walk reaches (largest n_contrib of the tile's pixels), `touched` fraction of the slots, strips evaluated
    per entry with a non-zero strip mask."""
    import numpy as np
    import torch
    from event_3dgs_amd import rasterizer
    res = {}
    for name, cam in zip(("intensity", "now", "next"), trip):
        raw = tr.render_raw(cam, bg)
        st = rasterizer.state_views(raw, tr.N, W, H)
        rg = st["ranges"].long()
        ln = (rg[:, 1] - rg[:, 0]).cpu().numpy().astype(np.float64)
        gx = (W + 15) // 16
        gy = (H + 15) // 16
        nc = st["n_contrib"].view(H, W).long()
        pad = torch.zeros(gy * 16, gx * 16, dtype=torch.long, device=nc.device)
        pad[:H, :W] = nc
        walked = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values.cpu().numpy().astype(np.float64)
        rec = {"instances": int(ln.sum()), "walked_entries": int(walked.sum()),
               "walked_fraction": round(float(walked.sum() / max(ln.sum(), 1)), 4),
               "list_length_percentiles_50_90_99_max": [float(x) for x in np.percentile(ln, [50, 90, 99, 100])],
               "walked_percentiles_50_90_99_max": [float(x) for x in np.percentile(walked, [50, 90, 99, 100])]}
        if "touched" in st:
            rec["touched_fraction_of_slots"] = round(float(st["touched"].float().mean()), 4)
        if "strip_mask" in st:
            # (the forward writes the mask bytes of the entries it walked; the rest of a list holds whatever the scratch held)
            sm = st["strip_mask"].to(torch.int32)
            pos = torch.arange(sm.numel(), device=sm.device)
            tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=sm.device), (rg[:, 1] - rg[:, 0]))
            wt = torch.from_numpy(walked).to(sm.device).long()
            inside = torch.zeros_like(sm, dtype=torch.bool)
            if tile_of.numel() == sm.numel():
                # lists are stored tile-major in range order only when ranges are ascending: use each entry's own range start
                inside = pos < (rg[tile_of, 0] + wt[tile_of])
                order_ok = bool((pos >= rg[tile_of, 0]).all())
                if not order_ok:
                    inside = torch.zeros_like(inside)
            sm = torch.where(inside, sm, torch.zeros_like(sm))
            bits = ((sm & 1) + ((sm >> 1) & 1) + ((sm >> 2) & 1) + ((sm >> 3) & 1)).float()
            nz = sm != 0
            rec["entries_with_a_strip"] = int(nz.sum())
            rec["strips_per_entry_with_a_strip"] = round(float(bits[nz].mean()), 3) if bool(nz.any()) else None
        res[name] = rec
    return res


def measure_kernel_clocks(trainer, cams, bg, L, W, H):
    """Effective shader clock inside render_fwd_kernel / render_bwd_kernel: one 3-view forward + backward with the debug
    trace armed (e3dgs_debug_set_trace: per tile {wall start, wall end at 100 MHz, entries, hw id, s_memtime start,
    s_memtime end}); GHz of a tile = shader cycles / wall nanoseconds, reported as median and 5th / 95th percentile over
    the tiles that ran for at least 20 us."""
    import ctypes as C
    import numpy as np
    import torch
    from event_3dgs_amd import rasterizer
    L.e3dgs_debug_set_trace.argtypes = [C.c_void_p]
    dev = trainer.device
    v = trainer.views
    settings = [trainer._settings(c, bg) for c in cams]
    T = len(cams) * ((W + 15) // 16) * ((H + 15) // 16)
    out = {n: torch.empty_like(t) for n, t in dict(means3D=v["xyz"], sh=v["features"], opacities=v["opacity"],
                                                  scales=v["scaling"], rots=v["rotation"]).items()}
    dpix = torch.randn(len(cams), 3, H, W, device=dev)
    res = {}
    try:
        for which in ("render_fwd_kernel", "render_bwd_kernel"):
            buf = torch.zeros(T * 6, dtype=torch.int64, device=dev)
            for rep in range(2):
                arm = rep == 1
                L.e3dgs_debug_set_trace(buf.data_ptr() if (arm and which == "render_fwd_kernel") else None)
                raw = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                               flags=trainer.FWD_FLAGS)
                torch.cuda.synchronize()
                L.e3dgs_debug_set_trace(buf.data_ptr() if (arm and which == "render_bwd_kernel") else None)
                # (the timed iteration's form: the contrast renders' pixel gradients are rank 1)
                rasterizer.backward_multi(raw, dpix, out, rank1={1: rasterizer.LUV_WEIGHTS, 2: rasterizer.LUV_WEIGHTS}
                                          if (trainer.rank1 and len(cams) == 3) else None)
                torch.cuda.synchronize()
                L.e3dgs_debug_set_trace(None)
            t = buf.cpu().numpy().reshape(T, 6)
            t = t[t[:, 1] > t[:, 0]]
            wall_ns = (t[:, 1] - t[:, 0]).astype(np.float64) * 10.0
            cyc = (t[:, 5] - t[:, 4]).astype(np.float64)
            ok = (wall_ns >= 20000.0) & (cyc > 0)
            if ok.sum() >= 16:
                g = cyc[ok] / wall_ns[ok]
                res[which] = {"ghz_median": round(float(np.median(g)), 3), "ghz_p5": round(float(np.percentile(g, 5)), 3),
                              "ghz_p95": round(float(np.percentile(g, 95)), 3), "tiles": int(ok.sum()),
                              "how": "s_memtime delta / wall_clock64 delta per tile wave, inside the kernel"}
    finally:
        L.e3dgs_debug_set_trace(None)
    return res


def measure_dropin(params, cams, gts, bg, dev, L, iters=8):
    """The iteration a maintainer of the reference gets after swapping the two packages and running train.py UNMODIFIED
    (train.py:144,159,161 three render() calls -> :165-203 loss in torch -> :211 loss.backward() -> :212 optimizer_c.step()
    -> :330-332 optimizer.step()): render() with its forced torch-SH branch (gaussian_renderer/__init__.py:71-81),
    torch activations (scene/gaussian_model.py:95-118), the GaussianRasterizer autograd operator through the compiled
    extension (_C_native), the loss formulas of utils/loss_utils.py in torch, torch.optim.Adam over the six groups
    (scene/gaussian_model.py:154-163) + Adam([c]) -- and the ADOPTION LADDER above it (event_3dgs_amd/adopt.py): the same
    loop with render(), then the loss block, then the two optimizers swapped for this repo's, one at a time, all still
    inside torch autograd / torch.optim.  Reported next to -- never instead of -- the fused iteration."""
    import ctypes as C
    import torch
    from event_3dgs_amd import adopt, rasterizer

    def timed(loop, n):
        for _ in range(2):
            loop.step(cams, gts, bg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loop.step(cams, gts, bg)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n
        host = []
        for _ in range(3):                   # host time of one iteration with the GPU idle at entry (enqueue cost)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loop.step(cams, gts, bg)
            host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        return wall, sorted(host)[1]

    def kernels_ms(loop):
        # rasteriser kernels of one iteration (HIP events around the library's stages; the profiler is process-wide, so the
        # backward calls autograd makes from its own thread are timed too)
        L.e3dgs_profile_enable(0xFF)
        for _ in range(2):
            loop.step(cams, gts, bg)
        torch.cuda.synchronize()
        per = {}
        for slot in range(8):
            ms, n = C.c_double(0), C.c_int(0)
            L.e3dgs_profile_query(slot, C.byref(ms), C.byref(n))
            per[L.e3dgs_profile_slot_name(slot).decode()] = ms.value / 2
        L.e3dgs_profile_enable(0)
        return per

    ladder, loops = {}, {}
    names = {0: "rung0_unmodified", 1: "rung1_render", 2: "rung2_render_loss", 3: "rung3_render_loss_optimizer",
             4: "rung4_one_call_for_the_three_renders"}
    what = {0: "the reference's train.py:144-212,330-332 unmodified on the two drop-in packages",
            1: "+ render -> event_3dgs_amd.adopt.render (SH, activations and their chain rule inside the rasteriser, one C++ "
               "autograd node per render; gaussian_renderer/__init__.py:20-104)",
            2: "+ the loss block train.py:165-203 -> adopt.event_loss (fused event-loss kernels as one autograd node)",
            3: "+ gaussians.optimizer / optimizer_c -> adopt.FusedAdam (torch.optim interface and state layout, one launch "
               "per parameter; scene/gaussian_model.py:154-163, train.py:71-73)",
            4: "+ the three render() calls train.py:144,159,161 -> one adopt.render_views(...) (one multi-view pass of the "
               "rasteriser, forward and backward, in one autograd node)"}
    for rung in (0, 1, 2, 3, 4):
        loop = adopt.LadderLoop(rung, params, dev)
        wall, host = timed(loop, iters if rung == 0 else 2 * iters)
        ker = kernels_ms(loop)
        ladder[names[rung]] = {"ms": round(1e3 * wall, 3), "per_s": round(1.0 / wall, 2), "host_enqueue_ms": round(1e3 * host, 3),
                               "rasteriser_kernels_ms": round(sum(ker.values()), 3),
                               "render_bwd_ms": round(ker.get("render_bwd", 0.0), 3), "geom_bwd_ms": round(ker.get("geom_bwd", 0.0), 3),
                               "what": what[rung]}
        if rung == 0:
            # the same iteration with ONE line of the reference removed (gaussian_renderer/__init__.py:71, which forces the
            # torch SH branch): SH evaluated inside the rasteriser, forward and backward, as upstream 3DGS does
            loop.python_sh = False
            loop.pipe.convert_SHs_python = False
            wall_sh, _ = timed(loop, iters)
        del loop
        torch.cuda.empty_cache()
    r0 = ladder[names[0]]
    return {"ms": r0["ms"], "ms_with_sh_in_the_rasteriser": round(1e3 * wall_sh, 3), "per_s": r0["per_s"],
            "host_enqueue_ms": r0["host_enqueue_ms"], "rasteriser_kernels_ms": r0["rasteriser_kernels_ms"],
            "torch_side_ms": round(r0["ms"] - r0["rasteriser_kernels_ms"], 3),
            "native_extension": rasterizer.native_ext() is not None,
            "cpp_autograd_node": rasterizer.cpp_autograd_ext() is not None,
            "what": "3 x render() (torch SH + activations, GaussianRasterizer via _C_native) + torch loss "
                    "(train.py:165-203) + loss.backward() + torch.optim.Adam (6 groups) + Adam([c])",
            "ladder": ladder}


def source_fingerprint():
    """sha256 over the sources the library is built from: ties a committed profile (profiles/traffic*.json) to the code
    it describes."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "event_3dgs_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in [os.path.join(csrc, x) for x in files] + [os.path.join(ROOT, "include", "e3dgs_hip.h")]:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def self_launch(n):
    """Re-run this command under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port); stdout/stderr are inherited, so the one JSON line of rank 0 is this process's output too."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def _oracle_inputs(trainer):
    """The trained parameters on the host, activated by the ORACLE's statement of the activations (gso_activate:
    scene/gaussian_model.py:33-41 with the deterministic exp) -- what both CPU legs and the parity leg use, and what the
    timed PREACT kernels compute themselves from the raw parameters."""
    import numpy as np
    from oracle import c_oracle
    v = {k: t.detach().cpu().numpy() for k, t in trainer.views.items()}
    scales, rots, opac = c_oracle.activate(v["scaling"], v["rotation"], v["opacity"])
    return dict(means=v["xyz"], scales=scales, rots=rots, opac=opac, raw_rots=v["rotation"],
                shs=np.ascontiguousarray(v["features"].T.reshape(-1, 16, 3)))   # (48,N) planar -> (N,16,3)


def _pick_threads():
    """Thread count for the PyTorch CPU leg: all host cores unless a smaller team is faster on the op mix of the per-tile
    compositing (hundreds of small elementwise / cumprod ops, where a 256-thread barrier costs more than it saves)."""
    import torch
    ncpu = os.cpu_count() or 1
    best, best_t = ncpu, None
    x = torch.rand(1200, 256)
    for n in sorted({ncpu, min(ncpu, 64), min(ncpu, 16)}, reverse=True):
        torch.set_num_threads(n)
        for _ in range(2):
            torch.cumprod(1.0 - 0.01 * torch.exp(-x * x), 0)
        t0 = time.perf_counter()
        for _ in range(10):
            torch.cumprod(1.0 - 0.01 * torch.exp(-x * x), 0)
        t = time.perf_counter() - t0
        if best_t is None or t < 0.9 * best_t:
            best, best_t = n, t
    torch.set_num_threads(best)
    return best


def run_torch_cpu_baseline(inp, cams, bg, W, H, rows, budget_s=90.0):
    """The baseline north_star names: the pure-PyTorch rasteriser (oracle/torch_oracle.py: vectorised per tile, gradients
    from autograd) on the host cores.  Sample: projection of ALL Gaussians, binning + compositing forward and backward
    on `rows` tile rows around the image centre, per-Gaussian backward of all Gaussians; the binning / compositing
    times are extrapolated linearly in tile instances to the whole frame (rectangle binning, as the reference)."""
    import numpy as np
    import torch
    from oracle import torch_oracle
    threads = _pick_threads()
    gy = (H + 15) // 16
    r0 = max(0, gy // 2 - rows // 2)
    total, detail = 0.0, []
    # untimed warm-up (thread pool start, first-touch of the allocator arenas, autograd graph caches): the first timed view
    # used to carry ~25 % of one-off cost
    sub = slice(0, min(100_000, inp["means"].shape[0]))
    wl = [torch.from_numpy(inp[k][sub]).clone().requires_grad_(True) for k in ("means", "opac", "shs", "scales", "rots")]
    wimg, _, _ = torch_oracle.rasterize(
        wl[0], wl[1], viewmatrix=cams[0].world_view_transform.cpu(), projmatrix=cams[0].full_proj_transform.cpu(),
        campos=cams[0].camera_center.cpu(), bg=bg.cpu(), width=W, height=H, tanfovx=math.tan(cams[0].FoVx * 0.5),
        tanfovy=math.tan(cams[0].FoVy * 0.5), shs=wl[2], sh_degree=3, scales=wl[3], rotations=wl[4],
        tile_rows=(r0, r0 + 1), return_aux=True)
    wimg.sum().backward()
    del wl, wimg
    t_begin = time.perf_counter()
    for cam in cams:
        if detail and time.perf_counter() - t_begin > budget_s:       # slow host: remaining views cost what the mean did
            total += sum(d["est_s"] for d in detail) / len(detail)
            detail.append({"skipped": True})
            continue
        leaves = [torch.from_numpy(inp[k]).clone().requires_grad_(True) for k in ("means", "opac", "shs", "scales", "rots")]
        tm = {}
        img, radii, aux = torch_oracle.rasterize(
            leaves[0], leaves[1], viewmatrix=cam.world_view_transform.cpu(), projmatrix=cam.full_proj_transform.cpu(),
            campos=cam.camera_center.cpu(), bg=bg.cpu(), width=W, height=H, tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), shs=leaves[2], sh_degree=3, scales=leaves[3], rotations=leaves[4],
            tile_rows=(r0, r0 + rows), timings=tm, return_aux=True)
        mids = aux["diff"]
        t1 = time.perf_counter()
        g_mid = torch.autograd.grad(img.sum(), mids, retain_graph=True, allow_unused=True)      # compositing backward
        t2 = time.perf_counter()
        keep = [(m, g) for m, g in zip(mids, g_mid) if g is not None and m.requires_grad]
        torch.autograd.backward([m for m, _ in keep], [g for _, g in keep])                    # per-Gaussian backward
        t3 = time.perf_counter()
        pre, binning = tm["preprocess_done"] - tm["start"], tm["binning_done"] - tm["preprocess_done"]
        comp = tm["composite_done"] - tm["binning_done"]
        rect = aux["rect"].numpy()
        full_I = int(aux["tiles_touched"].sum())
        scale = full_I / max(aux["num_rendered"], 1)
        est = pre + (binning + comp + (t2 - t1)) * scale + (t3 - t2)
        total += est
        detail.append({"pre_s": round(pre, 3), "bin_s": round(binning, 3), "comp_fwd_s": round(comp, 3),
                       "comp_bwd_s": round(t2 - t1, 3), "geom_bwd_s": round(t3 - t2, 3),
                       "row_instances": int(aux["num_rendered"]), "frame_instances": full_I, "extrap": round(scale, 2),
                       "est_s": round(est, 2)})
        del img, aux, leaves, g_mid, keep, rect
    return {"value": round(1.0 / total, 6), "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": f"pure-PyTorch CPU rasteriser (oracle/torch_oracle.py, autograd, {threads} threads of {os.cpu_count()} host "
                      f"CPUs): 3 views, projection + per-Gaussian backward of all Gaussians, binning + compositing fwd+bwd on "
                      f"{rows} of {gy} tile rows extrapolated by tile instances; loss/Adam excluded",
            "host_cpus": os.cpu_count(), "detail": detail}


def run_c_oracle_baseline(trainer, inp, cams, bg, W, H, rows):
    """C oracle (single thread) on a bounded sample of the same workload: full preprocess + binning of
    all Gaussians for the three views, compositing fwd+bwd restricted to `rows` tile rows around the image
    centre, extrapolated linearly in tile instances to the whole image.  The same leg checks THE PATH THAT WAS TIMED --
    one multi-view pass on the raw parameters with in-kernel activations and coefficient-major SH
    (rasterizer.forward_multi / backward_multi with EventTrainer.FWD_FLAGS) -- against the oracle on the sampled rows:
    radii, image (bit for bit expected) and, per Gaussian, the gradients w.r.t. the raw parameters of a pixel gradient
    that is non-zero on those rows only."""
    import numpy as np
    import torch
    from event_3dgs_amd import rasterizer
    from oracle import c_oracle
    from oracle.metrics import rel_l2
    means, scales, rots, opac, shs = (inp[k] for k in ("means", "scales", "rots", "opac", "shs"))
    dev = trainer.device
    P = means.shape[0]
    gy = (H + 15) // 16
    r0 = max(0, gy // 2 - rows // 2)
    y0, y1 = r0 * 16, min(H, (r0 + rows) * 16)
    total = 0.0
    detail, diffs, radii_bad = [], [], []
    gw = np.zeros((len(cams), 3, H, W), np.float32)
    rng = np.random.default_rng(7)
    for k in range(len(cams)):
        gw[k, :, y0:y1] = rng.standard_normal((3, y1 - y0, W)).astype(np.float32)
    # ---- the timed path: ONE multi-view pass on the raw parameters
    v = trainer.views
    settings = [trainer._settings(c, bg) for c in cams]
    hip = rasterizer.forward_multi(v["xyz"], v["features"], v["opacity"], v["scaling"], v["rotation"], settings,
                                   flags=trainer.FWD_FLAGS)
    acc = {k: 0.0 for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    for k, cam in enumerate(cams):
        f = c_oracle.Forward(means3D=means, opacities=opac, viewmatrix=cam.world_view_transform.contiguous().cpu().numpy(),
                             projmatrix=cam.full_proj_transform.cpu().numpy(),
                             campos=cam.camera_center.contiguous().cpu().numpy(), bg=bg.cpu().numpy(), width=W,
                             height=H, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), shs=shs,
                             sh_degree=3, scales=scales, rotations=rots, tile_rows=(r0, r0 + rows))
        t1 = time.perf_counter()
        gb = f.backward(gw[k])                   # timed alone: windowed compositing backward + per-Gaussian backward
        t2 = time.perf_counter()
        diffs.append(float(np.abs(hip["color"][k][:, y0:y1].cpu().numpy() - f.out_color[:, y0:y1]).max()))
        radii_bad.append(int((hip["radii"][k].cpu().numpy() != f.radii).sum()))
        for name in acc:
            acc[name] = acc[name] + np.asarray(gb[name], np.float64)
        tm = f.timings            # preprocess, binning, composite seconds
        ranges = f.ranges.reshape(gy, -1, 2)
        inst_rows = int((ranges[r0:r0 + rows, :, 1] - ranges[r0:r0 + rows, :, 0]).sum())
        scale = f.num_rendered / max(inst_rows, 1)
        est = tm[0] + tm[1] + (tm[2] + (t2 - t1)) * scale
        total += est
        detail.append({"pre_s": round(tm[0], 3), "bin_s": round(tm[1], 3), "comp_fwd_s": round(tm[2], 3),
                       "bwd_s": round(t2 - t1, 3), "row_instances": inst_rows, "extrap": round(scale, 2)})
        f.close()
    # oracle gradients w.r.t. the RAW parameters (chain rule of gso_activate), HIP gradients of the multi-view backward
    gs, gq, go = c_oracle.activate_backward(inp["raw_rots"], scales, rots, opac, acc["scales"], acc["rotations"],
                                            acc["opacities"])
    ref = {"xyz": acc["means3D"], "scaling": gs, "rotation": gq, "opacity": go,
           "features": np.ascontiguousarray(np.asarray(acc["shs"]).reshape(P, 48).T)}
    e = lambda like: torch.full_like(like, float("nan"))
    out = dict(means3D=e(v["xyz"]), sh=e(v["features"]), opacities=e(v["opacity"]), scales=e(v["scaling"]),
               rots=e(v["rotation"]))
    rasterizer.backward_multi(hip, torch.from_numpy(gw).to(dev), out)
    got = {"xyz": out["means3D"], "scaling": out["scales"], "rotation": out["rots"], "opacity": out["opacities"],
           "features": out["sh"]}
    gerr = {}
    for name, t in got.items():
        a = t.cpu().numpy().astype(np.float64)
        b = np.asarray(ref[name], np.float64).reshape(a.shape)
        if name == "features":
            a, b = a.T, b.T
        a2, b2 = a.reshape(P, -1), b.reshape(P, -1)
        per = np.abs(a2 - b2).max(axis=1) / (np.abs(b2).max(axis=1) + 1e-3 * np.abs(b2).max())
        gerr[name] = (rel_l2(a2, b2), float(per.max()))
    del hip, out
    c_part = {"value": round(1.0 / total, 5), "unit": "iters/s", "cores": 1, "kind": "port",
              "sample": f"C oracle (oracle/gs_oracle.c, 1 thread): 3 views, full preprocess+sort of all Gaussians, "
                        f"compositing fwd+bwd on {rows} of {gy} tile rows extrapolated by tile instances; loss/Adam excluded",
              "detail": detail}
    parity = {"path": "forward_multi/backward_multi, E3DGS_FLAG_PREACT | E3DGS_FLAG_SH_PLANAR (the timed path)",
              "rows_checked": rows * 16 * len(cams), "max_abs_diff": max(diffs), "radii_mismatch": sum(radii_bad),
              "psnr_db": None if max(diffs) == 0.0 else round(-20.0 * math.log10(max(diffs)), 1),
              "grad_rel_l2": {k: float("%.3g" % v[0]) for k, v in gerr.items()},
              "grad_per_gaussian_max": {k: float("%.3g" % v[1]) for k, v in gerr.items()},
              "note": "the multi-view PREACT path bench.py timed vs the C oracle fed gso_activate() of the same raw parameters "
                      "(the trained parameters after the timed steps), sampled tile rows of the three views; image: 0.0 = "
                      "bit-identical (PSNR unbounded); gradients w.r.t. the raw parameters of a random pixel gradient supported "
                      "on those rows: global relative L2 and max over Gaussians of |d_i| / (|ref_i| + 1e-3 max_j |ref_j|)"}
    return c_part, parity


def run_cpu_baseline(trainer, cams, bg, W, H, rows, torch_rows):
    inp = _oracle_inputs(trainer)
    c_part, parity = run_c_oracle_baseline(trainer, inp, cams, bg, W, H, rows)
    out = run_torch_cpu_baseline(inp, cams, bg, W, H, torch_rows)
    out["c_oracle_single_thread"] = c_part
    out["parity_vs_oracle"] = parity
    return out


if __name__ == "__main__":
    main()
