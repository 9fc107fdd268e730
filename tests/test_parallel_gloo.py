"""View-parallel data parallelism on CPU: world_size 2, gloo.  Each rank renders its own camera with the
PyTorch oracle (CPU), gradients are averaged with event_3dgs_amd.parallel.allreduce_mean_ exactly as
EventTrainer.step does on the GPU, and the result must equal the single-process mean over both views."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import scene


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _view_grad(act, cam_k, W, H):
    import math
    from event_3dgs_amd.cameras import orbit_camera
    from oracle import torch_oracle
    cam = orbit_camera(cam_k, 8, W, H)
    leaves = {k: act[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors")}
    img, _ = torch_oracle.rasterize(leaves["means3D"], leaves["opacities"], viewmatrix=cam.world_view_transform,
                                    projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                                    bg=torch.zeros(3), width=W, height=H, tanfovx=math.tan(cam.FoVx / 2),
                                    tanfovy=math.tan(cam.FoVy / 2), colors_precomp=leaves["colors"],
                                    scales=leaves["scales"], rotations=leaves["rotations"])
    (img ** 2).mean().backward()
    return torch.cat([leaves[k].grad.reshape(-1) for k in sorted(leaves)])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_3dgs_amd import parallel
    torch.set_num_threads(2)
    act, _ = scene(300, 64, 48, seed=4)
    flat = _view_grad(act, rank, 64, 48)          # every rank: its own view of the replicated model
    # the trainer's pipelined form: contiguous chunks, all collectives issued first, waited for one by one
    chunked = flat.clone()
    cuts = [0, 7, 100, 1000, chunked.numel()]
    pend = [parallel.allreduce_mean_async_(chunked[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for p in pend:
        p.wait()
    parallel.allreduce_mean_(flat)
    assert torch.equal(chunked, flat)
    # the all-gather used for the per-view colour gradients: row r of `out` = rank r's block
    local = torch.full((5,), float(rank + 1))
    gathered = parallel.allgather_async_(torch.zeros(world, 5), local).wait()
    assert torch.equal(gathered, torch.tensor([[1.0] * 5, [2.0] * 5]))
    # the direct schedule: reduce-scatter of the mean + all-gather of the shards == the all-reduced mean
    n = flat.numel()
    sh = (n + world - 1) // world
    staging = torch.zeros(world * sh)
    staging[:n] = _view_grad(act, rank, 64, 48)
    shard = parallel.reduce_scatter_mean_async_(torch.empty(sh), staging).wait()
    full = parallel.allgather_flat_async_(torch.empty(world * sh), shard).wait()
    assert torch.equal(full[:n], flat)
    idx = [parallel.rank_camera_indices(r, world, 100, iteration=7) for r in range(world)]
    if rank == 0:
        torch.save({"flat": flat, "idx": idx}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process_mean(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    act, _ = scene(300, 64, 48, seed=4)
    ref = 0.5 * (_view_grad(act, 0, 64, 48) + _view_grad(act, 1, 64, 48))
    assert float(ref.abs().max()) > 0
    assert np.allclose(got["flat"].numpy(), ref.numpy(), rtol=1e-6, atol=1e-9)
    # camera draws: deterministic, in range, held-out views avoided, reproducible on any rank
    from event_3dgs_amd import parallel
    assert got["idx"] == [parallel.rank_camera_indices(r, 2, 100, iteration=7) for r in range(2)]
    assert all(2 <= i <= 96 and i not in (5, 25, 45, 65, 85) for i in got["idx"])


def test_allreduce_is_identity_without_process_group():
    from event_3dgs_amd import parallel
    t = torch.arange(5.0)
    assert torch.equal(parallel.allreduce_mean_(t.clone()), t)


def test_comm_chunks_tile_the_flat_buffer():
    """EventTrainer._comm_chunks: the six collectives cover every element of the flat gradient buffer once, in order,
    and the f_dc / f_rest learning-rate boundary falls inside the first feature chunk."""
    import types
    from event_3dgs_amd.train_step import EventTrainer, SEGMENTS
    for N in (1, 255, 1000, 1_000_000):
        seg, off = {}, 0
        for name, per in SEGMENTS:
            seg[name] = (off, N * per); off += N * per
        seg["c"] = (off, 1)
        fake = types.SimpleNamespace(N=N, seg=seg, flat=torch.empty(0).new_empty(off + 1), FEATURE_CHUNKS=4)
        chunks = EventTrainer._comm_chunks(fake)
        pos = 0
        for kind, o, n in chunks:
            assert o == pos and n > 0
            pos += n
        assert pos == off + 1
        feats = [c for c in chunks if c[0] == "features"]
        assert feats[0][1] == seg["features"][0] and sum(c[2] for c in feats) == 48 * N
        assert feats[0][2] >= min(3 * N, 48 * N)          # all f_dc rows sit in the first feature chunk


def test_sh_exchange_is_chosen_by_link_bytes():
    """Factorised (all-gather of 9 floats per Gaussian and rank) against a ring all-reduce of the 48 SH-gradient floats:
    252 vs 336 MB per rank at 8 ranks and 1 M Gaussians; the all-reduce moves fewer bytes from 11 ranks on."""
    from event_3dgs_amd import parallel
    b = parallel.sh_exchange_bytes(8, 1_000_000)
    assert b["factorised_link_bytes"] == 7 * 36_000_000 and b["allreduce_link_bytes"] == 336_000_000
    assert [parallel.choose_sh_exchange(w) for w in (1, 2, 8, 10, 11, 16)] == \
        ["allreduce", "factorised", "factorised", "factorised", "allreduce", "allreduce"]
