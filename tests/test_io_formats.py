"""On-disk formats: COLMAP readers against the reference's own loader (golden G9), model PLY
layout, point-cloud PLY, checkpoint tuple."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from event_3dgs_amd import io_formats as IO

CD = os.path.join(GOLDEN, "colmap_tiny")


def test_g9_colmap_readers_match_reference_loader():
    g = np.load(os.path.join(GOLDEN, "colmap_tiny.npz"))
    for tag, cams, imgs, pts in (("bin", IO.read_cameras_binary(os.path.join(CD, "cameras.bin")),
                                  IO.read_images_binary(os.path.join(CD, "images.bin")),
                                  IO.read_points3D_binary(os.path.join(CD, "points3D.bin"))),
                                 ("txt", IO.read_cameras_text(os.path.join(CD, "cameras.txt")),
                                  IO.read_images_text(os.path.join(CD, "images.txt")),
                                  IO.read_points3D_text(os.path.join(CD, "points3D.txt")))):
        assert len(cams) == 2 and len(imgs) == 5
        for k, c in cams.items():
            assert np.array_equal(np.array([c.id, c.width, c.height] + list(c.params), np.float64), g[f"cam_{tag}_{k}"])
            assert c.model == str(g[f"cammodel_{tag}_{k}"])
        for k, im in imgs.items():
            assert np.array_equal(np.concatenate([[im.id], im.qvec, im.tvec, [im.camera_id]]), g[f"img_{tag}_{k}"])
            assert im.name == str(g[f"imgname_{tag}_{k}"])
            assert np.array_equal(im.xys, g[f"imgxys_{tag}_{k}"]) and np.array_equal(im.point3D_ids, g[f"imgpids_{tag}_{k}"])
        assert np.array_equal(pts[0], g[f"xyz_{tag}"]) and np.array_equal(pts[1], g[f"rgb_{tag}"])
        assert np.array_equal(pts[2], g[f"err_{tag}"])
    views = IO.colmap_cameras_to_views(IO.read_images_binary(os.path.join(CD, "images.bin")),
                                       IO.read_cameras_binary(os.path.join(CD, "cameras.bin")))
    assert [v["image_name"] for v in views] == sorted(v["image_name"] for v in views)      # dataset_readers.py:148
    imgs = IO.read_images_binary(os.path.join(CD, "images.bin"))
    by_name = {os.path.basename(im.name).split(".")[0]: k for k, im in imgs.items()}
    for v in views:
        ref = g[f"view_{by_name[v['image_name']]}"]
        assert np.allclose(np.concatenate([v["R"].reshape(-1), v["T"], [v["FovX"], v["FovY"]]]), ref, rtol=0, atol=1e-12)
    tr, rad = IO.nerf_normalization(views)
    assert np.allclose(tr, g["nerfpp_translate"], atol=1e-6) and abs(rad - float(g["nerfpp_radius"])) <= 1e-6


def test_model_ply_layout_and_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(0)
    N = 37
    p = dict(xyz=torch.randn(N, 3, generator=g), features_dc=torch.randn(N, 1, 3, generator=g),
             features_rest=torch.randn(N, 15, 3, generator=g), opacity=torch.randn(N, 1, generator=g),
             scaling=torch.randn(N, 3, generator=g), rotation=torch.randn(N, 4, generator=g))
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    IO.save_model_ply(path, **p)
    raw = open(path, "rb").read()
    header = raw[:raw.index(b"end_header\n") + 11].decode()
    lines = header.strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {N}"]
    names = [l.split()[2] for l in lines[3:-1]]
    assert names == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)]
                     + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    assert all(l.split()[1] == "float" for l in lines[3:-1]) and len(raw) == len(header) + N * 62 * 4
    body = np.frombuffer(raw[len(header):], "<f4").reshape(N, 62)
    # f_rest is channel-major on disk: (N,15,3) -> transpose -> (N,3,15) -> flatten (gaussian_model.py:197)
    assert np.array_equal(body[:, 9:54], p["features_rest"].transpose(1, 2).flatten(1).numpy())
    assert np.array_equal(body[:, 3:6], np.zeros((N, 3), np.float32)) and np.array_equal(body[:, 54], p["opacity"][:, 0].numpy())
    q = IO.load_model_ply(path)
    for k in p:
        assert torch.equal(q[k], p[k]), k
    assert q["active_sh_degree"] == 3


def test_pointcloud_ply_roundtrip(tmp_path):
    rs = np.random.RandomState(0)
    xyz, rgb = rs.randn(20, 3).astype(np.float32), rs.randint(0, 256, (20, 3))
    path = str(tmp_path / "points3D.ply")
    IO.store_pointcloud_ply(path, xyz, rgb)
    pc = IO.fetch_pointcloud_ply(path)
    assert np.array_equal(pc.points, xyz) and np.allclose(pc.colors, rgb / 255.0) and np.all(pc.normals == 0)


def test_checkpoint_tuple_is_loadable_by_torch_adam(tmp_path):
    """capture -> torch.save -> torch.load -> restore; the optimizer state_dict loads into a torch Adam built
    like training_setup (scene/gaussian_model.py:154-163)."""
    from event_3dgs_amd.densify import DensifyStats
    g = torch.Generator().manual_seed(1)
    N = 11
    shapes = {"xyz": (N, 3), "f_dc": (N, 1, 3), "f_rest": (N, 15, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}
    groups = {k: [torch.randn(s, generator=g), torch.randn(s, generator=g), torch.rand(s, generator=g)] for k, s in shapes.items()}
    stats = DensifyStats(N, "cpu")
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    tup = IO.capture_checkpoint(groups, stats, 2, 1.5, lrs, step=40)
    path = str(tmp_path / "chkpnt40.pth")
    torch.save((tup, 40), path)                                   # train.py:334-336
    (model_args, it) = torch.load(path, weights_only=False)
    assert it == 40 and len(model_args) == 12 and model_args[0] == 2 and model_args[-1] == 1.5
    params = [model_args[1], model_args[2], model_args[3], model_args[6], model_args[4], model_args[5]]
    opt = torch.optim.Adam([{"params": [p], "lr": lrs[n], "name": n} for p, n in zip(params, lrs)], lr=0.0, eps=1e-15)
    opt.load_state_dict(model_args[10])                           # what GaussianModel.restore does (:93)
    assert torch.equal(opt.state[params[0]]["exp_avg"], groups["xyz"][1])
    groups2, st2, deg, scale = IO.restore_checkpoint(model_args)
    for k in groups:
        for j in range(3):
            assert torch.equal(groups2[k][j], groups[k][j]), (k, j)


def test_checkpoint_keeps_the_per_group_adam_step_counts():
    """The trainer's step dict ({"gauss", "opacity", "c"}: the opacity group lags after resets) goes into the optimizer
    state per parameter as torch keeps it, and comes back out: a resumed run continues the bias correction."""
    from event_3dgs_amd.densify import DensifyStats
    N = 5
    shapes = {"xyz": (N, 3), "f_dc": (N, 1, 3), "f_rest": (N, 15, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}
    groups = {k: [torch.zeros(s), torch.ones(s), torch.ones(s)] for k, s in shapes.items()}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    tup = IO.capture_checkpoint(groups, DensifyStats(N, "cpu"), 3, 1.0, lrs, step={"gauss": 120, "opacity": 20, "c": 130})
    state = tup[10]["state"]
    assert [int(state[i]["step"]) for i in range(6)] == [120, 120, 120, 20, 120, 120]       # order: xyz f_dc f_rest opacity ...
    assert IO.restored_steps(tup) == {"gauss": 120, "opacity": 20}
    # the per-name form still works
    tup = IO.capture_checkpoint(groups, DensifyStats(N, "cpu"), 3, 1.0, lrs, step={k: 7 for k in shapes})
    assert IO.restored_steps(tup) == {"gauss": 7, "opacity": 7}


def _make_dataset(root):
    """A tiny scene directory in the reference's layout, on top of the golden COLMAP model."""
    import shutil
    from PIL import Image
    os.makedirs(os.path.join(root, "sparse/0"))
    for f in ("cameras.bin", "images.bin", "points3D.bin"):
        shutil.copy(os.path.join(CD, f), os.path.join(root, "sparse/0", f))
    rs = np.random.RandomState(0)
    imgs = IO.read_images_binary(os.path.join(CD, "images.bin"))
    for d in ("images", "images_event", "images_blurry", "renders"):
        os.makedirs(os.path.join(root, d))
        for im in imgs.values():
            w, h = (1920, 1080) if d == "images" else (64, 48)
            Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, d, im.name))
    return imgs


def test_scene_directory_loader(tmp_path):
    from event_3dgs_amd import scene_io
    root = str(tmp_path / "scene")
    imgs = _make_dataset(root)
    sc = scene_io.load_colmap_scene(root, gray=True, event=True, deblur=False)
    names = [c.image_name for c in sc.train_cameras]
    assert names == sorted(os.path.basename(i.name).split(".")[0] for i in imgs.values())
    assert len(sc.event_cameras) == 5 and len(sc.test_cameras) == 5 and sc.blurry_cameras == []
    # -r -1 caps the width at 1600 (utils/camera_utils.py:26-36): 1920x1080 -> 1600x900
    assert (sc.train_cameras[0].image_width, sc.train_cameras[0].image_height) == (1600, 900)
    assert sc.event_cameras[0].original_image.shape == (3, 48, 64)
    # the event cameras are read with the training cameras' extrinsics / intrinsics (scene/dataset_readers.py:157): the
    # event camera `index` is the training camera `index` seen again -- what EventTrainer's shared-pose iteration uses
    for tc, ec in zip(sc.train_cameras, sc.event_cameras):
        assert tc is not ec and tc.image_name == ec.image_name
        assert torch.equal(tc.world_view_transform, ec.world_view_transform) and torch.equal(tc.camera_center, ec.camera_center)
        assert (tc.FoVx, tc.FoVy) == (ec.FoVx, ec.FoVy)
    assert float(sc.event_cameras[0].original_image.max()) <= 1.0
    assert os.path.exists(sc.ply_path) and np.all(sc.point_cloud.colors == 0.5)          # --gray initial colours
    assert scene_io.target_resolution(1920, 1080, 1) == (1920, 1080)
    assert scene_io.target_resolution(1920, 1080, 2) == (960, 540)
    assert scene_io.target_resolution(1920, 1080, 800) == (800, 450)
    g = np.load(os.path.join(GOLDEN, "colmap_tiny.npz"))
    assert abs(sc.cameras_extent - float(g["nerfpp_radius"])) <= 1e-6
    sc2 = scene_io.load_colmap_scene(root, deblur=True)
    assert len(sc2.blurry_cameras) == 5 and not np.all(sc2.point_cloud.colors == 0.5)
