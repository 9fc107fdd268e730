"""On-disk formats touched by the path's callers: model PLY, input point-cloud PLY,
training checkpoint tuple, COLMAP binary/text models.  Pure host code (numpy); `plyfile` is not needed.

  model PLY         scene/gaussian_model.py:177-256   x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_* (all f4,
                                                      PRE-activation values, f_rest channel-major)
  point-cloud PLY   scene/dataset_readers.py:109-132  x,y,z,nx,ny,nz f4 + red,green,blue u1
  checkpoint        scene/gaussian_model.py:61-93, train.py:334-336
  COLMAP            scene/colmap_loader.py:83-294
"""
import collections
import os
import struct

import numpy as np
import torch

_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
              "int32": "i4", "uint": "u4", "uint32": "u4"}
_PLY_NAMES = {"f4": "float", "f8": "double", "u1": "uchar", "i4": "int", "u4": "uint", "i2": "short", "u2": "ushort",
              "i1": "char"}


def write_ply_vertices(path, names_dtypes, columns):
    """Binary little-endian PLY with one `vertex` element (what plyfile's PlyData([el]).write produces)."""
    dt = np.dtype([(n, "<" + t) for n, t in names_dtypes])
    n = len(columns[0]) if columns else 0
    arr = np.empty(n, dtype=dt)
    for (name, _), col in zip(names_dtypes, columns):
        arr[name] = col
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
        hdr += [f"property {_PLY_NAMES[t]} {name}" for name, t in names_dtypes]
        hdr.append("end_header")
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(arr.tobytes())


def read_ply_vertices(path):
    """Returns a numpy structured array of the `vertex` element (binary LE/BE or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, props, count, in_vertex = None, [], 0, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            arr = np.empty(count, dtype=[(n, t) for n, t in props])
            for i, (n, _) in enumerate(props):
                arr[n] = data[:, i]
            return arr
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def model_attribute_names(n_rest=45):
    """construct_list_of_attributes, scene/gaussian_model.py:177-189"""
    l = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    return l + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def save_model_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """scene/gaussian_model.py:191-208.  Inputs in the reference layout: (N,3) (N,1,3) (N,15,3) (N,1) (N,3) (N,4)."""
    t = lambda x: x.detach().cpu()
    xyz_n = t(xyz).numpy()
    f_dc = t(features_dc).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = t(features_rest).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    attrs = np.concatenate((xyz_n, np.zeros_like(xyz_n), f_dc, f_rest, t(opacity).numpy(), t(scaling).numpy(),
                            t(rotation).numpy()), axis=1)
    names = model_attribute_names(f_rest.shape[1])
    write_ply_vertices(path, [(n, "f4") for n in names], [attrs[:, i] for i in range(attrs.shape[1])])


def load_model_ply(path, max_sh_degree=3, device="cpu"):
    """scene/gaussian_model.py:215-256 -> dict in the layout of synth.make_scene / EventTrainer."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    key = lambda x: int(x.split("_")[-1])
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]                  # (N,3,1)
    extra = sorted([n for n in names if n.startswith("f_rest_")], key=key)
    assert len(extra) == 3 * (max_sh_degree + 1) ** 2 - 3
    f_rest = np.stack([v[n] for n in extra], axis=1).reshape(xyz.shape[0], 3, (max_sh_degree + 1) ** 2 - 1)
    scales = np.stack([v[n] for n in sorted([n for n in names if n.startswith("scale_")], key=key)], axis=1)
    rots = np.stack([v[n] for n in sorted([n for n in names if n.startswith("rot")], key=key)], axis=1)
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return dict(xyz=T(xyz), features_dc=T(f_dc).transpose(1, 2).contiguous(), features_rest=T(f_rest).transpose(1, 2).contiguous(),
                opacity=T(v["opacity"][:, None]), scaling=T(scales), rotation=T(rots), active_sh_degree=max_sh_degree)


BasicPointCloud = collections.namedtuple("BasicPointCloud", ["points", "colors", "normals"])


def store_pointcloud_ply(path, xyz, rgb):
    """storePly, scene/dataset_readers.py:117-132 (rgb 0..255)."""
    xyz = np.asarray(xyz, np.float32)
    cols = [xyz[:, 0], xyz[:, 1], xyz[:, 2]] + [np.zeros(len(xyz), np.float32)] * 3 + [np.asarray(rgb)[:, i].astype(np.uint8) for i in range(3)]
    write_ply_vertices(path, [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                              ("red", "u1"), ("green", "u1"), ("blue", "u1")], cols)


def fetch_pointcloud_ply(path):
    """fetchPly, scene/dataset_readers.py:109-115"""
    v = read_ply_vertices(path)
    return BasicPointCloud(points=np.vstack([v["x"], v["y"], v["z"]]).T,
                           colors=np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0,
                           normals=np.vstack([v["nx"], v["ny"], v["nz"]]).T)


# ------------------------------------------------------------------------------------------------ checkpoint
def capture_checkpoint(groups, stats, active_sh_degree, spatial_lr_scale, lrs, step, eps=1e-15):
    """The 12-tuple of GaussianModel.capture() (scene/gaussian_model.py:61-75) incl. a torch.optim.Adam
    state_dict with the reference's six groups in its order (:154-163), so `torch.save((tuple, iteration), path)`
    (train.py:334-336) yields a file the reference's restore() accepts.  `step`: one count, or a dict group -> count
    (EventTrainer.steps: the opacity group lags after resets)."""
    order = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    state, pg = {}, []
    if isinstance(step, dict) and "gauss" in step:
        # EventTrainer.steps: {"gauss", "opacity", "c"} -- the five non-opacity groups share one count
        step = {name: int(step["opacity" if name == "opacity" else "gauss"]) for name in order}
    for i, name in enumerate(order):
        st = step[name] if isinstance(step, dict) else step     # torch keeps one step count per parameter
        state[i] = {"step": torch.tensor(float(st)), "exp_avg": groups[name][1].clone(), "exp_avg_sq": groups[name][2].clone()}
        pg.append({"lr": lrs[name], "name": name, "betas": (0.9, 0.999), "eps": eps, "weight_decay": 0, "amsgrad": False,
                   "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                   "params": [i]})
    P = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
    return (active_sh_degree, P(groups["xyz"][0]), P(groups["f_dc"][0]), P(groups["f_rest"][0]), P(groups["scaling"][0]),
            P(groups["rotation"][0]), P(groups["opacity"][0]), stats.max_radii2D.clone(), stats.xyz_gradient_accum.clone(),
            stats.denom.clone(), {"state": state, "param_groups": pg}, spatial_lr_scale)


def restore_checkpoint(model_args):
    """Inverse of capture_checkpoint / GaussianModel.restore (:77-93) -> (groups, stats dict, sh degree, lr scale).
    The per-group Adam step counts of the optimizer state are in `restored_steps(model_args)`: a trainer resumed with
    non-zero moments must continue their bias correction (EventTrainer.import_groups(groups, steps=...))."""
    (deg, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, accum, denom, opt, lr_scale) = model_args
    params = {"xyz": xyz, "f_dc": f_dc, "f_rest": f_rest, "opacity": opacity, "scaling": scaling, "rotation": rotation}
    by_name = {g["name"]: g["params"][0] for g in opt["param_groups"]}
    groups = {}
    for name, p in params.items():
        st = opt["state"].get(by_name[name])
        m = st["exp_avg"] if st else torch.zeros_like(p)
        v = st["exp_avg_sq"] if st else torch.zeros_like(p)
        groups[name] = [p.detach().clone(), m.clone(), v.clone()]
    return groups, dict(max_radii2D=max_radii2D, xyz_gradient_accum=accum, denom=denom), deg, lr_scale


def restored_steps(model_args):
    """Adam step counts of a checkpoint tuple in EventTrainer.steps form: {"gauss": n, "opacity": n} (the five
    non-opacity groups of the reference share one count; "c" belongs to train.py's separate optimizer_c and is not
    part of GaussianModel.capture())."""
    opt = model_args[10]
    by_name = {g["name"]: g["params"][0] for g in opt["param_groups"]}
    def cnt(name):
        st = opt["state"].get(by_name[name])
        return int(float(st["step"])) if st and "step" in st else 0
    return {"gauss": max(cnt(n) for n in ("xyz", "f_dc", "f_rest", "scaling", "rotation")), "opacity": cnt("opacity")}


# ------------------------------------------------------------------------------------------------ COLMAP
CameraModel = collections.namedtuple("CameraModel", ["model_id", "model_name", "num_params"])
ColmapCamera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
ColmapImage = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
CAMERA_MODELS = {m.model_id: m for m in (
    CameraModel(0, "SIMPLE_PINHOLE", 3), CameraModel(1, "PINHOLE", 4), CameraModel(2, "SIMPLE_RADIAL", 4),
    CameraModel(3, "RADIAL", 5), CameraModel(4, "OPENCV", 8), CameraModel(5, "OPENCV_FISHEYE", 8),
    CameraModel(6, "FULL_OPENCV", 12), CameraModel(7, "FOV", 5), CameraModel(8, "SIMPLE_RADIAL_FISHEYE", 4),
    CameraModel(9, "RADIAL_FISHEYE", 5), CameraModel(10, "THIN_PRISM_FISHEYE", 12))}


def qvec2rotmat(q):
    """scene/colmap_loader.py:43-53"""
    return np.array([
        [1 - 2 * q[2] ** 2 - 2 * q[3] ** 2, 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[3] * q[1] + 2 * q[0] * q[2]],
        [2 * q[1] * q[2] + 2 * q[0] * q[3], 1 - 2 * q[1] ** 2 - 2 * q[3] ** 2, 2 * q[2] * q[3] - 2 * q[0] * q[1]],
        [2 * q[3] * q[1] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], 1 - 2 * q[1] ** 2 - 2 * q[2] ** 2]])


def _rd(f, n, fmt):
    return struct.unpack("<" + fmt, f.read(n))


def read_cameras_binary(path):
    """scene/colmap_loader.py:215-242"""
    cams = {}
    with open(path, "rb") as f:
        for _ in range(_rd(f, 8, "Q")[0]):
            cid, mid, w, h = _rd(f, 24, "iiQQ")
            n = CAMERA_MODELS[mid].num_params
            cams[cid] = ColmapCamera(cid, CAMERA_MODELS[mid].model_name, w, h, np.array(_rd(f, 8 * n, "d" * n)))
    return cams


def read_images_binary(path):
    """scene/colmap_loader.py:180-212"""
    images = {}
    with open(path, "rb") as f:
        for _ in range(_rd(f, 8, "Q")[0]):
            p = _rd(f, 64, "idddddddi")
            name = b""
            c = f.read(1)
            while c != b"\x00":
                name += c
                c = f.read(1)
            m = _rd(f, 8, "Q")[0]
            x = _rd(f, 24 * m, "ddq" * m)
            xys = np.column_stack([tuple(map(float, x[0::3])), tuple(map(float, x[1::3]))]) if m else np.zeros((0, 2))
            images[p[0]] = ColmapImage(p[0], np.array(p[1:5]), np.array(p[5:8]), p[8], name.decode("utf-8"), xys,
                                       np.array(tuple(map(int, x[2::3]))))
    return images


def read_points3D_binary(path):
    """scene/colmap_loader.py:125-153 -> (xyz (n,3) f64, rgb (n,3), error (n,1))"""
    with open(path, "rb") as f:
        n = _rd(f, 8, "Q")[0]
        xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
        for i in range(n):
            p = _rd(f, 43, "QdddBBBd")
            xyz[i], rgb[i], err[i] = p[1:4], p[4:7], p[7]
            t = _rd(f, 8, "Q")[0]
            f.read(8 * t)
    return xyz, rgb, err


def read_cameras_text(path):
    """scene/colmap_loader.py:156-178"""
    cams = {}
    for line in open(path):
        line = line.strip()
        if line and line[0] != "#":
            e = line.split()
            assert e[1] == "PINHOLE", "While the loader support other types, the rest of the code assumes PINHOLE"
            cams[int(e[0])] = ColmapCamera(int(e[0]), e[1], int(e[2]), int(e[3]), np.array(tuple(map(float, e[4:]))))
    return cams


def read_images_text(path):
    """scene/colmap_loader.py:244-271 (two lines per image)"""
    images = {}
    with open(path) as f:
        while True:
            line = f.readline()
            if not line:
                break
            line = line.strip()
            if line and line[0] != "#":
                e = line.split()
                e2 = f.readline().split()
                images[int(e[0])] = ColmapImage(int(e[0]), np.array(tuple(map(float, e[1:5]))), np.array(tuple(map(float, e[5:8]))),
                                                int(e[8]), e[9], np.column_stack([tuple(map(float, e2[0::3])), tuple(map(float, e2[1::3]))]) if e2 else np.zeros((0, 2)),
                                                np.array(tuple(map(int, e2[2::3]))))
    return images


def read_points3D_text(path):
    """scene/colmap_loader.py:83-123"""
    xyz, rgb, err = [], [], []
    for line in open(path):
        line = line.strip()
        if line and line[0] != "#":
            e = line.split()
            xyz.append(tuple(map(float, e[1:4]))); rgb.append(tuple(map(int, e[4:7]))); err.append(float(e[7]))
    return np.array(xyz).reshape(-1, 3), np.array(rgb).reshape(-1, 3), np.array(err).reshape(-1, 1)


def colmap_cameras_to_views(cam_extrinsics, cam_intrinsics):
    """readColmapCameras, scene/dataset_readers.py:70-107 without the image load: per image
    (R = qvec2rotmat(q)^T, T = tvec, FovX, FovY, width, height, name), sorted by name (:148)."""
    from .cameras import focal2fov
    out = []
    for key in cam_extrinsics:
        ex = cam_extrinsics[key]
        intr = cam_intrinsics[ex.camera_id]
        R, T = np.transpose(qvec2rotmat(ex.qvec)), np.array(ex.tvec)
        if intr.model == "SIMPLE_PINHOLE":
            fy = fx = intr.params[0]
        elif intr.model == "PINHOLE":
            fx, fy = intr.params[0], intr.params[1]
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE cameras) supported!")
        out.append(dict(uid=intr.id, R=R, T=T, FovX=focal2fov(fx, intr.width), FovY=focal2fov(fy, intr.height),
                        width=intr.width, height=intr.height, image_name=os.path.basename(ex.name).split(".")[0], name=ex.name))
    return sorted(out, key=lambda c: c["image_name"])


def nerf_normalization(views):
    """getNerfppNorm, scene/dataset_readers.py:47-68 -> (translate, radius); radius = cameras_extent."""
    from .cameras import getWorld2View2
    centers = [np.linalg.inv(getWorld2View2(v["R"], v["T"]))[:3, 3:4] for v in views]
    cc = np.hstack(centers)
    avg = np.mean(cc, axis=1, keepdims=True)
    diagonal = np.max(np.linalg.norm(cc - avg, axis=0, keepdims=True))
    return -avg.flatten(), diagonal * 1.1
