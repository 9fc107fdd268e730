"""GPU: the radix sort used for binning is a correct STABLE sort -- called on its own through the C ABI
(e3dgs_sort_pairs) against numpy's stable argsort for every variant the rasteriser uses (32-bit keys with the
compacting first pass of the depth sort; 16-/32-bit tile keys with one, two and three passes and the tile ranges
derived inside the sort), and end to end through the rasteriser's sorted lists."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(keys, nbits, key_bytes, identity=True, drop=False, nranges=0):
    from event_3dgs_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n = len(keys)
    kt = torch.int32 if key_bytes == 4 else torch.int16
    k0 = torch.from_numpy(keys.astype(np.uint32 if key_bytes == 4 else np.uint16).view(np.int32 if key_bytes == 4 else np.int16)).to(dev)
    k1 = torch.empty_like(k0)
    vals = np.arange(n, dtype=np.int32) if identity else (np.arange(n, dtype=np.int32) * 7 + 3)
    v0 = torch.from_numpy(vals).to(dev) if not identity else torch.full((n,), -1, dtype=torch.int32, device=dev)
    v1 = torch.full((n,), -1, dtype=torch.int32, device=dev)
    scratch = torch.empty(L.e3dgs_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    kept = torch.full((1,), -1, dtype=torch.int32, device=dev) if drop else None
    ranges = torch.zeros(max(nranges, 1), 2, dtype=torch.int32, device=dev) if nranges else None
    idx = C.c_int(-1)
    rc = L.e3dgs_sort_pairs(n, nbits, key_bytes, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(v0), _lib.ptr(v1), int(identity),
                            _lib.ptr(scratch), _lib.ptr(kept), _lib.ptr(ranges), nranges, C.byref(idx), _lib.current_stream())
    _lib.check(rc, "e3dgs_sort_pairs")
    torch.cuda.synchronize()
    vout = (v0, v1)[idx.value].cpu().numpy()
    kout = (k0, k1)[idx.value].cpu().numpy()
    kout = kout.view(np.uint32 if key_bytes == 4 else np.uint16)
    return (kout, vout, None if kept is None else int(kept[0]), None if ranges is None else ranges.cpu().numpy(), vals)


def _reference(keys, nbits, vals, drop=False):
    mask = (1 << nbits) - 1 if nbits < 32 else 0xFFFFFFFF
    k = keys.astype(np.uint64) & mask
    sel = np.arange(len(keys))
    if drop:
        sel = sel[keys != 0xFFFFFFFF]
    order = sel[np.argsort(k[sel], kind="stable")]
    return keys[order], vals[order]


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 300_001])
@pytest.mark.parametrize("identity", [True, False])
def test_depth_sort_32_bit_keys_with_and_without_the_compacting_first_pass(n, identity):
    rs = np.random.RandomState(n)
    keys = rs.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    keys[rs.rand(n) < 0.35] = 0xFFFFFFFF                     # culled splats
    keys[rs.rand(n) < 0.05] = keys[0]                        # ties keep index order
    for drop in (False, True):
        kout, vout, kept, _, vals = _sort(keys, 32, 4, identity=identity, drop=drop)
        kref, vref = _reference(keys, 32, vals, drop)
        m = len(kref)
        if drop:
            assert kept == m
        assert np.array_equal(kout[:m], kref) and np.array_equal(vout[:m], vref), (n, identity, drop)


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 300_001, 3_000_000])
@pytest.mark.parametrize("kind", ["random", "depths", "few"])
def test_depth_sort_in_three_11_bit_passes(n, kind):
    """e3dgs_sort_depth_keys (what the forward runs): all 32 key bits in passes of 11 + 11 + 10, culled splats dropped,
    ties in index order -- against numpy's stable argsort."""
    from event_3dgs_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(n + len(kind))
    if kind == "random":
        keys = rs.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    elif kind == "depths":            # float bits of view-space depths 0.2 ... 100 (what the projection stores)
        keys = np.exp(rs.uniform(np.log(0.2), np.log(100.0), n)).astype(np.float32).view(np.uint32).copy()
    else:                             # three distinct values: every digit of every pass collides
        keys = rs.choice(np.array([0x40000000, 0x40000800, 0x40400000], np.uint32), n)
    keys[rs.rand(n) < 0.35] = 0xFFFFFFFF
    keys[rs.rand(n) < 0.05] = keys[0]
    k0 = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    k1 = torch.empty_like(k0)
    v0 = torch.full((n,), -1, dtype=torch.int32, device=dev)
    v1 = torch.full((n,), -1, dtype=torch.int32, device=dev)
    scratch = torch.empty(L.e3dgs_depth_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    kept = torch.full((1,), -1, dtype=torch.int32, device=dev)
    rc = L.e3dgs_sort_depth_keys(n, _lib.ptr(k0), _lib.ptr(k1), _lib.ptr(v0), _lib.ptr(v1), _lib.ptr(scratch), _lib.ptr(kept),
                                 _lib.current_stream())
    _lib.check(rc, "e3dgs_sort_depth_keys")
    torch.cuda.synchronize()
    sel = np.arange(n)[keys != 0xFFFFFFFF]
    ref = sel[np.argsort(keys[sel], kind="stable")]
    assert int(kept[0]) == len(ref)
    assert np.array_equal(v0.cpu().numpy()[:len(ref)], ref.astype(np.int32))


def test_all_keys_dropped():
    keys = np.full(10_000, 0xFFFFFFFF, np.uint32)
    _, _, kept, _, _ = _sort(keys, 32, 4, drop=True)
    assert kept == 0


@pytest.mark.parametrize("key_bytes,ntiles", [(2, 7), (2, 256), (2, 257), (2, 24_480), (2, 65_536), (4, 24_480), (4, 97_200 * 3)])
@pytest.mark.parametrize("n", [5, 70_001, 1_200_000])
def test_tile_sort_and_the_ranges_it_derives(key_bytes, ntiles, n):
    """One pass (<= 256 tiles), two passes (ranges inside the last pass, no sorted keys), three passes (32-bit keys)."""
    rs = np.random.RandomState(ntiles + n)
    nbits = max(1, int(np.ceil(np.log2(max(ntiles, 2)))))
    # clustered tile ids with whole low-digit segments empty and tiles without an instance
    keys = (rs.randint(0, ntiles, n, dtype=np.int64) // 3 * 3 % ntiles).astype(np.uint32)
    if ntiles > 600:
        keys[(keys & 255) == 17] += 1
    kout, vout, _, ranges, vals = _sort(keys, nbits, key_bytes, identity=True, nranges=ntiles)
    kref, vref = _reference(keys, nbits, vals)
    assert np.array_equal(vout, vref)
    two_pass = 8 < nbits <= 16
    if not two_pass:
        assert np.array_equal(kout, kref.astype(kout.dtype))
    # ranges: [first, last + 1) of every tile that has an instance; empty ranges elsewhere
    first = np.searchsorted(kref, np.arange(ntiles), side="left")
    last = np.searchsorted(kref, np.arange(ntiles), side="right")
    has = last > first
    assert np.array_equal(ranges[has, 0], first[has]) and np.array_equal(ranges[has, 1], last[has])
    assert np.all(ranges[~has, 0] == ranges[~has, 1])


def _lists(n_gauss, W, H, seed):
    from event_3dgs_amd import rasterizer
    from helpers import scene
    from test_hip_parity import _settings
    dev = torch.device("cuda:0")
    act, cam = scene(n_gauss, W, H, seed=seed)
    rs = _settings(cam, (0, 0, 0), dev)
    d = lambda t: t.to(dev)
    raw = rasterizer.forward_raw(d(act["means3D"]), None, d(act["colors"]), d(act["opacities"]), d(act["scales"]),
                                 d(act["rotations"]), None, rs)
    torch.cuda.synchronize()
    st = rasterizer.state_views(raw, n_gauss, W, H)
    return raw["num_rendered"], st["point_list"].cpu().numpy().copy(), st["ranges"].cpu().numpy().copy(), act, cam


@pytest.mark.parametrize("n_gauss,W,H", [(30000, 640, 480), (200000, 1280, 720)])
def test_sorted_lists_are_depth_ordered(n_gauss, W, H):
    """Inside every tile the list is ordered by (view-space depth, index) -- recomputed here in fp32 as the kernel does."""
    I, pl, rg, act, cam = _lists(n_gauss, W, H, seed=5)
    assert I > 10 * 4096          # many sort workgroups -> the look-back chains of the scans are exercised
    V = cam.world_view_transform.contiguous().numpy().astype(np.float32).reshape(-1)
    m = act["means3D"].numpy().astype(np.float32)
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    # XFORM(V, 2, x, y, z) = fma(V[2], x, fma(V[6], y, fma(V[10], z, V[14])))
    z = fma(np.full_like(m[:, 0], V[2]), m[:, 0], fma(np.full_like(m[:, 0], V[6]), m[:, 1],
            fma(np.full_like(m[:, 0], V[10]), m[:, 2], np.full_like(m[:, 0], V[14]))))
    assert int(rg[:, 1].max()) == I and np.all(rg[:, 0] <= rg[:, 1])
    checked = 0
    for t in np.random.RandomState(0).permutation(len(rg))[:400]:
        a, b = rg[t]
        if b - a < 2:
            continue
        ids = pl[a:b]
        key = z[ids].view(np.uint32).astype(np.uint64) << 32 | ids.astype(np.uint64)
        assert np.all(key[1:] > key[:-1]), t
        checked += 1
    assert checked > 50
