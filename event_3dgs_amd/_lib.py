"""Loader of the C-ABI HIP library (include/e3dgs_hip.h) -- ctypes, no torch types cross it.

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libe3dgs_hip.so")
ABI_VERSION = 17

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
NOTIFY_FN = C.CFUNCTYPE(None, C.c_void_p)
_fp, _ip, _vp, _cp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; import it FIRST so that our library binds to the same
    # HIP runtime instance (two runtimes in one process -> "no ROCm-capable device", hipError 100).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.e3dgs_abi_version.restype = C.c_int
    if L.e3dgs_abi_version() != ABI_VERSION:
        raise HipLibraryError("libe3dgs_hip.so ABI version mismatch; rebuild")
    L.e3dgs_last_error.restype = C.c_char_p
    L.e3dgs_rasterize_forward.restype = C.c_int
    L.e3dgs_rasterize_forward.argtypes = (
        [ALLOC_FN, _vp] * 3 + [C.c_int] * 3 + [_fp, C.c_int, C.c_int] + [_fp] * 5 + [C.c_float] + [_fp] * 5
        + [C.c_float, C.c_float, C.c_int, _fp, _ip, C.c_int, C.c_int, C.POINTER(C.c_int), _vp])
    L.e3dgs_rasterize_forward_begin.restype = C.c_int
    L.e3dgs_rasterize_forward_begin.argtypes = (
        [ALLOC_FN, _vp] * 2 + [C.c_int] * 5 + [_fp] * 5 + [C.c_float] + [_fp] * 5 + [C.c_float, C.c_float, _ip, C.c_int,
                                                                                   C.c_int, _vp, _vp])
    L.e3dgs_rasterize_forward_finish.restype = C.c_int
    L.e3dgs_rasterize_forward_finish.argtypes = [ALLOC_FN, _vp, C.c_int, C.c_int, C.c_int, _fp, _cp, _cp, C.c_int, _fp,
                                                 C.c_int, C.c_int, _vp]
    L.e3dgs_rasterize_backward.restype = C.c_int
    L.e3dgs_rasterize_backward.argtypes = (
        [C.c_int] * 4 + [_fp, C.c_int, C.c_int] + [_fp] * 5 + [C.c_float] + [_fp] * 5 + [C.c_float, C.c_float]
        + [_ip] + [_cp] * 3 + [_fp] * 10 + [C.c_int, C.c_int, _vp])
    _pp, _hf = C.POINTER(C.c_void_p), C.POINTER(C.c_float)      # host arrays: per-view device pointers / floats
    L.e3dgs_rasterize_forward_multi.restype = C.c_int
    L.e3dgs_rasterize_forward_multi.argtypes = (
        [ALLOC_FN, _vp] * 3 + [C.c_int] * 4 + [_fp, C.c_int, C.c_int] + [_fp] * 5 + [C.c_float] + [_fp] * 2
        + [_pp] * 3 + [_hf] * 2 + [_fp, _ip, C.c_int, C.c_int, C.POINTER(C.c_int), _vp])
    L.e3dgs_rasterize_forward_multi_begin.restype = C.c_int
    L.e3dgs_rasterize_forward_multi_begin.argtypes = (
        [ALLOC_FN, _vp] * 2 + [C.c_int] * 6 + [_fp] * 5 + [C.c_float] + [_fp] * 2 + [_pp] * 3 + [_hf] * 2
        + [_ip, C.c_int, C.c_int, _vp, _vp])
    L.e3dgs_rasterize_forward_multi_finish.restype = C.c_int
    L.e3dgs_rasterize_forward_multi_finish.argtypes = [ALLOC_FN, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _cp, _cp,
                                                       C.c_int, _fp, C.c_int, C.c_int, _vp]
    L.e3dgs_rasterize_forward_multi_finish_colour.restype = C.c_int
    L.e3dgs_rasterize_forward_multi_finish_colour.argtypes = (
        [ALLOC_FN, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _cp, _cp, C.c_int, _fp, C.c_int]
        + [C.c_int, C.c_int, _fp, _fp, _pp, C.c_int, NOTIFY_FN, _vp, _vp])
    L.e3dgs_rasterize_forward_multi_capacity.restype = C.c_int
    L.e3dgs_rasterize_forward_multi_capacity.argtypes = (
        [ALLOC_FN, _vp] * 3 + [C.c_int] * 4 + [_fp, C.c_int, C.c_int] + [_fp] * 4 + [C.c_float, _fp] + [_pp] * 3 + [_hf] * 2
        + [_fp, _ip, C.c_int, C.c_int, C.c_int, _vp, NOTIFY_FN, _vp, _vp])
    L.e3dgs_rasterize_backward_multi.restype = C.c_int
    L.e3dgs_rasterize_backward_multi.argtypes = (
        [C.c_int] * 5 + [_fp, C.c_int, C.c_int] + [_fp] * 4 + [C.c_float, _fp] + [_pp] * 3 + [_hf] * 2
        + [_ip] + [_cp] * 3 + [_fp] * 9 + [C.c_int, C.c_int, _vp])
    L.e3dgs_rasterize_backward_multi_stats.restype = C.c_int
    L.e3dgs_rasterize_backward_multi_stats.argtypes = (
        [C.c_int] * 5 + [_fp, C.c_int, C.c_int] + [_fp] * 4 + [C.c_float, _fp] + [_pp] * 3 + [_hf] * 2
        + [_ip] + [_cp] * 3 + [_fp] * 10 + [C.c_int, C.c_int, _vp])
    L.e3dgs_rasterize_backward_multi_rank1.restype = C.c_int
    L.e3dgs_rasterize_backward_multi_rank1.argtypes = (
        [C.c_int] * 5 + [_fp, C.c_int, C.c_int] + [_fp] * 4 + [C.c_float, _fp] + [_pp] * 3 + [_hf] * 2
        + [_ip] + [_cp] * 3 + [_fp] * 2 + [_hf, C.c_uint] + [_fp] * 8 + [C.c_int, C.c_int, _vp])
    L.e3dgs_sh_grad_from_colour.restype = C.c_int
    L.e3dgs_sh_grad_from_colour.argtypes = [C.c_int] * 5 + [_fp, _fp, C.c_size_t, C.c_float, _fp, C.c_int, _vp]
    L.e3dgs_sh_adam_from_colour.restype = C.c_int
    L.e3dgs_sh_adam_from_colour.argtypes = ([C.c_int] * 5 + [_fp, _fp, C.c_size_t, C.c_float, _fp, _fp, _fp]
                                            + [C.c_float] * 5 + [C.c_int, C.c_int, _vp])
    L.e3dgs_sh_adam_from_colour_mean.restype = C.c_int
    L.e3dgs_sh_adam_from_colour_mean.argtypes = L.e3dgs_sh_adam_from_colour.argtypes[:-1] + [_fp, _vp]
    L.e3dgs_set_tile_cull.restype = None
    L.e3dgs_set_tile_cull.argtypes = [C.c_int]
    L.e3dgs_get_tile_cull.restype = C.c_int
    L.e3dgs_set_small_scene_paths.restype = None
    L.e3dgs_set_small_scene_paths.argtypes = [C.c_int]
    L.e3dgs_get_small_scene_paths.restype = C.c_int
    L.e3dgs_state_offset_emit_gid.restype = C.c_size_t
    L.e3dgs_state_offset_emit_gid.argtypes = [C.c_int]
    L.e3dgs_state_offsets_binning.restype = None
    L.e3dgs_state_offsets_binning.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
    L.e3dgs_state_offsets.restype = None
    L.e3dgs_state_offsets.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)]
    L.e3dgs_state_offsets_multi.restype = None
    L.e3dgs_state_offsets_multi.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_size_t)]
    L.e3dgs_mark_visible.restype = C.c_int
    L.e3dgs_mark_visible.argtypes = [C.c_int, _fp, _fp, _fp, _vp, _vp]
    L.e3dgs_knn_scratch_bytes.restype = C.c_size_t
    L.e3dgs_knn_scratch_bytes.argtypes = [C.c_int]
    L.e3dgs_dist_knn3.restype = C.c_int
    L.e3dgs_dist_knn3.argtypes = [C.c_int, _fp, _fp, _cp, _vp]
    L.e3dgs_event_loss_scratch_bytes.restype = C.c_size_t
    L.e3dgs_event_loss_scratch_bytes.argtypes = [C.c_int, C.c_int]
    L.e3dgs_event_loss.restype = C.c_int
    L.e3dgs_event_loss.argtypes = [C.c_int, C.c_int] + [_fp] * 8 + [C.c_float] + [_fp] * 5 + [_cp, _vp]
    L.e3dgs_event_loss_cached.restype = C.c_int
    L.e3dgs_event_loss_cached.argtypes = [C.c_int, C.c_int] + [_fp] * 8 + [C.c_float] + [_fp] * 5 + [_vp, C.c_int, _cp, _vp]
    L.e3dgs_event_loss_rank1.restype = C.c_int
    L.e3dgs_event_loss_rank1.argtypes = L.e3dgs_event_loss_cached.argtypes
    L.e3dgs_image_loss_rank1.restype = C.c_int
    L.e3dgs_image_loss_rank1.argtypes = [C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, _fp, _cp, _vp]
    L.e3dgs_ssim_scratch_bytes.restype = C.c_size_t
    L.e3dgs_ssim_scratch_bytes.argtypes = [C.c_int] * 3
    L.e3dgs_ssim.restype = C.c_int
    L.e3dgs_ssim.argtypes = [C.c_int] * 4 + [_fp] * 4 + [_cp, _vp]
    L.e3dgs_adam_step_segments.restype = C.c_int
    L.e3dgs_adam_step_segments.argtypes = [C.c_size_t] + [_fp] * 4 + [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                                                       C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int, _vp]
    L.e3dgs_adam_step_groups.restype = C.c_int
    L.e3dgs_adam_step_groups.argtypes = [C.c_size_t] + [_fp] * 4 + [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_float),
                                                                     C.POINTER(C.c_float), C.c_float, C.c_float,
                                                                     C.POINTER(C.c_int), _vp]
    L.e3dgs_adam_step_groups_gap.restype = C.c_int
    L.e3dgs_adam_step_groups_gap.argtypes = L.e3dgs_adam_step_groups.argtypes[:-1] + [C.c_size_t, C.c_size_t, _vp]
    L.e3dgs_densify_scratch_bytes.restype = C.c_size_t
    L.e3dgs_densify_scratch_bytes.argtypes = [C.c_int]
    L.e3dgs_densify_plan.restype = C.c_int
    L.e3dgs_densify_plan.argtypes = [C.c_int, _fp, _fp, _fp] + [C.c_float] * 4 + [C.c_int, _cp, C.POINTER(C.c_int), _vp]
    L.e3dgs_densify_split_rows.restype = C.c_void_p
    L.e3dgs_densify_split_rows.argtypes = [C.c_int, _cp]
    L.e3dgs_densify_apply.restype = C.c_int
    L.e3dgs_densify_apply.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)] + [_fp] * 7 + [_cp, _vp]
    L.e3dgs_densify_stats_update.restype = C.c_int
    L.e3dgs_densify_stats_update.argtypes = [C.c_int, _fp, _ip, _fp, _fp, _fp, _vp]
    L.e3dgs_image_loss_scratch_bytes.restype = C.c_size_t
    L.e3dgs_image_loss_scratch_bytes.argtypes = [C.c_int] * 3
    L.e3dgs_image_loss.restype = C.c_int
    L.e3dgs_image_loss.argtypes = [C.c_int] * 4 + [C.c_float, _fp, _fp, _fp, _fp, _cp, _vp]
    L.e3dgs_adam_step.restype = C.c_int
    L.e3dgs_adam_step.argtypes = [C.c_size_t] + [_fp] * 4 + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_int, C.c_int, _vp]
    L.e3dgs_sort_scratch_bytes.restype = C.c_size_t
    L.e3dgs_sort_scratch_bytes.argtypes = [C.c_size_t]
    L.e3dgs_depth_sort_scratch_bytes.restype = C.c_size_t
    L.e3dgs_depth_sort_scratch_bytes.argtypes = [C.c_size_t]
    L.e3dgs_sort_depth_keys.restype = C.c_int
    L.e3dgs_sort_depth_keys.argtypes = [C.c_size_t, _vp, _vp, _vp, _vp, _cp, _vp, _vp]
    L.e3dgs_sort_pairs.restype = C.c_int
    L.e3dgs_sort_pairs.argtypes = [C.c_size_t, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _cp, _vp, _vp, C.c_uint32,
                                   C.POINTER(C.c_int), _vp]
    L.e3dgs_profile_enable.restype = None
    L.e3dgs_profile_enable.argtypes = [C.c_int]
    L.e3dgs_profile_select.restype = None
    L.e3dgs_profile_select.argtypes = [C.c_int]
    L.e3dgs_profile_query.restype = C.c_int
    L.e3dgs_profile_query.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.e3dgs_profile_slot_name.restype = C.c_char_p
    L.e3dgs_profile_slot_name.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().e3dgs_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    """Raw handle of torch's current HIP stream on the current device.  The private raw getter costs ~0.3 us against
    ~8 us for torch.cuda.current_stream() (a Stream object per call) -- ten calls per training step."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


FLAG_PREACT = 1
FLAG_ACCUMULATE = 2
FLAG_SH_PLANAR = 4
FLAG_BWD_ONLY_RENDER = 8
FLAG_BWD_ONLY_GEOM = 16
FLAG_COUNT_MAPPED = 64
FLAG_DEFER_COLOR = 128
FLAG_COUNT_DEVICE = 256
FLAG_DEFER_SH_MEAN = 512
# per-call options (include/e3dgs_hip.h): with FLAG_OPTIONS the bits describe the call and no process-wide default is read
FLAG_OPTIONS = 0x0800
FLAG_CULL_RECT = 0x1000
FLAG_CULL_NO_BOX = 0x2000
FLAG_NO_SMALL_PATHS = 0x4000
FLAG_FAST_EXP = 0x8000
FLAG_MEAN2D_VIEWS = 0x10000
OPTION_MASK = FLAG_OPTIONS | FLAG_CULL_RECT | FLAG_CULL_NO_BOX | FLAG_NO_SMALL_PATHS | FLAG_FAST_EXP
ACC_STRIDE = 12


def option_flags(tile_cull=None, small_scene_paths=None, fast_exp=False):
    """The option bits of a call.  tile_cull: True / 1 exact culling, False / 0 the reference's rectangle binning, 3 exact
    culling without the tight candidate box; small_scene_paths: adapt the work decomposition to few splats.  Leaving both
    None keeps the process-wide defaults (environment E3DGS_TILE_CULL / E3DGS_SMALL_SCENE_PATHS, deprecated setters);
    giving either makes the call self-describing (missing ones default to ON)."""
    f = FLAG_FAST_EXP if fast_exp else 0
    if tile_cull is None and small_scene_paths is None:
        return f
    f |= FLAG_OPTIONS
    if tile_cull is not None:
        if int(tile_cull) == 0:
            f |= FLAG_CULL_RECT
        elif int(tile_cull) == 3:
            f |= FLAG_CULL_NO_BOX
    if small_scene_paths is not None and not small_scene_paths:
        f |= FLAG_NO_SMALL_PATHS
    return f

EXPORTED_SYMBOLS = [
    "e3dgs_abi_version", "e3dgs_last_error", "e3dgs_rasterize_forward", "e3dgs_rasterize_forward_begin", "e3dgs_rasterize_forward_finish",
    "e3dgs_rasterize_backward", "e3dgs_rasterize_forward_multi", "e3dgs_rasterize_forward_multi_begin",
    "e3dgs_rasterize_forward_multi_finish", "e3dgs_rasterize_forward_multi_finish_colour", "e3dgs_rasterize_backward_multi", "e3dgs_rasterize_backward_multi_stats",
    "e3dgs_sh_grad_from_colour", "e3dgs_sh_adam_from_colour", "e3dgs_sh_adam_from_colour_mean",
    "e3dgs_set_tile_cull", "e3dgs_get_tile_cull", "e3dgs_set_small_scene_paths", "e3dgs_get_small_scene_paths", "e3dgs_state_offsets", "e3dgs_state_offsets_multi", "e3dgs_state_offset_emit_gid", "e3dgs_mark_visible", "e3dgs_knn_scratch_bytes", "e3dgs_dist_knn3", "e3dgs_event_loss_scratch_bytes",
    "e3dgs_event_loss", "e3dgs_event_loss_cached", "e3dgs_ssim_scratch_bytes", "e3dgs_ssim", "e3dgs_image_loss_scratch_bytes", "e3dgs_image_loss", "e3dgs_densify_stats_update", "e3dgs_densify_scratch_bytes", "e3dgs_densify_plan", "e3dgs_densify_split_rows", "e3dgs_densify_apply", "e3dgs_adam_step_segments", "e3dgs_adam_step_groups", "e3dgs_adam_step_groups_gap", "e3dgs_adam_step", "e3dgs_profile_enable", "e3dgs_profile_select", "e3dgs_profile_query", "e3dgs_profile_slot_name",
    "e3dgs_sort_scratch_bytes", "e3dgs_sort_pairs", "e3dgs_rasterize_forward_multi_capacity",
    "e3dgs_rasterize_backward_multi_rank1", "e3dgs_event_loss_rank1", "e3dgs_image_loss_rank1",
    "e3dgs_state_offsets_binning", "e3dgs_depth_sort_scratch_bytes", "e3dgs_sort_depth_keys",
]
