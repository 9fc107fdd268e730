/*
 * e3dgs_hip.h -- C ABI of the MI355X (gfx950) Gaussian-splat hot path.
 *
 * This is the drop-in boundary of the repo: a plain C interface (raw device
 * pointers, sizes, a hipStream_t passed as void*) with no torch types.  Each
 * entry point replaces one function of the reference's un-vendored CUDA
 * submodules; the reference-side call sites are cited per function.
 *
 *   reference import  gaussian_renderer/__init__.py:15
 *       from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
 *   reference import  scene/gaussian_model.py:20
 *       from simple_knn._C import distCUDA2
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - matrices are 16 floats in the reference's row-vector layout
 *     (scene/cameras.py:54-56): flat[4*c + r] = row r, column c of the
 *     textbook column-vector matrix;
 *   - every function enqueues on `stream` and returns 0 (hipSuccess) or a
 *     non-zero hipError_t value; e3dgs_last_error() gives the text;
 *   - nothing here allocates device memory: scratch comes from the caller
 *     through e3dgs_alloc_fn (the reference op grows three torch uint8 tensors
 *     the same way and keeps them alive in the autograd ctx).
 */
#ifndef E3DGS_HIP_H
#define E3DGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Returns a device pointer to at least `bytes` bytes (256-B aligned), owned by the caller. */
typedef char* (*e3dgs_alloc_fn)(void* user, size_t bytes);
/* Plain host-side notification (see e3dgs_rasterize_forward_multi_finish_colour). */
typedef void (*e3dgs_notify_fn)(void* user);

/* ABI version; bumped on any signature change. */
int e3dgs_abi_version(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char* e3dgs_last_error(void);

/* flags for e3dgs_rasterize_forward / e3dgs_rasterize_backward */
#define E3DGS_FLAG_PREACT 1      /* scales = log-scales, rotations = raw quaternions, opacities = logits: the
                                    activations of scene/gaussian_model.py:95-118 (exp, normalize, sigmoid) run
                                    inside the kernels, and backward returns gradients w.r.t. the raw parameters */
#define E3DGS_FLAG_ACCUMULATE 2  /* backward ADDS into dL_dmean3D / dL_dsh / dL_dscale / dL_drot / dL_dopacity /
                                    dL_dcolor / dL_dcov3D (visible Gaussians only) instead of overwriting them: the
                                    three renders of one training iteration (train.py:144,159,161) accumulate
                                    straight into one gradient buffer */

#define E3DGS_FLAG_SH_PLANAR 4   /* shs and dL_dsh are coefficient-major, (M*3, P), instead of the reference's
                                    (P,M,3): neighbouring lanes read neighbouring addresses (used by the fused
                                    trainer, which owns its parameter layout) */

#define E3DGS_FLAG_COUNT_MAPPED 64  /* forward_begin / forward_multi_begin: num_rendered_host points to PINNED host memory
                                       that the device can address (hipHostMalloc; a torch pinned tensor).  The GPU
                                       stores the count there itself (no copy command), so the caller may arm the
                                       word with a sentinel (-1) and poll it instead of synchronising the stream;
                                       with P == 0 the library writes 0 at once. */
#define E3DGS_FLAG_DEFER_COLOR 128  /* forward_multi_begin: do not evaluate the SH colours in the preprocess kernel.  The
                                       call MUST then be completed with e3dgs_rasterize_forward_multi_finish_colour,
                                       which evaluates them (same arithmetic, bit-identical) right before compositing.
                                       Nothing in front of the compositing kernel reads `shs`, so a trainer may still
                                       be reducing / updating the SH coefficients of the previous iteration on
                                       another stream while this iteration projects, sorts and bins. */
#define E3DGS_FLAG_COUNT_DEVICE 256   /* backward_multi: the forward was e3dgs_rasterize_forward_multi_capacity and
                                        num_rendered is its `capacity` (the count itself sits in the geometry scratch) */
#define E3DGS_FLAG_DEFER_SH_MEAN 512  /* backward_multi with dL_dcolour_views and WITHOUT dL_dsh (ABI 15): dL_dmean3D
                                        holds the geometric part only; the term through the SH view directions is added
                                        by e3dgs_sh_adam_from_colour_mean, which reads the coefficients anyway */
/* ---- per-call OPTIONS (ABI 13).  The library keeps no mutable process-wide state that a call reads when
 * E3DGS_FLAG_OPTIONS is set: calls with different settings may run concurrently from any number of host threads and
 * streams.  Every call of one forward / backward (begin, finish, backward) must be given the same option bits (and the
 * same P and view count): the halves derive scratch offsets and launch shapes from them on the host.  `begin` records
 * what it resolved -- cull mode, small-scene paths, E3DGS_FLAG_FAST_EXP, P * nviews -- on the HOST under the address of
 * the geometry scratch (no device read-back); `finish` and every backward entry point compare and return
 * hipErrorInvalidValue with a message, launching nothing, when they were given something else (ABI 17; until ABI 16 this
 * was undefined behaviour: out-of-bounds emission).  The geometry scratch must therefore stay at the address the
 * allocator returned (a scratch the caller moved is not recognised and not checked).
 * Without E3DGS_FLAG_OPTIONS the process-wide defaults apply (environment at load time, the deprecated setters
 * e3dgs_set_tile_cull / e3dgs_set_small_scene_paths). */
#define E3DGS_FLAG_OPTIONS 0x0800         /* the option bits below describe this call */
#define E3DGS_FLAG_CULL_RECT 0x1000       /* reference rectangle binning: no exact tile culling (see e3dgs_set_tile_cull) */
#define E3DGS_FLAG_CULL_NO_BOX 0x2000     /* exact culling without the tight candidate box (test switch) */
#define E3DGS_FLAG_NO_SMALL_PATHS 0x4000  /* large-scene work decomposition whatever the splat count (see
                                             e3dgs_set_small_scene_paths) */
#define E3DGS_FLAG_FAST_EXP 0x8000        /* TOLERANCE MODE (default off, independent of E3DGS_FLAG_OPTIONS): the compositing
                                             forward evaluates exp with the hardware v_exp_f32 (1 ulp) instead of the
                                             bit-reproducible polynomial -- integer outputs (radii, lists, ranges) are
                                             unchanged, the image agrees to <= 1e-4 except at the handful of pixels where
                                             an alpha >= 1/255 or T < 1e-4 decision flips (<= 1/255 each), gradients to
                                             1e-3.  Give the same bit to the backward. */
#define E3DGS_FLAG_MEAN2D_VIEWS 0x10000   /* backward_multi* (ABI 17): dL_dmean2D is (nviews, P, 3) and receives the NDC-unit
                                             screen-space gradient of EVERY view (block v = view v); without it, (P, 3):
                                             view 0's only -- what the densification statistics read, train.py:145 */
#define E3DGS_FLAG_BWD_ONLY_RENDER 8   /* backward: only the compositing backward (pixels -> grad_acc) */
#define E3DGS_FLAG_BWD_ONLY_GEOM 16    /* backward: only the per-Gaussian backward (grad_acc -> parameter gradients).
                                          Together these let a caller overlap the compositing backward of several
                                          views on separate HIP streams and serialise only the accumulating stage. */

/*
 * Forward rasterisation of P Gaussians into a (3,H,W) planar fp32 image.
 * Replaces: diff_gaussian_rasterization._C.rasterize_gaussians, called by
 * GaussianRasterizer.forward at gaussian_renderer/__init__.py:89-97 (and the
 * duplicates at :174-182, :342-350) with the settings built at :38-51.
 *
 *   D = active SH degree, M = SH coefficients per channel in `shs` (P,M,3).
 *   Exactly one of {shs, colors_precomp} and exactly one of
 *   {scales+rotations, cov3D_precomp} is non-NULL.
 *   out_color (3,H,W) and radii (P) are written in full (radii 0 = culled).
 *   geom/binning/image scratch is requested through the three allocators and
 *   must stay alive, unmodified, until the matching backward has run.
 *   Contains exactly one device->host wait (the instance count).  Since ABI 16 that wait is NOT a stream
 *   synchronisation: the GPU stores the count into a pinned word of the calling host thread (64 bytes, released
 *   when the thread exits) and the host polls it (cpu-relax between reads; after 2 s without a count it falls
 *   back to hipStreamSynchronize and reports that call's error).  The call therefore returns as soon as the count is
 *   known, with the rest of the stream still running: asynchronous errors of EARLIER work on the stream no longer
 *   surface here but at the caller's next synchronising call.  E3DGS_COUNT_POLL=0 in the environment, debug != 0
 *   and a failed pinned allocation take the stream synchronisation instead.
 *   *num_rendered_host receives the number of (tile, Gaussian) instances.
 */
int e3dgs_rasterize_forward(
    e3dgs_alloc_fn geom_alloc, void* geom_user,
    e3dgs_alloc_fn binning_alloc, void* binning_user,
    e3dgs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,          /* (3) */
    int width, int height,
    const float* means3D,             /* (P,3) */
    const float* shs,                 /* (P,M,3) or NULL */
    const float* colors_precomp,      /* (P,3) or NULL */
    const float* opacities,           /* (P) */
    const float* scales,              /* (P,3) or NULL */
    float scale_modifier,
    const float* rotations,           /* (P,4) (r,x,y,z) or NULL */
    const float* cov3D_precomp,       /* (P,6) or NULL */
    const float* viewmatrix,          /* (16) */
    const float* projmatrix,          /* (16) */
    const float* cam_pos,             /* (3) */
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,                 /* (3,H,W) */
    int* radii,                       /* (P) */
    int debug,
    int flags,                        /* E3DGS_FLAG_PREACT or 0 */
    int* num_rendered_host,
    void* stream);

/*
 * The same forward as two enqueue-only halves around the instance count.  A caller that renders
 * several views per iteration (train.py:144,159,161) issues every `begin`, synchronises the stream
 * ONCE, then issues every `finish`: one host synchronisation per iteration instead of one per render.
 *   begin : preprocess, depth sort, tile counting; enqueues the async copy of the instance count into
 *           *num_rendered_host (use pinned host memory); allocates geom + image scratch.
 *   finish: after the stream has been synchronised and *num_rendered_host read: binning, tile sort,
 *           compositing into out_color.  geom_buffer / image_buffer are the pointers `begin` obtained.
 * e3dgs_rasterize_forward(...) == begin + hipStreamSynchronize + finish.
 */
int e3dgs_rasterize_forward_begin(
    e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
    int* radii, int debug, int flags, int* num_rendered_host, void* stream);
int e3dgs_rasterize_forward_finish(
    e3dgs_alloc_fn binning_alloc, void* binning_user, int P, int width, int height, const float* background,
    char* geom_buffer, char* image_buffer, int num_rendered, float* out_color, int debug,
    int flags,                        /* the option bits given to `begin` (E3DGS_FLAG_OPTIONS ..., E3DGS_FLAG_FAST_EXP) */
    void* stream);

/*
 * Backward of the above.  Replaces:
 * diff_gaussian_rasterization._C.rasterize_gaussians_backward, reached from
 * loss.backward() at train.py:211 through _RasterizeGaussians.backward.
 *
 * grad_acc (num_rendered + P,12) floats is caller-owned scratch and needs NO initialisation (for the multi-view
 * entry point: (num_rendered + nviews*P,12)): the compositing backward stores one record per (tile, Gaussian)
 * instance (dmean2D.xy, dconic.xyz, dopacity, dcolor.rgb, 3 pad) at the instance's slot -- every record exactly
 * once, plain stores, no float atomics; a streaming pass then sums each Gaussian's contiguous run of records
 * in a fixed order into the P trailing rows, so gradients are bit-reproducible run to run (the reference's
 * atomics are not).  With BWD_ONLY_RENDER / BWD_ONLY_GEOM the same buffer must be passed to both calls.
 * Without E3DGS_FLAG_ACCUMULATE every other output is written in full (zeros for culled
 * Gaussians), so nothing else needs pre-zeroing.  dL_dmean2D is (P,3): first two components in
 * NDC units (consumed by scene/gaussian_model.py:405-407), third 0; always overwritten.
 * Optional outputs may be NULL: dL_dmean2D, dL_dopacity, dL_dcolor, dL_dcov3D.
 */
int e3dgs_rasterize_backward(
    int P, int D, int M, int num_rendered,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* opacities,           /* (P): only read with E3DGS_FLAG_PREACT */
    const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix,             /* (3,H,W) */
    float* grad_acc,                  /* (num_rendered + P,12) scratch (uninitialised is fine) */
    float* dL_dmean2D,                /* (P,3) or NULL */
    float* dL_dopacity,               /* (P) or NULL */
    float* dL_dcolor,                 /* (P,3) or NULL */
    float* dL_dmean3D,                /* (P,3) */
    float* dL_dcov3D,                 /* (P,6) or NULL */
    float* dL_dsh,                    /* (P,M,3) or NULL */
    float* dL_dscale,                 /* (P,3) or NULL */
    float* dL_drot,                   /* (P,4) or NULL */
    int debug,
    int flags,
    void* stream);

/*
 * SEVERAL VIEWS OF THE SAME GAUSSIANS IN ONE PASS (1 <= nviews <= 4, same width x height, same background).
 *
 * Replaces: the three rasterize_gaussians calls of one event iteration (train.py:144,159,161 ->
 * gaussian_renderer/__init__.py:89-97) and, for backward, the three rasterize_gaussians_backward calls plus
 * autograd's accumulation of their results into .grad that loss.backward() (train.py:211) performs.
 *
 * Every stage of the pipeline runs ONCE over all views ("splat" q = i * nviews + v, tile id
 * v * tiles + ty * gx + tx): nviews times fewer launches, parameters read once, one instance-count
 * read-back, and the backward writes every gradient element exactly once (sum over the views; zero
 * where no view saw the Gaussian) -- no pre-zeroing and no E3DGS_FLAG_ACCUMULATE.  Per view the
 * arithmetic is the single-view arithmetic: image v, radii[v] and the per-view gradients are
 * bit-identical to nviews separate e3dgs_rasterize_forward / _backward calls.
 *
 * viewmatrix / projmatrix / cam_pos are HOST arrays of nviews DEVICE pointers; tan_fovx / tan_fovy are
 * host arrays of nviews floats.  out_color is (nviews,3,H,W), radii (nviews,P), dL_dpix (nviews,3,H,W).
 * num_rendered counts the instances of all views.  The scratch buffers of a multi call are laid out for
 * nviews * P splats (see e3dgs_state_offsets with P * nviews / height * nviews for nviews > 1).
 * Backward needs shs + scales + rotations (no colors_precomp / cov3D_precomp); flags: PREACT, SH_PLANAR,
 * BWD_ONLY_RENDER, BWD_ONLY_GEOM.  dL_dmean2D (P,3), optional, receives view 0's screen-space gradient
 * (densification statistics use render #1 only, train.py:145).
 * SH degree: 0..4 (utils/sh_utils.py:57-112; M >= (D+1)^2) with the SH gradient dL_dsh as output; the stages that keep the
 * reference model's 16 coefficients per channel accept 0..3 only: E3DGS_FLAG_DEFER_COLOR (the separate colour stage),
 * dL_dcolour_views (the colour-gradient route) and e3dgs_sh_grad_from_colour / e3dgs_sh_adam_from_colour behind it.
 */
int e3dgs_rasterize_forward_multi(
    e3dgs_alloc_fn geom_alloc, void* geom_user,
    e3dgs_alloc_fn binning_alloc, void* binning_user,
    e3dgs_alloc_fn image_alloc, void* image_user,
    int nviews, int P, int D, int M,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp,
    const float* const* viewmatrix, const float* const* projmatrix, const float* const* cam_pos,
    const float* tan_fovx, const float* tan_fovy,
    float* out_color, int* radii, int debug, int flags,
    int* num_rendered_host, void* stream);

/* enqueue-only halves (as e3dgs_rasterize_forward_begin / _finish) */
int e3dgs_rasterize_forward_multi_begin(
    e3dgs_alloc_fn geom_alloc, void* geom_user,
    e3dgs_alloc_fn image_alloc, void* image_user,
    int nviews, int P, int D, int M, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp,
    const float* const* viewmatrix, const float* const* projmatrix, const float* const* cam_pos,
    const float* tan_fovx, const float* tan_fovy,
    int* radii, int debug, int flags,
    int* num_rendered_host,           /* HOST memory, valid after the stream is synchronised */
    void* stream);

int e3dgs_rasterize_forward_multi_finish(
    e3dgs_alloc_fn binning_alloc, void* binning_user,
    int nviews, int P, int width, int height, const float* background,
    char* geom_buffer, char* image_buffer, int num_rendered,
    float* out_color, int debug,
    int flags,                        /* the option bits given to `begin` */
    void* stream);

/*
 * The multi-view forward WITHOUT a host wait (a training loop: train.py:144-161 renders the same scene every iteration,
 * and the instance count moves by a few percent between iterations).  The caller sizes the binning buffers for
 * `capacity` instances BEFORE the count is known -- the previous iteration's count plus a margin -- and everything behind
 * the count pass reads the count from device memory: begin and finish are enqueued back to back, and so may be the
 * backward (e3dgs_rasterize_backward_multi with num_rendered = capacity: the scratch layouts are those of `capacity`
 * instances).  *num_rendered_host must be device-mapped pinned memory (E3DGS_FLAG_COUNT_MAPPED is required): the GPU
 * stores the count there; the caller polls it before it lets anything persistent consume the results (the optimizer
 * step).  If the count exceeds the capacity nothing was emitted or sorted (all tile lists empty, the image is the
 * background, the gradients are meaningless): the caller repeats the call with a larger capacity.
 * Results are bit-identical to begin + wait + finish.  With E3DGS_FLAG_DEFER_COLOR `before_colour` (may be NULL) is
 * called on the host right before the colour kernel is enqueued, as in ..._multi_finish_colour.
 */
int e3dgs_rasterize_forward_multi_capacity(
    e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn binning_alloc, void* binning_user,
    e3dgs_alloc_fn image_alloc, void* image_user, int nviews, int P, int D, int M, const float* background, int width,
    int height, const float* means3D, const float* shs, const float* opacities, const float* scales,
    float scale_modifier, const float* rotations, const float* const* viewmatrix, const float* const* projmatrix,
    const float* const* cam_pos, const float* tan_fovx, const float* tan_fovy, float* out_color, int* radii, int debug,
    int flags, int capacity, int* num_rendered_host, e3dgs_notify_fn before_colour, void* notify_user, void* stream);

/* finish() for a begin() issued with E3DGS_FLAG_DEFER_COLOR.  flags: E3DGS_FLAG_SH_PLANAR and the option bits as given to begin().
 * before_colour (optional) is called on the host immediately before the colour kernel is enqueued: make `stream`
 * wait there for whatever still writes `shs`. */
int e3dgs_rasterize_forward_multi_finish_colour(
    e3dgs_alloc_fn binning_alloc, void* binning_user,
    int nviews, int P, int width, int height, const float* background,
    char* geom_buffer, char* image_buffer, int num_rendered,
    float* out_color, int debug,
    int D, int M, const float* means3D, const float* shs, const float* const* cam_pos, int flags,
    e3dgs_notify_fn before_colour, void* notify_user,
    void* stream);

int e3dgs_rasterize_backward_multi(
    int nviews, int P, int D, int M, int num_rendered,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations,
    const float* const* viewmatrix, const float* const* projmatrix, const float* const* cam_pos,
    const float* tan_fovx, const float* tan_fovy,
    const int* radii,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix,             /* (nviews,3,H,W) */
    float* grad_acc,                  /* (num_rendered + nviews*P,12) scratch (uninitialised is fine) */
    float* dL_dmean2D,                /* (P,3) or NULL: view 0 */
    float* dL_dopacity, float* dL_dmean3D,
    float* dL_dsh,                    /* may be NULL if dL_dcolour_views is given */
    float* dL_dscale, float* dL_drot,
    float* dL_dcolour_views,          /* (nviews,P,3) or NULL: per-view dL/dcolour with the SH clamp mask applied
                                         (zero where the view does not see the Gaussian) */
    int debug, int flags, void* stream);

/*
 * The same backward for an iteration whose view 0 stands for TWO renders of the reference (render #1 and render #2 of
 * an event iteration share their pose: the event cameras are read with the training cameras' extrinsics,
 * scene/dataset_readers.py:157).  dL_dpix[0] holds the SUM of the two renders' pixel gradients; the densification
 * statistics (train.py:145,317-320) need the screen-space mean gradient of render #1 ALONE:
 * dL_dpix_view0_stats (3,H,W) is that render's own pixel gradient, and dL_dmean2D receives the gradient it induces
 * (a second dL/dalpha chain beside the first in the tiles of view 0; every other output is that of
 * e3dgs_rasterize_backward_multi).  dL_dmean2D is required.
 */
int e3dgs_rasterize_backward_multi_stats(
    int nviews, int P, int D, int M, int num_rendered,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations,
    const float* const* viewmatrix, const float* const* projmatrix, const float* const* cam_pos,
    const float* tan_fovx, const float* tan_fovy,
    const int* radii,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix,             /* (nviews,3,H,W) */
    const float* dL_dpix_view0_stats, /* (3,H,W) */
    float* grad_acc,
    float* dL_dmean2D,                /* (P,3): view 0 under dL_dpix_view0_stats */
    float* dL_dopacity, float* dL_dmean3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
    float* dL_dcolour_views,
    int debug, int flags, void* stream);

/*
 * The same backward when the pixel gradient of some views is RANK 1 (ABI 16): dL/dC(pixel) = s(pixel) * w with one
 * weight vector w per view.  That is what every loss on a luminance produces: the two contrast renders of an event
 * iteration (differentialable_event_simu -> rgb_to_LUVscale: w = (0.4124, 0.35758, 0.1804), utils/loss_utils.py:24-28,
 * 234-249, train.py:159-176) and the --gray losses (rgb_to_grayscale: w = (0.299, 0.587, 0.114), :18-23,40-48,
 * train.py:213-223).  For a view v with bit v of rank1_mask set, only plane 0 of its (3,H,W) block of dL_dpix is read --
 * the scalar field s -- and rank1_weights[3 v .. 3 v + 2] (HOST array, nviews x 3) is w.  The compositing backward then
 * carries one colour chain and seven sums per (pixel, entry) instead of three and nine; outputs as for
 * e3dgs_rasterize_backward_multi (equal to the general call on s * w up to fp32 summation order).
 * dL_dpix_view0_stats may be NULL (then dL_dmean2D follows dL_dpix as in e3dgs_rasterize_backward_multi); with it, view 0
 * carries the second gradient chain and cannot be rank 1.
 */
int e3dgs_rasterize_backward_multi_rank1(
    int nviews, int P, int D, int M, int num_rendered,
    const float* background, int width, int height,
    const float* means3D, const float* shs, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations,
    const float* const* viewmatrix, const float* const* projmatrix, const float* const* cam_pos,
    const float* tan_fovx, const float* tan_fovy,
    const int* radii,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix,             /* (nviews,3,H,W); rank-1 views: plane 0 = s, planes 1-2 not read */
    const float* dL_dpix_view0_stats, /* (3,H,W) or NULL */
    const float* rank1_weights,       /* host (nviews,3) */
    unsigned rank1_mask,
    float* grad_acc,
    float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
    float* dL_dcolour_views,
    int debug, int flags, void* stream);

/*
 * SH gradient from per-view colour gradients:  dL/dsh[k][ch] = scale * sum over all views of
 * Y_k(direction from the view's camera centre to the Gaussian) * dL/dcolour_view[ch].
 * New capability for view-parallel data parallelism (north_star: train.py's loop sharded by camera with an
 * all-reduce of the Gaussian gradients): the SH coefficients are 48 of the 59 gradient floats per Gaussian, but
 * their gradient is determined by 3 floats per (Gaussian, view).  Instead of all-reducing 48 floats per Gaussian the
 * ranks all-gather their (views_per_rank x P x 3) colour gradients + camera centres and each rank rebuilds the mean
 * SH gradient here (scale = 1 / nranks): 9 instead of 48 floats per Gaussian and rank for an event iteration.
 * `packed`: nranks blocks, `rank_stride` floats apart, each [views_per_rank*P*3 colour gradients as written by
 * e3dgs_rasterize_backward_multi(dL_dcolour_views) | views_per_rank*3 camera centres].  dL_dsh (P,M,3) or, with
 * E3DGS_FLAG_SH_PLANAR, (M*3,P); written in full (zeros above degree D).
 */
int e3dgs_sh_grad_from_colour(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                              const float* packed, size_t rank_stride, float scale, float* dL_dsh, int flags,
                              void* stream);

/*
 * The same rebuild fused with the optimizer step of the SH coefficients (train.py:330-332 for the f_dc / f_rest groups of
 * scene/gaussian_model.py:156-157): the rebuilt gradient is consumed in registers by torch.optim.Adam's update, `sh`
 * (layout per flags, (P,M,3) or coefficient-major) is updated IN PLACE together with its exp_avg / exp_avg_sq (same
 * layout).  One streaming pass that reads 3 colour-gradient floats per (Gaussian, view) instead of writing and
 * re-reading 48 gradient floats per Gaussian -- used on ONE rank too (nranks = 1): at 1 M Gaussians the iteration
 * sheds 0.3 GB of HBM traffic.  Bit-identical to e3dgs_sh_grad_from_colour + e3dgs_adam_step_groups.  `means3D`: the
 * positions the colour gradients were computed with (call it BEFORE the optimizer moves them).
 */
int e3dgs_sh_adam_from_colour(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                              const float* packed, size_t rank_stride, float scale, float* sh, float* exp_avg,
                              float* exp_avg_sq, float lr_f_dc, float lr_f_rest, float beta1, float beta2, float eps,
                              int step, int flags, void* stream);

/*
 * The same, PLUS the part of dL/dmean3D that E3DGS_FLAG_DEFER_SH_MEAN left out of e3dgs_rasterize_backward_multi: the
 * view-dependent SH colour makes the colour gradient act on the position through the unit view direction
 * (d colour / d dir x d dir / d mean, [UPSTREAM] computeColorFromSH backward), a term that needs all the coefficients --
 * a third of the per-Gaussian backward kernel's fetches -- while this kernel streams over those very coefficients with
 * the directions and colour gradients already in registers.  It ADDS the term to dL_dmean3D (P,3), view by view on top
 * of what the backward stored: the same operations in the same order as the undeferred backward, bit for bit.  At most 4
 * views in all (one rank's triplet).  Call it before the optimizer step of the positions.
 */
int e3dgs_sh_adam_from_colour_mean(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                                   const float* packed, size_t rank_stride, float scale, float* sh, float* exp_avg,
                                   float* exp_avg_sq, float lr_f_dc, float lr_f_rest, float beta1, float beta2, float eps,
                                   int step, int flags, float* dL_dmean3D, void* stream);

/*
 * DEPRECATED setters: they change the process-wide DEFAULT that calls without E3DGS_FLAG_OPTIONS fall back to (not
 * synchronised: set them before any call is in flight).  New code passes E3DGS_FLAG_OPTIONS | E3DGS_FLAG_CULL_RECT /
 * E3DGS_FLAG_NO_SMALL_PATHS per call.
 *
 * Exact tile culling (default ON; environment E3DGS_TILE_CULL=0 turns it off at load time).
 * The reference op bins every Gaussian into all tiles of its 3-sigma bounding rectangle.  With
 * culling ON, (tile, Gaussian) instances that provably reach no pixel of the tile with
 * alpha >= 1/255 are not emitted: image, radii and gradients are bit-identical, but
 * num_rendered, the sorted lists and n_contrib index a sub-sequence of the reference's lists.
 * Turn it OFF to reproduce the reference's integer binning exactly (used by the parity tests).
 */
void e3dgs_set_tile_cull(int on);
int e3dgs_get_tile_cull(void);

/*
 * Work decomposition for calls with few splats (a splat = one Gaussian under one view).  Right after a point-cloud
 * initialisation (scene/gaussian_model.py:124-147: thousands of Gaussians whose scale is the distance to their
 * neighbours) each splat covers hundreds of tiles; 64 splats per binning wave and one thread per splat in the
 * gradient-record reduction would then leave most of the chip idle.  ON (default; environment E3DGS_SMALL_SCENE_PATHS=0
 * turns it off): a binning wave owns fewer splats whenever that is needed for >= 8192 waves, and up to 262144 splats
 * the records of a splat are summed by a whole wave.  Lists, images and radii do not depend on the switch; gradients
 * agree to fp32 summation order.  OFF exists so that tests can run the large-scene kernels on small inputs.
 */
void e3dgs_set_small_scene_paths(int on);
int e3dgs_get_small_scene_paths(void);

/*
 * Byte offsets of the members of the three scratch buffers, for tests and tools that want to
 * inspect intermediate state (sorted lists, tile ranges, per-pixel n_contrib).
 *   out[0..4]  geom:    one 48-byte record per Gaussian (stride 12 floats): out[0] -> (x,y,conic.x,conic.y),
 *                       out[1] -> (conic.z,opacity,r,g), out[2] -> (b, strip-skip bound, -, -);
 *                       clamped (u32), rect (uint2 packed 16-bit xmin|ymin, xmax|ymax)
 *   out[5]     binning: perm (u32 EMISSION indices of the instances, tile-major, depth order); the Gaussian id of
 *                       emission index e is emit_gid[e], at byte offset e3dgs_state_offset_emit_gid(num_rendered)
 *   out[6..8]  image:   ranges (uint2 per tile), final_T (float per pixel), n_contrib (u32 per pixel)
 */
void e3dgs_state_offsets(int P, int num_rendered, int width, int height, size_t* out9);
/* the same for a call with nviews views (P Gaussians each): geometry members are indexed by splat q = i * nviews + v,
 * out[6] -> ranges of the nviews * tiles tiles, out[7] / out[8] -> final_T / n_contrib as (nviews, H, W) planes */
void e3dgs_state_offsets_multi(int nviews, int P, int num_rendered, int width, int height, size_t* out9);
size_t e3dgs_state_offset_emit_gid(int num_rendered);
/* (ABI 17) two more members of the binning scratch, for tools: out[0] -> strip_mask (one byte per LIST POSITION: bit k =
 * the forward evaluated the entry on the tile's 16x4 pixel strip k), out[1] -> touched (one byte per SLOT: the compositing
 * backward writes a gradient record for this instance) */
void e3dgs_state_offsets_binning(int num_rendered, size_t* out2);

/*
 * present[i] = 1 iff Gaussian i passes the near-plane test of the forward.
 * Replaces: diff_gaussian_rasterization._C.mark_visible
 * (GaussianRasterizer.markVisible; not called by this reference, kept for API
 * completeness).
 */
int e3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, uint8_t* present, void* stream);

/*
 * out[i] = mean squared distance from point i to its 3 nearest other points.
 * Replaces: simple_knn._C.distCUDA2 (scene/gaussian_model.py:134).
 * `scratch` must hold e3dgs_knn_scratch_bytes(P) bytes.
 */
size_t e3dgs_knn_scratch_bytes(int P);
int e3dgs_dist_knn3(int P, const float* points /* (P,3) */, float* out /* (P) */,
                    char* scratch, void* stream);

/*
 * Fused event iteration loss of train.py:165-203 (forward value and the
 * gradients w.r.t. the three rendered images and the contrast threshold c).
 *
 *   D  = (ln(Y(next)+1e-8) - ln(Y(now)+1e-8)) / c        utils/loss_utils.py:234-249
 *   D* = same on the ground-truth pair with c = 0.17      train.py:170
 *   loss = 0.9*mean|D-D*|*rho + 0.1*mean|image-gt_int|*(1-rho),  rho = mean(D* != 0)
 *   deblur (gt_blur != NULL): loss = 0.5*loss + 0.5*mean|image-gt_blur|   train.py:197-203
 *
 * Two launches: a reduction pass (partials -> scalars[8] on device) and a
 * gradient pass.  scalars_out[0]=loss, [1]=dL/dc, [2]=rho, [3]=L1 event,
 * [4]=L1 intensity, [5]=L1 blur.  `scratch` needs e3dgs_event_loss_scratch_bytes(W,H).
 *
 * Shared pose: the reference reads its event cameras with the training cameras' extrinsics
 * (scene/dataset_readers.py:157), so render #1 (train.py:144) and render #2 (:159) of an iteration are the same
 * render.  A caller that rendered it once passes the same pointer as `image` and `img_now` and the same pointer as
 * d_image and d_now: the SUM of the two gradients is stored there.  With the same input pointer but separate outputs,
 * d_now receives the sum (the shared render's total pixel gradient) and d_image the intensity term's part alone (the
 * second argument of e3dgs_rasterize_backward_multi_stats).
 */
size_t e3dgs_event_loss_scratch_bytes(int width, int height);
int e3dgs_event_loss(
    int width, int height,
    const float* image,     /* (3,H,W) intensity render */
    const float* img_now,   /* (3,H,W) */
    const float* img_next,  /* (3,H,W) */
    const float* gt_int,    /* (3,H,W) */
    const float* gt_now,    /* (3,H,W) */
    const float* gt_next,   /* (3,H,W) */
    const float* gt_blur,   /* (3,H,W) or NULL */
    const float* c,         /* (1) device scalar, learnable threshold */
    float gt_c,             /* 0.17 in the reference */
    float* d_image, float* d_now, float* d_next, /* (3,H,W) grads, overwritten */
    float* scalars_out,     /* (8) device */
    float* dc_out,          /* NULL, or a device word that also receives dL/dc (e.g. the threshold's slot of a flat
                               gradient buffer: saves the caller a copy kernel per iteration) */
    char* scratch,
    void* stream);
/*
 * The same loss in ONE sweep over the images for a ground-truth pair the caller has met before (ABI 14).  The gradient
 * pass needs two numbers of the reduction pass, and both depend on rho = count(D* != 0) / (W H) alone -- a property of
 * (gt_now, gt_next), not of the renders (utils/loss_utils.py:234-249 on the ground-truth pair, train.py:170-176).
 * `nz_count` is one device double owned by the caller, one per ground-truth pair:
 *   nz_valid == 0: the three launches of e3dgs_event_loss; *nz_count receives the pair's count;
 *   nz_valid != 0: *nz_count is read; partial sums and the three pixel gradients come out of one pass (every plane read
 *                  once), followed by the scalars.  Same arithmetic in the same order: bit-identical outputs.
 * A caller that changes a ground-truth frame in place starts over with nz_valid = 0.
 */
int e3dgs_event_loss_cached(
    int width, int height, const float* image, const float* img_now, const float* img_next, const float* gt_int,
    const float* gt_now, const float* gt_next, const float* gt_blur, const float* c, float gt_c, float* d_image,
    float* d_now, float* d_next, float* scalars_out, float* dc_out,
    double* nz_count,       /* (1) device: count of pixels with a non-zero contrast target of this ground-truth pair */
    int nz_valid,
    char* scratch, void* stream);

/*
 * e3dgs_event_loss / e3dgs_event_loss_cached (nz_count may be NULL: the three-launch form) with the contrast renders'
 * gradients in RANK-1 form (ABI 16), the input of e3dgs_rasterize_backward_multi_rank1: d_next -- and d_now when it is a
 * render of its own (img_now != image) -- receive in their FIRST plane the scalar field s with
 * dL/dC = s * (0.4124, 0.35758, 0.1804); their other two planes are not written.  d_image (and a d_now that carries the
 * shared render's total) stay full (3,H,W) gradients.  Loss scalars and dL/dc are those of e3dgs_event_loss, bit for bit.
 */
int e3dgs_event_loss_rank1(
    int width, int height, const float* image, const float* img_now, const float* img_next, const float* gt_int,
    const float* gt_now, const float* gt_next, const float* gt_blur, const float* c, float gt_c, float* d_image,
    float* d_now, float* d_next, float* scalars_out, float* dc_out, double* nz_count, int nz_valid, char* scratch,
    void* stream);

/*
 * Mean SSIM of two (C,H,W) images and (optionally) its gradient w.r.t. img1.
 * Replaces the torch conv2d chain of utils/loss_utils.py:359-418 (`ssim`, `_ssim`, `create_window`): 11x11
 * Gaussian window sigma 1.5, zero padding 5, C1 = 1e-4, C2 = 9e-4, mean over the map.  to_gray = 1 first applies
 * rgb_to_grayscale (:18-23) to both 3-channel inputs (`ssim_gray` :368-385; used by the --gray loss train.py:213-223
 * and by eval.py:146).  ssim_mean is a device scalar; d_img1 (C,H,W) may be NULL (forward only).
 */
size_t e3dgs_ssim_scratch_bytes(int channels, int height, int width);
int e3dgs_ssim(int channels, int height, int width, int to_gray, const float* img1, const float* img2,
               float* ssim_mean, float* d_img1, char* scratch, void* stream);

/*
 * Densification statistics of one iteration (train.py:317-320, scene/gaussian_model.py:405-407), in place, for the
 * Gaussians with radii > 0: max_radii2D (P) = max(., radii); xyz_gradient_accum (P) += |viewspace_grad[:, :2]|
 * (viewspace_grad (P,3) = the NDC-unit screen-space mean gradient of render #1); denom (P) += 1.
 */
int e3dgs_densify_stats_update(int P, const float* viewspace_grad, const int* radii, float* max_radii2D,
                               float* xyz_gradient_accum, float* denom, void* stream);

/*
 * Loss of the one-render iterations, fused: scalars[0] = (1 - lambda) L1 + lambda (1 - SSIM), [1] = L1, [2] = SSIM and
 * d_image (C,H,W) = its gradient w.r.t. `image`, in three launches.  to_gray = 0: train.py:292-296 (l1_loss
 * utils/loss_utils.py:270-271 + ssim :388-396 per channel); to_gray = 1: train.py:213-223 (l1_loss_gray :40-48 and
 * ssim_gray :368-385 on rgb_to_grayscale :18-23 of both 3-channel images).  scalars: 4 device floats.
 */
size_t e3dgs_image_loss_scratch_bytes(int channels, int height, int width);
int e3dgs_image_loss(int channels, int height, int width, int to_gray, float lambda_dssim, const float* image,
                     const float* gt_image, float* scalars, float* d_image, char* scratch, void* stream);
/* The --gray loss (to_gray = 1, 3-channel images) with its image gradient in rank-1 form (ABI 16): d_gray (H,W) receives
 * the scalar field s with d loss / d image = s * (0.299, 0.587, 0.114) -- the input of
 * e3dgs_rasterize_backward_multi_rank1.  Same scalars as e3dgs_image_loss; scratch of e3dgs_image_loss_scratch_bytes(3,H,W). */
int e3dgs_image_loss_rank1(int height, int width, float lambda_dssim, const float* image, const float* gt_image,
                           float* scalars, float* d_gray, char* scratch, void* stream);


/*
 * Fused Adam step over one flat parameter tensor (train.py:330-332; groups
 * scene/gaussian_model.py:154-163; eps 1e-15).  Matches torch.optim.Adam
 * (no amsgrad, no weight decay): bias-corrected with step count `step`.
 * beta1 / beta2 cross this boundary as fp32 while torch derives 1 - beta and beta^step from the Python double the
 * caller wrote; every Adam entry point of this header therefore takes the 7-digit decimal nearest to the fp32 value
 * WHEN that decimal converts back to the same fp32 value and lies in [0, 1) (0.9f -> 0.9, 0.999f -> 0.999), and the
 * fp32 value itself otherwise (a scheduled beta, 1 - 1/k, a beta within 5e-8 of 1).
 */
int e3dgs_adam_step(size_t n, float* param, const float* grad, float* exp_avg,
                    float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                    int step,
                    float lr_b, int period, int split, /* period > 0: element i uses lr if (i % period) < split else lr_b
                                                          (f_dc / f_rest groups interleaved in one (P,16,3) tensor) */
                    void* stream);

/*
 * The same step for SEVERAL parameter groups laid out back to back in one flat buffer, in one launch: segment k covers
 * elements [seg_end[k-1], seg_end[k]) (seg_end[-1] = 0, seg_end[nseg-1] = n) and uses lr[k] / eps[k]; nseg <= 8.
 * seg_end / lr / eps are HOST arrays.  (The optimizer of scene/gaussian_model.py:154-163 has six groups, train.py:71-73
 * a seventh for the contrast threshold.)
 */
int e3dgs_adam_step_segments(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                             const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2, int step,
                             void* stream);

/*
 * As above with ONE STEP COUNT PER GROUP (host array steps[nseg]).  torch.optim.Adam keeps `step` per parameter and skips
 * parameters whose .grad is None: in the reference that happens to all six Gaussian groups on every densification
 * iteration (train.py:317-332 replaces the parameters before optimizer.step()) and to the opacity group on reset
 * iterations (scene/gaussian_model.py:210-213,258-271), so the groups' bias corrections drift apart.  steps[k] <= 0
 * skips group k: parameter and both moments stay untouched.
 */
int e3dgs_adam_step_groups(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                           const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2,
                           const int* steps, void* stream);

/*
 * As e3dgs_adam_step_groups, with a GAP: the elements [gap_begin, gap_begin + gap_len) of the buffer are not visited at all
 * (a skipped segment still costs its grid-stride iterations; the gap costs nothing -- the launch is sized for the n -
 * gap_len elements around it).  The trainer's flat buffer is xyz | SH | opacity | scaling | rotation | c and its SH
 * coefficients are stepped by e3dgs_sh_adam_from_colour: everything around them (scene/gaussian_model.py:154-163 groups
 * xyz, opacity, scaling, rotation + train.py:71-73 `c`) is ONE launch instead of one per contiguous range.  seg_end are
 * offsets in the WHOLE buffer (segments may contain the gap).  Same arithmetic per element: bit-identical.
 */
int e3dgs_adam_step_groups_gap(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                               const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2,
                               const int* steps, size_t gap_begin, size_t gap_len, void* stream);

/*
 * Adaptive density control on the device: GaussianModel.densify_and_prune (scene/gaussian_model.py:389-403 = clone
 * :374-387 + split :349-372 + prune :273-305,396-402; schedule train.py:317-327) as ONE plan pass and ONE apply pass over
 * the flat training buffers -- parameters and both Adam moments compacted together, clones and split children appended
 * in the reference's row order [kept originals | kept clones | kept first children | kept second children], new rows
 * with zero moments.  Flat buffer of P Gaussians (floats): xyz 3P | SH coefficient-major (48, P) | opacity P |
 * scaling 3P | rotation 4P | c 1  (pre-activation values, scene/gaussian_model.py:44-59).
 *
 *   plan:  decisions + exclusive scans; ONE host wait; counts_host4 = {kept originals, kept clones, rows selected for
 *          the split, split rows whose children are kept}.  New size = counts[0] + counts[1] + 2 * counts[3].
 *          size_prune != 0 applies the world-space size test (max scale > 0.1 extent, :399); the screen-space half of
 *          that test reads max_radii2D AFTER the postfix zeroed it (:347) and can never fire, as in the reference.
 *   split_rows: device pointer (inside `scratch`) to the int32 indices of the counts[2] selected rows, ascending --
 *          the caller draws the split offsets for them (torch.normal(0, exp(scaling[rows]).repeat(2, 1)), :358-360) so
 *          that the random stream stays torch's.
 *   apply: writes the three new flat buffers (P_new Gaussians).  samples: (2 * counts[2], 3) float32, row j and
 *          counts[2] + j belong to the j-th selected row.
 */
size_t e3dgs_densify_scratch_bytes(int P);
int e3dgs_densify_plan(int P, const float* flat_param, const float* xyz_gradient_accum, const float* denom,
                       float max_grad, float min_opacity, float extent, float percent_dense, int size_prune, char* scratch,
                       int* counts_host4, void* stream);
const int* e3dgs_densify_split_rows(int P, char* scratch);
int e3dgs_densify_apply(int P, int P_new, const int* counts4, const float* param, const float* exp_avg,
                        const float* exp_avg_sq, const float* samples, float* param_new, float* exp_avg_new,
                        float* exp_avg_sq_new, char* scratch, void* stream);

/*
 * The rasteriser's sort, callable on its own: stable LSD radix sort of n (key, value) pairs on key bits [0, nbits).
 * Replaces: cub::DeviceRadixSort::SortPairs of the reference op's binning stage ([UPSTREAM] rasterizer_impl.cu, SURVEY 2.1
 * row "SortPairs"; the reference sorts 64-bit (tile << 32 | depth) keys, this library a 32-bit depth sort of the splats
 * followed by a 16-/32-bit tile-id sort of the instances -- DESIGN.md section 4, decision 1).
 *   key_bytes        4 (uint32 keys) or 2 (uint16 keys); keys0/keys1 and vals0/vals1 are ping-pong buffers of n elements,
 *                    the input sits in (keys0, vals0); *result_index_host = 0 or 1 says which pair holds the result.
 *   identity_payload != 0: vals0 is not read, the payload of element i is i.
 *   kept_count_dev   (32-bit keys, may be NULL) keys equal to 0xFFFFFFFF are DROPPED; the number of kept pairs is stored
 *                    there (device word) and only that many result elements are valid.
 *   ranges           (may be NULL) uint32 (nranges, 2), zero on entry: ranges[k] = [first, last + 1) positions of key k in
 *                    the result; keys without an element keep an empty range (start == end).  A two-pass sort (9..16
 *                    key bits) derives the ranges inside its last pass and does NOT write the sorted keys.
 *   scratch          e3dgs_sort_scratch_bytes(n) bytes, 8-byte aligned.
 */
size_t e3dgs_sort_scratch_bytes(size_t n);
int e3dgs_sort_pairs(size_t n, int nbits, int key_bytes, void* keys0, void* keys1, uint32_t* vals0, uint32_t* vals1,
                     int identity_payload, char* scratch, uint32_t* kept_count_dev, uint32_t* ranges, uint32_t nranges,
                     int* result_index_host, void* stream);
/*
 * The depth sort of the splats on its own (ABI 17): stable sort of n 32-bit keys on ALL 32 bits in three passes of
 * 11 + 11 + 10 bits (2048-bin histograms in LDS) with the identity payload -- order[j] = index of the j-th smallest key,
 * ties in index order; keys equal to 0xFFFFFFFF (splats the projection culled) are dropped and the kept count is stored
 * in *kept_count_dev (device word, required).  keys0 holds the keys on entry; keys0 / keys1 / order_alt are clobbered and
 * the sorted keys are not produced.  scratch: e3dgs_depth_sort_scratch_bytes(n) bytes, 8-byte aligned.
 * Replaces the depth half of cub::DeviceRadixSort::SortPairs ([UPSTREAM] rasterizer_impl.cu; DESIGN.md decision 51).
 */
size_t e3dgs_depth_sort_scratch_bytes(size_t n);
int e3dgs_sort_depth_keys(size_t n, uint32_t* keys0, uint32_t* keys1, uint32_t* order, uint32_t* order_alt, char* scratch,
                          uint32_t* kept_count_dev, void* stream);

/*
 * Kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * Slots: 0 preprocess, 1 sort_depth, 2 scan_emit, 3 sort_tile, 4 tile_ranges, 5 render_fwd,
 *        6 render_bwd, 7 geom_bwd.  enable(mask) resets the counters and times the slots whose bit is set
 * (0 = off, 0xFF = all; every timed slot costs two event packets per launch group, so a benchmark times only
 * the kernel it reports inside its timed region); query() synchronises the recorded events and returns the
 * accumulated milliseconds and the number of launches.  The profiler's state is PROCESS-WIDE (behind a mutex): calls made by
 * any host thread while a slot is enabled are timed -- including the backward calls torch's autograd engine makes from its
 * per-device worker thread under loss.backward() -- and a slot's totals are the sum over all threads.
 * select(mask) changes the set of timed slots WITHOUT resetting what has been recorded (a benchmark that brackets its
 * kernel on every n-th iteration only: the two event packets cost the launch ~6 us of GPU idle each).
 */
void e3dgs_profile_enable(int slot_mask);
void e3dgs_profile_select(int slot_mask);
int e3dgs_profile_query(int slot, double* total_ms, int* launches);
const char* e3dgs_profile_slot_name(int slot);

#ifdef __cplusplus
}
#endif
#endif /* E3DGS_HIP_H */
