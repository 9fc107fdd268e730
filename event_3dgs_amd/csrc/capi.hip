// capi.hip -- the extern "C" boundary declared in include/e3dgs_hip.h.
#include "common.h"
#include "../../include/e3dgs_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <unordered_map>

thread_local char g_err[512] = "";

int e3_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof g_err, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    return e == hipSuccess ? -1 : (int)e;
}

// exact tile culling (forward.hip: tile_touched); E3DGS_TILE_CULL=0 in the environment disables it
int g_tile_cull = [] { const char* e = getenv("E3DGS_TILE_CULL"); return (e && e[0] == '0') ? 0 : (e && e[0] == '3') ? 3 : 1; }();
int g_small_scene_paths = [] { const char* e = getenv("E3DGS_SMALL_SCENE_PATHS"); return (e && e[0] == '0') ? 0 : 1; }();

// The options of a call: its own flags word when E3_FLAG_OPTIONS is set -- nothing process-global is consulted, so
// calls with different settings may run concurrently from any number of threads and streams -- and the process-wide
// defaults above otherwise (environment at load time; the e3dgs_set_* setters are deprecated shims that change them).
CallOpts e3_call_opts(int flags) {
    CallOpts o;
    if (flags & E3_FLAG_OPTIONS) {
        o.cull = (flags & E3_FLAG_CULL_RECT) ? 0 : ((flags & E3_FLAG_CULL_NO_BOX) ? 3 : 1);
        o.small_paths = (flags & E3_FLAG_NO_SMALL_PATHS) ? 0 : 1;
    } else {
        o.cull = g_tile_cull;
        o.small_paths = g_small_scene_paths;
    }
    o.fast_exp = (flags & E3_FLAG_FAST_EXP) ? 1 : 0;
    return o;
}

// ---- the resolved options of the last `begin` on a geometry scratch (include/e3dgs_hip.h, "per-call OPTIONS").  Host
// memory only: a device read-back would cost a stream synchronisation per half.  Keyed by the scratch address the
// caller's allocator returned; a caller that MOVES the scratch between the halves is not checked (no entry: accepted).
// The table is validation state, not behaviour: no call reads its settings from it.
namespace {
struct GeomOpts { size_t Q; int cull, small_paths, fast_exp; };
std::mutex g_opts_mu;
std::unordered_map<const void*, GeomOpts> g_opts;
constexpr size_t OPTS_MAX = 1 << 14;          // addresses are recycled by every allocator; beyond this: start over
}  // namespace
void e3_geom_opts_remember(const void* geom, size_t Q, const CallOpts& o) {
    if (!geom) return;
    std::lock_guard<std::mutex> lock(g_opts_mu);
    if (g_opts.size() >= OPTS_MAX && !g_opts.count(geom)) g_opts.clear();
    g_opts[geom] = GeomOpts{Q, o.cull, o.small_paths, o.fast_exp};
}
int e3_geom_opts_check(const void* geom, size_t Q, const CallOpts& o, const char* who) {
    GeomOpts g;
    {
        std::lock_guard<std::mutex> lock(g_opts_mu);
        auto it = g_opts.find(geom);
        if (it == g_opts.end()) return 0;
        g = it->second;
    }
    if (g.Q == Q && g.cull == o.cull && g.small_paths == o.small_paths && g.fast_exp == o.fast_exp) return 0;
    snprintf(g_err, sizeof g_err,
             "%s: option bits / sizes differ from the ones `begin` resolved for this geometry scratch (begin: P*nviews %zu, "
             "cull %d, small_paths %d, fast_exp %d; this call: %zu, %d, %d, %d) -- nothing was launched (hipError %d)",
             who, g.Q, g.cull, g.small_paths, g.fast_exp, Q, o.cull, o.small_paths, o.fast_exp, (int)hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
}

// ---- event profiler.  PROCESS-WIDE state behind a mutex: torch runs the autograd nodes of a backward pass on its own
// per-device worker thread, so a profiler enabled on the caller's thread must also time the kernels that thread
// launches (round 4 kept the state per host thread and silently dropped render_bwd / geom_bwd under loss.backward()).
// A scope reserves its event pair under the lock and closes exactly that pair, so scopes opened concurrently by several
// threads (two trainers on two streams) cannot mix their events; the totals of a slot are then the sum over all threads.
std::atomic<unsigned> g_prof_mask{0};
namespace {
struct ProfPair { hipEvent_t a, b; bool closed; };
constexpr int PROF_MAX = 4096;
struct ProfState {
    ProfPair pairs[PS_COUNT][PROF_MAX];
    int created[PS_COUNT] = {0}, used[PS_COUNT] = {0};
};
std::mutex g_prof_mu;
ProfState* g_ps = nullptr;
const char* g_names[PS_COUNT] = {"preprocess", "sort_depth", "scan_emit", "sort_tile", "tile_ranges", "render_fwd",
                                 "render_bwd", "geom_bwd"};
}  // namespace
int prof_begin(int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    ProfState* P = g_ps;
    if (!P) return -1;
    const int k = P->used[slot];
    if (k >= PROF_MAX) return -1;
    if (k >= P->created[slot]) {
        if (hipEventCreate(&P->pairs[slot][k].a) != hipSuccess) return -1;
        if (hipEventCreate(&P->pairs[slot][k].b) != hipSuccess) { (void)hipEventDestroy(P->pairs[slot][k].a); return -1; }
        P->created[slot] = k + 1;
    }
    P->pairs[slot][k].closed = false;
    if (hipEventRecord(P->pairs[slot][k].a, s) != hipSuccess) return -1;
    P->used[slot] = k + 1;
    return k;
}
void prof_end(int slot, int k, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    ProfState* P = g_ps;
    if (!P || k < 0 || k >= P->used[slot]) return;      // (the profiler was reset while the scope was open)
    if (hipEventRecord(P->pairs[slot][k].b, s) == hipSuccess) P->pairs[slot][k].closed = true;
}

int e3_mark_visible_impl(int, const float*, const float*, uint8_t*, hipStream_t);
size_t e3_knn_scratch_bytes(int);
int e3_knn_impl(int, const float*, float*, char*, hipStream_t);
size_t e3_event_scratch_bytes(int, int);
int e3_event_loss_impl(int, int, const float*, const float*, const float*, const float*, const float*, const float*,
                       const float*, const float*, float, float*, float*, float*, float*, char*, hipStream_t, float*, double*,
                       int, int rank1 = 0);
size_t e3_ssim_scratch_bytes(int, int, int);
int e3_ssim_impl(int, int, int, int, const float*, const float*, float*, float*, char*, hipStream_t);
int e3_adam_segments_impl(size_t, float*, const float*, float*, float*, int, const size_t*, const float*, const float*, float,
                          float, int, const int*, hipStream_t, size_t gap_begin = 0, size_t gap_len = 0);
int e3_densify_stats_impl(int, const float*, const int*, float*, float*, float*, hipStream_t);
size_t e3_image_loss_scratch_bytes(int, int, int);
int e3_image_loss_impl(int, int, int, int, float, const float*, const float*, float*, float*, char*, hipStream_t, int rank1 = 0);
int e3_adam_impl(size_t, float*, const float*, float*, float*, float, float, float, float, int, float, int, int,
                 hipStream_t);
size_t e3_densify_scratch_bytes(int);
int e3_densify_plan_impl(int, const float*, const float*, const float*, float, float, float, float, int, char*, int*,
                         hipStream_t);
int e3_densify_apply_impl(int, int, const int*, const float*, const float*, const float*, const float*, float*, float*,
                          float*, char*, hipStream_t);
const int* e3_densify_split_rows(int, char*);

extern "C" {

int e3dgs_abi_version(void) { return 17; }
const char* e3dgs_last_error(void) { return g_err; }

static ViewBatch one_view(const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                          float tan_fovy) {
    ViewBatch b;
    b.n = 1;
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        b.view[v] = viewmatrix; b.proj[v] = projmatrix; b.campos[v] = cam_pos;
        b.tanfovx[v] = tan_fovx; b.tanfovy[v] = tan_fovy;
    }
    return b;
}
static int make_batch(int nviews, int P, const float* const* viewmatrix, const float* const* projmatrix,
                      const float* const* cam_pos, const float* tan_fovx, const float* tan_fovy, ViewBatch& b) {
    if (nviews < 1 || nviews > E3_MAX_VIEWS) return e3_fail(hipErrorInvalidValue, "nviews must be 1..4");
    if (!viewmatrix || !projmatrix || !cam_pos || !tan_fovx || !tan_fovy)
        return e3_fail(hipErrorInvalidValue, "per-view arrays are required");
    if ((long long)P * nviews > 0x7FFFFFFFll) return e3_fail(hipErrorInvalidValue, "P * nviews exceeds 2^31-1");
    b.n = nviews;
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        const int u = v < nviews ? v : 0;
        if (!viewmatrix[u] || !projmatrix[u] || !cam_pos[u]) return e3_fail(hipErrorInvalidValue, "null per-view pointer");
        b.view[v] = viewmatrix[u]; b.proj[v] = projmatrix[u]; b.campos[v] = cam_pos[u];
        b.tanfovx[v] = tan_fovx[u]; b.tanfovy[v] = tan_fovy[u];
    }
    return 0;
}

// max_degree: 4 (utils/sh_utils.py:57-112 goes that far) for the operator and for the several-views-in-one-pass entry
// points; 3 where a stage keeps the reference model's 16 coefficients per channel (the deferred colour stage, the
// colour-gradient route of the multi-view backward and the SH optimizer kernels behind it: the trainer's layout)
static int check_forward_args(int P, int D, int M, int width, int height, const float* shs, const float* colors_precomp,
                              const float* scales, const float* rotations, const float* cov3D_precomp, int flags,
                              int max_degree = 4) {
    if (P < 0 || width <= 0 || height <= 0) return e3_fail(hipErrorInvalidValue, "bad sizes");
    if ((shs == nullptr) == (colors_precomp == nullptr) && P > 0)
        return e3_fail(hipErrorInvalidValue, "provide exactly one of shs / colors_precomp");
    if (((scales == nullptr || rotations == nullptr) == (cov3D_precomp == nullptr)) && P > 0)
        return e3_fail(hipErrorInvalidValue, "provide exactly one of scales+rotations / cov3D_precomp");
    if ((flags & E3_FLAG_PREACT) && cov3D_precomp) return e3_fail(hipErrorInvalidValue, "PREACT needs scales+rotations");
    if (shs && (D < 0 || D > max_degree || M < (D + 1) * (D + 1)))
        return e3_fail(hipErrorInvalidValue, max_degree == 4 ? "SH degree must be 0..4 and M >= (D+1)^2"
                                                             : "SH degree must be 0..3 and M >= (D+1)^2");
    if ((width + 15) / 16 > 65535 || (height + 15) / 16 > 65535)
        return e3_fail(hipErrorInvalidValue, "image too large for 16-bit tile coordinates");
    return 0;
}

static int check_backward_args(int nviews, int P, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* cov3D_precomp, float* grad_acc,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                               float* dL_dsh, float* dL_dscale, float* dL_drot, int flags) {
    if (P > 0 && !grad_acc) return e3_fail(hipErrorInvalidValue, "grad_acc is required");
    if ((flags & E3_FLAG_PREACT) && (cov3D_precomp || !opacities))
        return e3_fail(hipErrorInvalidValue, "PREACT needs scales+rotations and opacities");
    if (flags & E3_FLAG_BWD_ONLY_RENDER) return 0;      // the per-Gaussian stage writes the rest
    if (P > 0 && !dL_dmean3D) return e3_fail(hipErrorInvalidValue, "dL_dmean3D is required");
    if (shs && !dL_dsh) return e3_fail(hipErrorInvalidValue, "dL_dsh required when shs is given");
    if (!cov3D_precomp && (!dL_dscale || !dL_drot))
        return e3_fail(hipErrorInvalidValue, "dL_dscale/dL_drot required when scales+rotations are given");
    if (nviews > 1) {
        if (!shs || colors_precomp || cov3D_precomp || dL_dcolor || dL_dcov3D)
            return e3_fail(hipErrorInvalidValue, "multi-view backward needs shs + scales + rotations (no precomputed inputs)");
        if (!dL_dopacity) return e3_fail(hipErrorInvalidValue, "multi-view backward needs dL_dopacity");
        if (flags & E3_FLAG_ACCUMULATE)
            return e3_fail(hipErrorInvalidValue, "multi-view backward overwrites its outputs (no ACCUMULATE)");
    }
    return 0;
}

// one-call forward: begin + the single stream synchronisation + finish
static int forward_sync(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn binning_alloc, void* binning_user,
                        e3dgs_alloc_fn image_alloc, void* image_user, const ViewBatch& vb, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, float* out_color, int* radii, int debug,
                        int flags, int* num_rendered_host, void* stream) {
    struct Keep { e3dgs_alloc_fn fn; void* user; char* ptr; };
    // remember the two buffers handed out in `begin` so that `finish` can be fed without a second callback
    static thread_local Keep kg, ki;
    kg = Keep{geom_alloc, geom_user, nullptr};
    ki = Keep{image_alloc, image_user, nullptr};
    auto grab_g = [](void* u, size_t n) -> char* { Keep* k = (Keep*)u; k->ptr = k->fn(k->user, n); return k->ptr; };
    // The op's single device->host read-back: the instance count that sizes the binning buffers.  The GPU stores it
    // straight into a pinned, device-mapped word of this host thread and the host polls that word -- no copy command, no
    // runtime wake-up behind a stream synchronisation, during which the GPU has nothing queued (measured on the trainer's
    // path: 68 -> 26 us of GPU idle per forward; the reference's iteration calls the operator three times).
    // E3DGS_COUNT_POLL=0, debug calls and a failed pinned allocation take the stream synchronisation.
    // Semantics that differ from a stream synchronisation (documented in include/e3dgs_hip.h): the host returns as soon
    // as the count is there, NOT when the stream has drained, so an asynchronous error of earlier work on the stream does
    // not surface here (it surfaces at the caller's next synchronising call, as with any enqueue-only API).
    struct MappedWord {                 // one 64-byte pinned word per host thread, released when the thread exits
        volatile int* p = nullptr;
        bool tried = false;
        ~MappedWord() { if (p) (void)hipHostFree((void*)p); }
    };
    static thread_local MappedWord mw;
    static const bool poll_ok = !(getenv("E3DGS_COUNT_POLL") && getenv("E3DGS_COUNT_POLL")[0] == '0');
    if (!mw.tried && poll_ok) {
        mw.tried = true;
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) mw.p = (volatile int*)hp;
        else (void)hipGetLastError();
    }
    volatile int* const mapped = mw.p;
    const bool poll = mapped != nullptr && poll_ok && !debug && P > 0;
    int count = 0;
    if (poll) *mapped = -1;
    int rc = e3_forward_begin_impl(grab_g, &kg, grab_g, &ki, vb, P, D, M, width, height, means3D, shs, colors_precomp,
                                   opacities, scales, scale_modifier, rotations, cov3D_precomp, radii, debug,
                                   poll ? (flags | E3_FLAG_COUNT_MAPPED) : flags, poll ? (int*)mapped : &count,
                                   (hipStream_t)stream);
    if (rc) return rc;
    if (poll) {
        // poll with a cpu-relax between reads; every 4096 polls look at the clock: after 2 s without a count the wait
        // becomes a stream synchronisation (a hung or dead stream then reports its own error instead of spinning forever)
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*mapped == -1) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFu) == 0u &&
                std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                hipError_t q = hipStreamSynchronize((hipStream_t)stream);
                if (q != hipSuccess) return e3_fail(q, "hipStreamSynchronize (instance count, after 2 s of polling)");
                if (*mapped == -1) return e3_fail(hipErrorUnknown, "the instance count never arrived");
            }
        }
        count = *mapped;
    } else {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return e3_fail(e, "hipStreamSynchronize (instance count)");
    }
    *num_rendered_host = count;
    return e3_forward_finish_impl(binning_alloc, binning_user, vb.n, P, width, height, background, kg.ptr, ki.ptr,
                                  count, out_color, debug, flags, (hipStream_t)stream);
}

int e3dgs_rasterize_forward(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn binning_alloc,
                            void* binning_user, e3dgs_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                            const float* background, int width, int height, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            float scale_modifier, const float* rotations, const float* cov3D_precomp,
                            const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                            float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, int flags,
                            int* num_rendered_host, void* stream) {
    (void)prefiltered;
    g_err[0] = 0;
    int rc = check_forward_args(P, D, M, width, height, shs, colors_precomp, scales, rotations, cov3D_precomp, flags);
    if (rc) return rc;
    flags &= ~(E3_FLAG_DEFER_COLOR | E3_FLAG_COUNT_MAPPED);     // only meaningful for the split multi-view calls
    return forward_sync(geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user,
                        one_view(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy), P, D, M, background, width,
                        height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                        cov3D_precomp, out_color, radii, debug, flags, num_rendered_host, stream);
}

int e3dgs_rasterize_forward_begin(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn image_alloc,
                                  void* image_user, int P, int D, int M, int width, int height, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* opacities,
                                  const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* cam_pos, float tan_fovx, float tan_fovy, int* radii, int debug, int flags,
                                  int* num_rendered_host, void* stream) {
    g_err[0] = 0;
    int rc = check_forward_args(P, D, M, width, height, shs, colors_precomp, scales, rotations, cov3D_precomp, flags);
    if (rc) return rc;
    flags &= ~E3_FLAG_DEFER_COLOR;                             // the single-view finish has no colour stage
    return e3_forward_begin_impl(geom_alloc, geom_user, image_alloc, image_user,
                                 one_view(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy), P, D, M, width, height,
                                 means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                 cov3D_precomp, radii, debug, flags, num_rendered_host, (hipStream_t)stream);
}

int e3dgs_rasterize_forward_finish(e3dgs_alloc_fn binning_alloc, void* binning_user, int P, int width, int height,
                                   const float* background, char* geom_buffer, char* image_buffer, int num_rendered,
                                   float* out_color, int debug, int flags, void* stream) {
    g_err[0] = 0;
    if (P < 0 || num_rendered < 0 || !geom_buffer || !image_buffer) return e3_fail(hipErrorInvalidValue, "bad arguments");
    return e3_forward_finish_impl(binning_alloc, binning_user, 1, P, width, height, background, geom_buffer,
                                  image_buffer, num_rendered, out_color, debug, flags, (hipStream_t)stream);
}

int e3dgs_rasterize_backward(int P, int D, int M, int num_rendered, const float* background, int width, int height,
                             const float* means3D, const float* shs, const float* colors_precomp,
                             const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                             const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                             const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                             const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                             const float* dL_dpix, float* grad_acc, float* dL_dmean2D, float* dL_dopacity,
                             float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                             float* dL_drot, int debug, int flags, void* stream) {
    g_err[0] = 0;
    int rc = check_backward_args(1, P, shs, colors_precomp, opacities, cov3D_precomp, grad_acc, dL_dopacity, dL_dcolor,
                                 dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, flags);
    if (rc) return rc;
    return e3_backward_impl(one_view(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy), P, D, M, num_rendered,
                            background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                            rotations, cov3D_precomp, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix,
                            grad_acc, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                            dL_drot, debug, flags, (hipStream_t)stream);
}

// ---- several views of the same Gaussians in one pass (see ViewSet in common.h)
int e3dgs_rasterize_forward_multi(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn binning_alloc,
                                  void* binning_user, e3dgs_alloc_fn image_alloc, void* image_user, int nviews, int P,
                                  int D, int M, const float* background, int width, int height, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* opacities,
                                  const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* const* viewmatrix,
                                  const float* const* projmatrix, const float* const* cam_pos, const float* tan_fovx,
                                  const float* tan_fovy, float* out_color, int* radii, int debug, int flags,
                                  int* num_rendered_host, void* stream) {
    g_err[0] = 0;
    int rc = check_forward_args(P, D, M, width, height, shs, colors_precomp, scales, rotations, cov3D_precomp, flags, 4);
    if (rc) return rc;
    ViewBatch vb;
    rc = make_batch(nviews, P, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, vb);
    if (rc) return rc;
    flags &= ~(E3_FLAG_DEFER_COLOR | E3_FLAG_COUNT_MAPPED);
    return forward_sync(geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user, vb, P, D, M,
                        background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                        rotations, cov3D_precomp, out_color, radii, debug, flags, num_rendered_host, stream);
}

int e3dgs_rasterize_forward_multi_begin(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn image_alloc,
                                        void* image_user, int nviews, int P, int D, int M, int width, int height,
                                        const float* means3D, const float* shs, const float* colors_precomp,
                                        const float* opacities, const float* scales, float scale_modifier,
                                        const float* rotations, const float* cov3D_precomp,
                                        const float* const* viewmatrix, const float* const* projmatrix,
                                        const float* const* cam_pos, const float* tan_fovx, const float* tan_fovy,
                                        int* radii, int debug, int flags, int* num_rendered_host, void* stream) {
    g_err[0] = 0;
    // (degree 4 -- 25 coefficients -- is served by the same projection kernel; the separate colour stage of
    // E3DGS_FLAG_DEFER_COLOR and the colour-gradient route of the backward keep the reference model's 16)
    int rc = check_forward_args(P, D, M, width, height, shs, colors_precomp, scales, rotations, cov3D_precomp, flags,
                                (flags & E3_FLAG_DEFER_COLOR) ? 3 : 4);
    if (rc) return rc;
    ViewBatch vb;
    rc = make_batch(nviews, P, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, vb);
    if (rc) return rc;
    return e3_forward_begin_impl(geom_alloc, geom_user, image_alloc, image_user, vb, P, D, M, width, height, means3D,
                                 shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, radii,
                                 debug, flags, num_rendered_host, (hipStream_t)stream);
}

int e3dgs_rasterize_forward_multi_finish(e3dgs_alloc_fn binning_alloc, void* binning_user, int nviews, int P, int width,
                                         int height, const float* background, char* geom_buffer, char* image_buffer,
                                         int num_rendered, float* out_color, int debug, int flags, void* stream) {
    g_err[0] = 0;
    if (nviews < 1 || nviews > E3_MAX_VIEWS || P < 0 || num_rendered < 0 || !geom_buffer || !image_buffer)
        return e3_fail(hipErrorInvalidValue, "bad arguments");
    return e3_forward_finish_impl(binning_alloc, binning_user, nviews, P, width, height, background, geom_buffer,
                                  image_buffer, num_rendered, out_color, debug, flags, (hipStream_t)stream);
}

int e3dgs_rasterize_forward_multi_finish_colour(e3dgs_alloc_fn binning_alloc, void* binning_user, int nviews, int P,
                                                int width, int height, const float* background, char* geom_buffer,
                                                char* image_buffer, int num_rendered, float* out_color, int debug,
                                                int D, int M, const float* means3D, const float* shs,
                                                const float* const* cam_pos, int flags,
                                                e3dgs_notify_fn before_colour, void* notify_user, void* stream) {
    g_err[0] = 0;
    if (nviews < 1 || nviews > E3_MAX_VIEWS || P < 0 || num_rendered < 0 || !geom_buffer || !image_buffer)
        return e3_fail(hipErrorInvalidValue, "bad arguments");
    if (P > 0 && (!means3D || !shs || !cam_pos)) return e3_fail(hipErrorInvalidValue, "means3D, shs and cam_pos are required");
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1)) return e3_fail(hipErrorInvalidValue, "SH degree must be 0..3 and M >= (D+1)^2");
    DeferredColour dc;
    dc.views.n = nviews;
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        const int u = v < nviews ? v : 0;
        if (P > 0 && !cam_pos[u]) return e3_fail(hipErrorInvalidValue, "null per-view pointer");
        dc.views.view[v] = nullptr; dc.views.proj[v] = nullptr; dc.views.campos[v] = P > 0 ? cam_pos[u] : nullptr;
        dc.views.tanfovx[v] = 1.0f; dc.views.tanfovy[v] = 1.0f;        // colour only needs the camera position
    }
    dc.D = D; dc.M = M; dc.flags = flags; dc.means3D = means3D; dc.shs = shs;
    dc.before = before_colour; dc.user = notify_user;
    return e3_forward_finish_impl(binning_alloc, binning_user, nviews, P, width, height, background, geom_buffer,
                                  image_buffer, num_rendered, out_color, debug, flags, (hipStream_t)stream, &dc);
}

// ---- the same multi-view forward with NO host wait: binning buffers sized by the caller before the count is known
int e3dgs_rasterize_forward_multi_capacity(e3dgs_alloc_fn geom_alloc, void* geom_user, e3dgs_alloc_fn binning_alloc,
                                           void* binning_user, e3dgs_alloc_fn image_alloc, void* image_user, int nviews,
                                           int P, int D, int M, const float* background, int width, int height,
                                           const float* means3D, const float* shs, const float* opacities,
                                           const float* scales, float scale_modifier, const float* rotations,
                                           const float* const* viewmatrix, const float* const* projmatrix,
                                           const float* const* cam_pos, const float* tan_fovx, const float* tan_fovy,
                                           float* out_color, int* radii, int debug, int flags, int capacity,
                                           int* num_rendered_host, e3dgs_notify_fn before_colour, void* notify_user,
                                           void* stream) {
    g_err[0] = 0;
    int rc = check_forward_args(P, D, M, width, height, shs, nullptr, scales, rotations, nullptr, flags,
                                (flags & E3_FLAG_DEFER_COLOR) ? 3 : 4);
    if (rc) return rc;
    if (capacity < 1) return e3_fail(hipErrorInvalidValue, "capacity must be positive");
    if (!num_rendered_host || !(flags & E3_FLAG_COUNT_MAPPED))
        return e3_fail(hipErrorInvalidValue, "num_rendered_host must be device-mapped pinned memory (E3DGS_FLAG_COUNT_MAPPED): "
                                             "it is how the caller learns the count, and whether it fitted the capacity");
    ViewBatch vb;
    rc = make_batch(nviews, P, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, vb);
    if (rc) return rc;
    struct Keep { e3dgs_alloc_fn fn; void* user; char* ptr; };
    static thread_local Keep kg, ki;
    kg = Keep{geom_alloc, geom_user, nullptr};
    ki = Keep{image_alloc, image_user, nullptr};
    auto grab = [](void* u, size_t n) -> char* { Keep* k = (Keep*)u; k->ptr = k->fn(k->user, n); return k->ptr; };
    rc = e3_forward_begin_impl(grab, &kg, grab, &ki, vb, P, D, M, width, height, means3D, shs, nullptr, opacities, scales,
                               scale_modifier, rotations, nullptr, radii, debug, flags, num_rendered_host,
                               (hipStream_t)stream);
    if (rc) return rc;
    DeferredColour dc;
    const bool defer = (flags & E3_FLAG_DEFER_COLOR) != 0;
    if (defer) {
        dc.views = vb;
        dc.D = D; dc.M = M; dc.flags = flags; dc.means3D = means3D; dc.shs = shs;
        dc.before = before_colour; dc.user = notify_user;
    }
    return e3_forward_finish_impl(binning_alloc, binning_user, vb.n, P, width, height, background, kg.ptr, ki.ptr,
                                  P > 0 ? capacity : 0, out_color, debug, flags, (hipStream_t)stream, defer ? &dc : nullptr,
                                  P > 0 ? 1 : 0);
}

static int backward_multi(int nviews, int P, int D, int M, int num_rendered, const float* background,
                          int width, int height, const float* means3D, const float* shs,
                          const float* opacities, const float* scales, float scale_modifier,
                          const float* rotations, const float* const* viewmatrix,
                          const float* const* projmatrix, const float* const* cam_pos, const float* tan_fovx,
                          const float* tan_fovy, const int* radii, const char* geom_buffer,
                          const char* binning_buffer, const char* image_buffer, const float* dL_dpix,
                          float* grad_acc, float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D,
                          float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dcolour_views, int debug,
                          int flags, void* stream, const float* dL_dpix_view0_stats, const float* rank1_weights = nullptr,
                          unsigned rank1_mask = 0u) {
    g_err[0] = 0;
    ViewBatch vb;
    Rank1Views r1;
    memset(&r1, 0, sizeof r1);
    if (rank1_mask != 0u) {
        if (!rank1_weights) return e3_fail(hipErrorInvalidValue, "rank1_mask without rank1_weights");
        if (nviews >= 1 && nviews <= E3_MAX_VIEWS && (rank1_mask >> nviews) != 0u)
            return e3_fail(hipErrorInvalidValue, "rank1_mask names a view beyond nviews");
        if (dL_dpix_view0_stats && (rank1_mask & 1u))
            return e3_fail(hipErrorInvalidValue, "view 0 carries the second gradient chain: it cannot be rank 1");
        r1.mask = rank1_mask;
        for (int v = 0; v < nviews && v < E3_MAX_VIEWS; ++v)
            for (int k = 0; k < 3; ++k) r1.w[v][k] = rank1_weights[3 * v + k];
    }
    int rc = make_batch(nviews, P, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, vb);
    if (rc) return rc;
    if (P > 0 && (!shs || !scales || !rotations)) return e3_fail(hipErrorInvalidValue, "shs + scales + rotations are required");
    // degree 4 (utils/sh_utils.py:97-110) with the SH gradient itself as output; the colour-gradient route (dL_dcolour_views:
    // the trainer's, whose SH optimizer kernels hold 16 coefficients per channel) keeps the reference model's degrees 0..3
    const int max_degree = dL_dcolour_views ? 3 : 4;
    if (D < 0 || D > max_degree || M < (D + 1) * (D + 1))
        return e3_fail(hipErrorInvalidValue, max_degree == 4 ? "SH degree must be 0..4 and M >= (D+1)^2"
                                                             : "SH degree must be 0..3 (with dL_dcolour_views) and M >= (D+1)^2");
    // nviews == 1 runs the general single-view kernel (unless dL_dcolour_views is taken), so pass 2 to apply the
    // multi-view argument rules always
    // dL_dsh may be NULL when the caller takes the per-view colour gradients instead (e3dgs_sh_grad_from_colour)
    float dummy_sh = 0.0f;
    rc = check_backward_args(2, P, shs, nullptr, opacities, nullptr, grad_acc, dL_dopacity, nullptr, dL_dmean3D,
                             nullptr, (dL_dsh || !dL_dcolour_views) ? dL_dsh : &dummy_sh, dL_dscale, dL_drot, flags);
    if (rc) return rc;
    if (dL_dpix_view0_stats && nviews < 2 && !dL_dcolour_views)
        return e3_fail(hipErrorInvalidValue, "dL_dpix_view0_stats needs the multi-view per-Gaussian kernel (nviews >= 2 or dL_dcolour_views)");
    if (dL_dpix_view0_stats && !dL_dmean2D)
        return e3_fail(hipErrorInvalidValue, "dL_dpix_view0_stats without dL_dmean2D: nothing would read the second chain");
    return e3_backward_impl(vb, P, D, M, num_rendered, background, width, height, means3D, shs, nullptr, opacities,
                            scales, scale_modifier, rotations, nullptr, radii, geom_buffer, binning_buffer, image_buffer,
                            dL_dpix, grad_acc, dL_dmean2D, dL_dopacity, nullptr, dL_dmean3D, nullptr, dL_dsh, dL_dscale,
                            dL_drot, debug, flags, (hipStream_t)stream, dL_dcolour_views, dL_dpix_view0_stats,
                            rank1_mask != 0u ? &r1 : nullptr);
}

int e3dgs_rasterize_backward_multi(int nviews, int P, int D, int M, int num_rendered, const float* background,
                                   int width, int height, const float* means3D, const float* shs,
                                   const float* opacities, const float* scales, float scale_modifier,
                                   const float* rotations, const float* const* viewmatrix,
                                   const float* const* projmatrix, const float* const* cam_pos, const float* tan_fovx,
                                   const float* tan_fovy, const int* radii, const char* geom_buffer,
                                   const char* binning_buffer, const char* image_buffer, const float* dL_dpix,
                                   float* grad_acc, float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D,
                                   float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dcolour_views, int debug,
                                   int flags, void* stream) {
    return backward_multi(nviews, P, D, M, num_rendered, background, width, height, means3D, shs, opacities, scales,
                          scale_modifier, rotations, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii,
                          geom_buffer, binning_buffer, image_buffer, dL_dpix, grad_acc, dL_dmean2D, dL_dopacity,
                          dL_dmean3D, dL_dsh, dL_dscale, dL_drot, dL_dcolour_views, debug, flags, stream, nullptr);
}

int e3dgs_rasterize_backward_multi_stats(int nviews, int P, int D, int M, int num_rendered, const float* background,
                                         int width, int height, const float* means3D, const float* shs,
                                         const float* opacities, const float* scales, float scale_modifier,
                                         const float* rotations, const float* const* viewmatrix,
                                         const float* const* projmatrix, const float* const* cam_pos,
                                         const float* tan_fovx, const float* tan_fovy, const int* radii,
                                         const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                                         const float* dL_dpix, const float* dL_dpix_view0_stats, float* grad_acc,
                                         float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D, float* dL_dsh,
                                         float* dL_dscale, float* dL_drot, float* dL_dcolour_views, int debug, int flags,
                                         void* stream) {
    return backward_multi(nviews, P, D, M, num_rendered, background, width, height, means3D, shs, opacities, scales,
                          scale_modifier, rotations, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii,
                          geom_buffer, binning_buffer, image_buffer, dL_dpix, grad_acc, dL_dmean2D, dL_dopacity,
                          dL_dmean3D, dL_dsh, dL_dscale, dL_drot, dL_dcolour_views, debug, flags, stream,
                          dL_dpix_view0_stats);
}

int e3dgs_rasterize_backward_multi_rank1(int nviews, int P, int D, int M, int num_rendered, const float* background,
                                         int width, int height, const float* means3D, const float* shs,
                                         const float* opacities, const float* scales, float scale_modifier,
                                         const float* rotations, const float* const* viewmatrix,
                                         const float* const* projmatrix, const float* const* cam_pos,
                                         const float* tan_fovx, const float* tan_fovy, const int* radii,
                                         const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                                         const float* dL_dpix, const float* dL_dpix_view0_stats,
                                         const float* rank1_weights, unsigned rank1_mask, float* grad_acc,
                                         float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D, float* dL_dsh,
                                         float* dL_dscale, float* dL_drot, float* dL_dcolour_views, int debug, int flags,
                                         void* stream) {
    return backward_multi(nviews, P, D, M, num_rendered, background, width, height, means3D, shs, opacities, scales,
                          scale_modifier, rotations, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii,
                          geom_buffer, binning_buffer, image_buffer, dL_dpix, grad_acc, dL_dmean2D, dL_dopacity,
                          dL_dmean3D, dL_dsh, dL_dscale, dL_drot, dL_dcolour_views, debug, flags, stream,
                          dL_dpix_view0_stats, rank1_weights, rank1_mask);
}

int e3dgs_sh_grad_from_colour(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                              const float* packed, size_t rank_stride, float scale, float* dL_dsh, int flags,
                              void* stream) {
    g_err[0] = 0;
    if (P < 0 || nranks < 1 || views_per_rank < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1))
        return e3_fail(hipErrorInvalidValue, "bad sizes");
    if (P > 0 && (!means3D || !packed || !dL_dsh)) return e3_fail(hipErrorInvalidValue, "null pointer");
    if (rank_stride < (size_t)views_per_rank * ((size_t)P * 3 + 3))
        return e3_fail(hipErrorInvalidValue, "rank_stride smaller than one rank block");
    int rc = e3_sh_grad_views_impl(P, nranks, views_per_rank, D, M, means3D, packed, rank_stride, scale, dL_dsh, flags,
                                   (hipStream_t)stream);
    return rc ? e3_fail((hipError_t)rc, "sh_grad_views_kernel") : 0;
}

int e3dgs_sh_adam_from_colour(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                              const float* packed, size_t rank_stride, float scale, float* sh, float* exp_avg,
                              float* exp_avg_sq, float lr_f_dc, float lr_f_rest, float beta1, float beta2, float eps,
                              int step, int flags, void* stream) {
    g_err[0] = 0;
    if (P < 0 || nranks < 1 || views_per_rank < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || step < 1)
        return e3_fail(hipErrorInvalidValue, "bad sizes");
    if (M % 2 != 0 || M > 64) return e3_fail(hipErrorInvalidValue, "M must be even (the optimizer kernel works on two coefficients per slice)");
    if (P > 0 && (!means3D || !packed || !sh || !exp_avg || !exp_avg_sq)) return e3_fail(hipErrorInvalidValue, "null pointer");
    if (rank_stride < (size_t)views_per_rank * ((size_t)P * 3 + 3))
        return e3_fail(hipErrorInvalidValue, "rank_stride smaller than one rank block");
    int rc = e3_sh_adam_views_impl(P, nranks, views_per_rank, D, M, means3D, packed, rank_stride, scale, sh, exp_avg,
                                   exp_avg_sq, lr_f_dc, lr_f_rest, beta1, beta2, eps, step, flags, (hipStream_t)stream);
    return rc ? e3_fail((hipError_t)rc, "sh_grad_views_kernel<adam>") : 0;
}

int e3dgs_sh_adam_from_colour_mean(int P, int nranks, int views_per_rank, int D, int M, const float* means3D,
                                   const float* packed, size_t rank_stride, float scale, float* sh, float* exp_avg,
                                   float* exp_avg_sq, float lr_f_dc, float lr_f_rest, float beta1, float beta2, float eps,
                                   int step, int flags, float* dL_dmean3D, void* stream) {
    g_err[0] = 0;
    if (P < 0 || nranks < 1 || views_per_rank < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || step < 1)
        return e3_fail(hipErrorInvalidValue, "bad sizes");
    if (M % 2 != 0 || M > 64) return e3_fail(hipErrorInvalidValue, "M must be even (the optimizer kernel works on two coefficients per slice)");
    if (nranks * views_per_rank > 4) return e3_fail(hipErrorInvalidValue, "the deferred position term needs at most 4 views in all");
    if (P > 0 && (!means3D || !packed || !sh || !exp_avg || !exp_avg_sq || !dL_dmean3D)) return e3_fail(hipErrorInvalidValue, "null pointer");
    if (rank_stride < (size_t)views_per_rank * ((size_t)P * 3 + 3))
        return e3_fail(hipErrorInvalidValue, "rank_stride smaller than one rank block");
    int rc = e3_sh_adam_views_impl(P, nranks, views_per_rank, D, M, means3D, packed, rank_stride, scale, sh, exp_avg,
                                   exp_avg_sq, lr_f_dc, lr_f_rest, beta1, beta2, eps, step, flags, (hipStream_t)stream,
                                   dL_dmean3D);
    return rc ? e3_fail((hipError_t)rc, "sh_adam_views_kernel<mean>") : 0;
}

size_t e3dgs_state_offset_emit_gid(int num_rendered) {
    char* p = nullptr;
    BinningState b = BinningState::from(p, (size_t)(num_rendered > 0 ? num_rendered : 0));
    return (size_t)b.emit_gid;
}
void e3dgs_state_offsets_binning(int num_rendered, size_t* out2) {
    char* p = nullptr;
    BinningState b = BinningState::from(p, (size_t)(num_rendered > 0 ? num_rendered : 0));
    out2[0] = (size_t)b.strip_mask; out2[1] = (size_t)b.touched;
}
void e3dgs_state_offsets(int P, int num_rendered, int width, int height, size_t* out9) {
    char* p = nullptr;
    GeomState g = GeomState::from(p, (size_t)(P > 0 ? P : 0));
    out9[0] = (size_t)g.rec; out9[1] = (size_t)g.rec + 16; out9[2] = (size_t)g.rec + 32; out9[3] = (size_t)g.clamped;
    out9[4] = (size_t)g.rect;
    p = nullptr;
    BinningState b = BinningState::from(p, (size_t)(num_rendered > 0 ? num_rendered : 0));
    out9[5] = (size_t)b.perm;
    p = nullptr;
    int gx = (width + E3_TILE - 1) / E3_TILE, gy = (height + E3_TILE - 1) / E3_TILE;
    ImageState im = ImageState::from(p, (size_t)width * height, (size_t)gx * gy);
    out9[6] = (size_t)im.ranges; out9[7] = (size_t)im.final_T; out9[8] = (size_t)im.n_contrib;
}

void e3dgs_state_offsets_multi(int nviews, int P, int num_rendered, int width, int height, size_t* out9) {
    char* p = nullptr;
    const size_t Q = (size_t)(P > 0 ? P : 0) * (size_t)(nviews > 0 ? nviews : 1);
    GeomState g = GeomState::from(p, Q);
    out9[0] = (size_t)g.rec; out9[1] = (size_t)g.rec + 16; out9[2] = (size_t)g.rec + 32; out9[3] = (size_t)g.clamped;
    out9[4] = (size_t)g.rect;
    p = nullptr;
    BinningState b = BinningState::from(p, (size_t)(num_rendered > 0 ? num_rendered : 0));
    out9[5] = (size_t)b.perm;
    p = nullptr;
    const int gx = (width + E3_TILE - 1) / E3_TILE, gy = (height + E3_TILE - 1) / E3_TILE;
    ImageState im = ImageState::from(p, (size_t)width * height * (nviews > 0 ? nviews : 1), (size_t)gx * gy * (nviews > 0 ? nviews : 1));
    out9[6] = (size_t)im.ranges; out9[7] = (size_t)im.final_T; out9[8] = (size_t)im.n_contrib;
}

int e3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                       void* stream) {
    (void)projmatrix;
    g_err[0] = 0;
    return e3_mark_visible_impl(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

size_t e3dgs_knn_scratch_bytes(int P) { return e3_knn_scratch_bytes(P); }
int e3dgs_dist_knn3(int P, const float* points, float* out, char* scratch, void* stream) {
    g_err[0] = 0;
    return e3_knn_impl(P, points, out, scratch, (hipStream_t)stream);
}

size_t e3dgs_event_loss_scratch_bytes(int width, int height) { return e3_event_scratch_bytes(width, height); }
int e3dgs_event_loss(int width, int height, const float* image, const float* img_now, const float* img_next,
                     const float* gt_int, const float* gt_now, const float* gt_next, const float* gt_blur,
                     const float* c, float gt_c, float* d_image, float* d_now, float* d_next, float* scalars_out,
                     float* dc_out, char* scratch, void* stream) {
    g_err[0] = 0;
    return e3_event_loss_impl(width, height, image, img_now, img_next, gt_int, gt_now, gt_next, gt_blur, c, gt_c,
                              d_image, d_now, d_next, scalars_out, scratch, (hipStream_t)stream, dc_out, nullptr, 0);
}
int e3dgs_event_loss_cached(int width, int height, const float* image, const float* img_now, const float* img_next,
                            const float* gt_int, const float* gt_now, const float* gt_next, const float* gt_blur,
                            const float* c, float gt_c, float* d_image, float* d_now, float* d_next, float* scalars_out,
                            float* dc_out, double* nz_count, int nz_valid, char* scratch, void* stream) {
    g_err[0] = 0;
    if (!nz_count) return e3_fail(hipErrorInvalidValue, "e3dgs_event_loss_cached: nz_count is NULL");
    return e3_event_loss_impl(width, height, image, img_now, img_next, gt_int, gt_now, gt_next, gt_blur, c, gt_c,
                              d_image, d_now, d_next, scalars_out, scratch, (hipStream_t)stream, dc_out, nz_count, nz_valid);
}

int e3dgs_event_loss_rank1(int width, int height, const float* image, const float* img_now, const float* img_next,
                           const float* gt_int, const float* gt_now, const float* gt_next, const float* gt_blur,
                           const float* c, float gt_c, float* d_image, float* d_now, float* d_next, float* scalars_out,
                           float* dc_out, double* nz_count, int nz_valid, char* scratch, void* stream) {
    g_err[0] = 0;
    return e3_event_loss_impl(width, height, image, img_now, img_next, gt_int, gt_now, gt_next, gt_blur, c, gt_c,
                              d_image, d_now, d_next, scalars_out, scratch, (hipStream_t)stream, dc_out, nz_count,
                              nz_count ? nz_valid : 0, 1);
}

size_t e3dgs_ssim_scratch_bytes(int channels, int height, int width) { return e3_ssim_scratch_bytes(channels, height, width); }
int e3dgs_ssim(int channels, int height, int width, int to_gray, const float* img1, const float* img2, float* ssim_mean,
               float* d_img1, char* scratch, void* stream) {
    g_err[0] = 0;
    if (channels <= 0 || height <= 0 || width <= 0) return e3_fail(hipErrorInvalidValue, "bad sizes");
    return e3_ssim_impl(channels, height, width, to_gray, img1, img2, ssim_mean, d_img1, scratch, (hipStream_t)stream);
}

int e3dgs_adam_step_segments(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                             const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2, int step,
                             void* stream) {
    g_err[0] = 0;
    if (!seg_end || !lr || !eps) return e3_fail(hipErrorInvalidValue, "segment tables are required");
    return e3_adam_segments_impl(n, param, grad, exp_avg, exp_avg_sq, nseg, seg_end, lr, eps, beta1, beta2, step, nullptr,
                                 (hipStream_t)stream);
}

int e3dgs_adam_step_groups(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                           const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2,
                           const int* steps, void* stream) {
    g_err[0] = 0;
    if (!seg_end || !lr || !eps || !steps) return e3_fail(hipErrorInvalidValue, "segment tables are required");
    return e3_adam_segments_impl(n, param, grad, exp_avg, exp_avg_sq, nseg, seg_end, lr, eps, beta1, beta2, 0, steps,
                                 (hipStream_t)stream);
}

int e3dgs_adam_step_groups_gap(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int nseg,
                               const size_t* seg_end, const float* lr, const float* eps, float beta1, float beta2,
                               const int* steps, size_t gap_begin, size_t gap_len, void* stream) {
    g_err[0] = 0;
    if (!seg_end || !lr || !eps || !steps) return e3_fail(hipErrorInvalidValue, "segment tables are required");
    return e3_adam_segments_impl(n, param, grad, exp_avg, exp_avg_sq, nseg, seg_end, lr, eps, beta1, beta2, 0, steps,
                                 (hipStream_t)stream, gap_begin, gap_len);
}

int e3dgs_densify_stats_update(int P, const float* viewspace_grad, const int* radii, float* max_radii2D,
                               float* xyz_gradient_accum, float* denom, void* stream) {
    g_err[0] = 0;
    if (P < 0) return e3_fail(hipErrorInvalidValue, "bad sizes");
    return e3_densify_stats_impl(P, viewspace_grad, radii, max_radii2D, xyz_gradient_accum, denom, (hipStream_t)stream);
}

size_t e3dgs_image_loss_scratch_bytes(int channels, int height, int width) {
    return e3_image_loss_scratch_bytes(channels, height, width);
}
int e3dgs_image_loss(int channels, int height, int width, int to_gray, float lambda_dssim, const float* image,
                     const float* gt_image, float* scalars, float* d_image, char* scratch, void* stream) {
    g_err[0] = 0;
    if (channels <= 0 || height <= 0 || width <= 0) return e3_fail(hipErrorInvalidValue, "bad sizes");
    return e3_image_loss_impl(channels, height, width, to_gray, lambda_dssim, image, gt_image, scalars, d_image, scratch,
                              (hipStream_t)stream);
}

int e3dgs_image_loss_rank1(int height, int width, float lambda_dssim, const float* image, const float* gt_image,
                           float* scalars, float* d_gray, char* scratch, void* stream) {
    g_err[0] = 0;
    if (height <= 0 || width <= 0) return e3_fail(hipErrorInvalidValue, "bad sizes");
    return e3_image_loss_impl(3, height, width, 1, lambda_dssim, image, gt_image, scalars, d_gray, scratch,
                              (hipStream_t)stream, 1);
}

int e3dgs_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                    float beta1, float beta2, float eps, int step, float lr_b, int period, int split, void* stream) {
    g_err[0] = 0;
    return e3_adam_impl(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, lr_b, period, split,
                        (hipStream_t)stream);
}

size_t e3dgs_densify_scratch_bytes(int P) { return e3_densify_scratch_bytes(P); }
int e3dgs_densify_plan(int P, const float* flat_param, const float* xyz_gradient_accum, const float* denom,
                       float max_grad, float min_opacity, float extent, float percent_dense, int size_prune, char* scratch,
                       int* counts_host4, void* stream) {
    g_err[0] = 0;
    if (P < 0 || !counts_host4) return e3_fail(hipErrorInvalidValue, "bad arguments");
    if (P > 0 && (!flat_param || !xyz_gradient_accum || !denom || !scratch)) return e3_fail(hipErrorInvalidValue, "null pointer");
    return e3_densify_plan_impl(P, flat_param, xyz_gradient_accum, denom, max_grad, min_opacity, extent, percent_dense,
                                size_prune, scratch, counts_host4, (hipStream_t)stream);
}
const int* e3dgs_densify_split_rows(int P, char* scratch) { return e3_densify_split_rows(P, scratch); }
int e3dgs_densify_apply(int P, int P_new, const int* counts4, const float* param, const float* exp_avg,
                        const float* exp_avg_sq, const float* samples, float* param_new, float* exp_avg_new,
                        float* exp_avg_sq_new, char* scratch, void* stream) {
    g_err[0] = 0;
    if (P < 0 || P_new < 0 || !counts4) return e3_fail(hipErrorInvalidValue, "bad arguments");
    if (P > 0 && (!param || !exp_avg || !exp_avg_sq || !param_new || !exp_avg_new || !exp_avg_sq_new || !scratch))
        return e3_fail(hipErrorInvalidValue, "null pointer");
    if (counts4[2] > 0 && !samples) return e3_fail(hipErrorInvalidValue, "samples are required when rows are split");
    return e3_densify_apply_impl(P, P_new, counts4, param, exp_avg, exp_avg_sq, samples, param_new, exp_avg_new,
                                 exp_avg_sq_new, scratch, (hipStream_t)stream);
}

size_t e3dgs_sort_scratch_bytes(size_t n) { return sort_scratch_words(n ? n : 1) * sizeof(uint32_t) + 256; }
int e3dgs_sort_pairs(size_t n, int nbits, int key_bytes, void* keys0, void* keys1, uint32_t* vals0, uint32_t* vals1,
                     int identity_payload, char* scratch, uint32_t* kept_count_dev, uint32_t* ranges, uint32_t nranges,
                     int* result_index_host, void* stream) {
    g_err[0] = 0;
    if ((key_bytes != 2 && key_bytes != 4) || nbits < 0 || nbits > 8 * key_bytes || !result_index_host)
        return e3_fail(hipErrorInvalidValue, "key_bytes must be 2 or 4, 0 <= nbits <= 8 * key_bytes");
    if (n >= 0xFFFFFFFFull) return e3_fail(hipErrorInvalidValue, "n must be below 2^32 - 1");
    if (n > 0 && (!keys0 || !keys1 || !vals0 || !vals1 || !scratch)) return e3_fail(hipErrorInvalidValue, "null pointer");
    if (kept_count_dev && key_bytes != 4) return e3_fail(hipErrorInvalidValue, "dropping all-ones keys needs 32-bit keys");
    if ((reinterpret_cast<uintptr_t>(scratch) & 7) != 0) return e3_fail(hipErrorInvalidValue, "scratch must be 8-byte aligned");
    uint32_t* vout = nullptr;
    int rc;
    if (key_bytes == 4) {
        uint32_t* kout;
        rc = launch_radix_sort_pairs((uint32_t*)keys0, (uint32_t*)keys1, vals0, vals1, n, nbits, (uint32_t*)scratch, &kout,
                                     &vout, (hipStream_t)stream, identity_payload != 0, kept_count_dev, nullptr,
                                     (uint2*)ranges, nranges);
    } else {
        uint16_t* kout;
        rc = launch_radix_sort_pairs_u16((uint16_t*)keys0, (uint16_t*)keys1, vals0, vals1, n, nbits, (uint32_t*)scratch,
                                         &kout, &vout, (hipStream_t)stream, identity_payload != 0, nullptr, (uint2*)ranges,
                                         nranges);
    }
    if (rc) return rc;
    *result_index_host = (vout == vals1) ? 1 : 0;
    return 0;
}

size_t e3dgs_depth_sort_scratch_bytes(size_t n) { return depth_sort_scratch_words(n ? n : 1) * sizeof(uint32_t) + 256; }
int e3dgs_sort_depth_keys(size_t n, uint32_t* keys0, uint32_t* keys1, uint32_t* order, uint32_t* order_alt, char* scratch,
                          uint32_t* kept_count_dev, void* stream) {
    g_err[0] = 0;
    if (n >= 0xFFFFFFFFull) return e3_fail(hipErrorInvalidValue, "n must be below 2^32 - 1");
    if (!kept_count_dev) return e3_fail(hipErrorInvalidValue, "kept_count_dev is required");
    if (n > 0 && (!keys0 || !keys1 || !order || !order_alt || !scratch)) return e3_fail(hipErrorInvalidValue, "null pointer");
    if ((reinterpret_cast<uintptr_t>(scratch) & 7) != 0) return e3_fail(hipErrorInvalidValue, "scratch must be 8-byte aligned");
    if (n == 0) {
        hipError_t e = hipMemsetAsync(kept_count_dev, 0, sizeof(uint32_t), (hipStream_t)stream);
        return e == hipSuccess ? 0 : e3_fail(e, "hipMemsetAsync(kept_count_dev)");
    }
    return launch_depth_sort_wide(keys0, keys1, order, order_alt, n, (uint32_t*)scratch, (hipStream_t)stream, kept_count_dev);
}

extern unsigned long long* g_trace;
void e3dgs_debug_set_trace(void* buf) { g_trace = (unsigned long long*)buf; }   /* not in the public header */
/* deprecated shims: they change the process-wide DEFAULTS that calls without E3DGS_FLAG_OPTIONS fall back to */
void e3dgs_set_tile_cull(int on) { g_tile_cull = on == 3 ? 3 : (on ? 1 : 0); }
int e3dgs_get_tile_cull(void) { return g_tile_cull; }
void e3dgs_set_small_scene_paths(int on) { g_small_scene_paths = on ? 1 : 0; }
int e3dgs_get_small_scene_paths(void) { return g_small_scene_paths; }

void e3dgs_profile_enable(int slot_mask) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (slot_mask != 0 && !g_ps) g_ps = new ProfState();
    if (g_ps) for (int i = 0; i < PS_COUNT; ++i) g_ps->used[i] = 0;
    g_prof_mask.store((unsigned)slot_mask, std::memory_order_relaxed);
}
void e3dgs_profile_select(int slot_mask) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (slot_mask != 0 && !g_ps) g_ps = new ProfState();
    g_prof_mask.store((unsigned)slot_mask, std::memory_order_relaxed);
}
int e3dgs_profile_query(int slot, double* total_ms, int* launches) {
    if (slot < 0 || slot >= PS_COUNT) return -1;
    double t = 0.0;
    int n = 0;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    ProfState* P = g_ps;
    const int used = P ? P->used[slot] : 0;
    for (int k = 0; k < used; ++k) {
        if (!P->pairs[slot][k].closed) continue;
        hipError_t e = hipEventSynchronize(P->pairs[slot][k].b);
        if (e != hipSuccess) return e3_fail(e, "profile event sync");
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, P->pairs[slot][k].a, P->pairs[slot][k].b);
        if (e != hipSuccess) return e3_fail(e, "profile elapsed");
        t += ms;
        ++n;
    }
    *total_ms = t;
    *launches = n;
    return 0;
}
const char* e3dgs_profile_slot_name(int slot) { return (slot >= 0 && slot < PS_COUNT) ? g_names[slot] : ""; }

}  // extern "C"
