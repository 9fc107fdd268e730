// common.h -- shared device helpers and scratch-buffer layouts (gfx950 / wave64 only).
//
// Arithmetic contract: this library is compiled with -ffp-contract=off; every fused
// multiply-add is an explicit FMA().  The forward kernels follow the operation order
// documented in DESIGN.md ("Arithmetic contract") so that integer outputs (radii, tile
// rectangles, sorted lists, n_contrib) and the fp32 image are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define E3_TILE 16
#define E3_NEAR_CULL_Z 0.2f
#define E3_GUARD_BAND 1.3f
#define E3_DILATION 0.3f
#define E3_EIGEN_FLOOR 0.1f
#define E3_ALPHA_CLAMP 0.99f
#define E3_ALPHA_SKIP (1.0f / 255.0f)
#define E3_T_STOP 0.0001f
#define E3_W_EPS 0.0000001f
#define E3_DET2_EPS 0.0000001f

// flags of e3dgs_rasterize_forward / _backward (include/e3dgs_hip.h)
#define E3_FLAG_PREACT 1      // scales = log-scales, rotations = raw quaternions, opacities = logits
#define E3_FLAG_ACCUMULATE 2  // backward adds into its outputs (only for visible Gaussians)
#define E3_FLAG_SH_PLANAR 4   // shs / dL_dsh are coefficient-major (M*3, P): lane-coalesced, no (P,M,3) stride
#define E3_FLAG_BWD_ONLY_RENDER 8   // backward: run only the compositing backward (fills grad_acc)
#define E3_FLAG_BWD_ONLY_GEOM 16    // backward: run only the per-Gaussian backward (consumes grad_acc)
#define E3_FLAG_COUNT_MAPPED 64     // begin: num_rendered_host is pinned + device-mapped; the GPU stores the count there
#define E3_FLAG_DEFER_COLOR 128     // multi begin: no SH evaluation in preprocess; finish runs colour_kernel
#define E3_FLAG_COUNT_DEVICE 256    // backward: num_rendered is the CAPACITY of a forward_multi_capacity call
#define E3_FLAG_DEFER_SH_MEAN 512   // backward_multi without dL_dsh: dL_dmean3D lacks the SH view-direction term (sh_adam_views_kernel adds it)
// per-call options (include/e3dgs_hip.h: E3DGS_FLAG_OPTIONS ...): with E3_FLAG_OPTIONS the bits below describe the call;
// without it the process-wide defaults apply (environment at load time / the deprecated e3dgs_set_* setters)
#define E3_FLAG_OPTIONS 0x0800
#define E3_FLAG_CULL_RECT 0x1000          // reference rectangle binning (no exact tile culling)
#define E3_FLAG_CULL_NO_BOX 0x2000        // exact culling without the tight candidate box
#define E3_FLAG_NO_SMALL_PATHS 0x4000     // large-scene work decomposition whatever the splat count
#define E3_FLAG_FAST_EXP 0x8000           // tolerance mode: hardware exp2 in the compositing forward (and its backward twin)
#define E3_FLAG_MEAN2D_VIEWS 0x10000     // backward_multi: dL_dmean2D is (nviews, P, 3): EVERY view's screen-space gradient, not view 0's only
#define E3_MAX_VIEWS 4       // views per e3dgs_rasterize_backward_geom_multi call
#define E3_ACC_STRIDE 12      // floats the CALLER provides per (tile, Gaussian) instance and per splat sum in grad_acc
#define E3_REC_FLOATS 9       // floats of a per-instance gradient record as stored (packed, 36 B; the rest is slack)

#define E3_TRACE_WORDS 6       // debug trace (e3dgs_debug_set_trace): per tile {wall start, wall end (100 MHz), entries, hw id,
                              // shader-cycle start, shader-cycle end (s_memtime)}
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define WAVE 64

__device__ __forceinline__ float exp_det(float x) {
    // 2^(x*log2e), degree-5 polynomial on the fractional part; v_rndne + v_ldexp.
    float t = fmaxf(x * 1.4426950408889634f, -126.0f);
    float n = __builtin_rintf(t);
    float f = t - n;
    float p = 0.0013218672247603536f;
    p = FMA(p, f, 0.009671698324382305f);
    p = FMA(p, f, 0.05550893023610115f);
    p = FMA(p, f, 0.24022237956523895f);
    p = FMA(p, f, 0.6931468844413757f);
    p = FMA(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}
// Same value wherever o*exp can reach 1/255; without the underflow clamp (v_cvt_i32_f32 saturates and
// v_ldexp_f32 flushes to 0, so far-negative arguments give 0 instead of ~2^-126: both are < 1/255).
__device__ __forceinline__ float exp_det_noclamp(float x) {
    float t = x * 1.4426950408889634f;
    float n = __builtin_rintf(t);
    float f = t - n;
    float p = 0.0013218672247603536f;
    p = FMA(p, f, 0.009671698324382305f);
    p = FMA(p, f, 0.05550893023610115f);
    p = FMA(p, f, 0.24022237956523895f);
    p = FMA(p, f, 0.6931468844413757f);
    p = FMA(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// activations of scene/gaussian_model.py:33-41,95-118 (used when E3_FLAG_PREACT is set): scaling = exp, opacity = sigmoid,
// rotation = F.normalize.  DETERMINISTIC forms, restated operation for operation by the CPU checker (gso_activate() under oracle/): the
// polynomial exp_det (|rel err| < 2e-6 against exp on |x| <= 20), IEEE division / sqrt, explicit fmaf chain -- so that the
// rasteriser fed raw parameters is bit-identical to the oracle fed gso_activate()'s values (radii, lists, image), exactly
// as the operator fed activated values is.  (libm-style expf differs from torch's by <= 1 ulp, which is enough to flip a
// radius or an alpha threshold at a handful of Gaussians per million.)
// NaN in -> NaN out, as torch.exp / torch.sigmoid: exp_det()'s underflow clamp fmaxf(t, -126) would swallow a NaN (a
// diverged opacity logit would render as opacity 1.0, a NaN log-scale as ~1e-38 -- finite values instead of a NaN loss).
__device__ __forceinline__ float act_exp(float x) { const float r = exp_det(x); return x != x ? x : r; }
__device__ __forceinline__ float act_sigmoid(float x) { const float r = 1.0f / (1.0f + exp_det(-x)); return x != x ? x : r; }
__device__ __forceinline__ void act_load_scale_rot(const float* __restrict__ s3, const float* __restrict__ q4, bool preact,
                                                   float s[3], float q[4], float& qinv) {
    s[0] = s3[0]; s[1] = s3[1]; s[2] = s3[2];
    q[0] = q4[0]; q[1] = q4[1]; q[2] = q4[2]; q[3] = q4[3];
    qinv = 1.0f;
    if (preact) {
        s[0] = act_exp(s[0]); s[1] = act_exp(s[1]); s[2] = act_exp(s[2]);
        float nrm = __builtin_sqrtf(FMA(q[0], q[0], FMA(q[1], q[1], FMA(q[2], q[2], q[3] * q[3]))));
        qinv = 1.0f / fmaxf(nrm, 1e-12f);          // torch.nn.functional.normalize eps
        q[0] *= qinv; q[1] *= qinv; q[2] *= qinv; q[3] *= qinv;
    }
}
// libm-style sigmoid for the densification decisions (pinned to torch.sigmoid by golden G8, csrc/densify.hip)
__device__ __forceinline__ float sigmoid_libm(float x) { return 1.0f / (1.0f + expf(-x)); }

// flat[4*c + r]: row r of the column-vector matrix applied to (x,y,z,1)
#define XFORM(M, r, x, y, z) FMA((M)[(r)], (x), FMA((M)[4 + (r)], (y), FMA((M)[8 + (r)], (z), (M)[12 + (r)])))

// Sum over the 64 lanes of a wave with DPP adds; the total lands in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false);  v += __int_as_float(x);  // quad_perm [1,0,3,2]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false);  v += __int_as_float(x);  // quad_perm [2,3,0,1]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false); v += __int_as_float(x);  // row_half_mirror
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false); v += __int_as_float(x);  // row_mirror
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false); v += __int_as_float(x);  // row_bcast:15 -> rows 1,3
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false); v += __int_as_float(x);  // row_bcast:31 -> rows 2,3
    return v;
}

// ---- DPP helpers (wave64 = 4 rows of 16 lanes; quad = 4 lanes)
#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DPP_ROW_SHL4 0x104   // row_shl:4: lane i reads lane i + 4 of its row of 16 (0 past the row end with bound_ctrl)
#define DPP_ROW_ROR8 0x128   // row_ror:8
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// bound_ctrl:1 form: for controls that read a valid lane everywhere (rotations) the result is the same, and the compiler
// may fold the move into the consuming add (v_add_f32_dpp: one instruction instead of v_mov + v_mov_dpp + v_add)
template <int CTRL>
__device__ __forceinline__ float dpp_bc_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}

__device__ __forceinline__ void wave_sync() {
    // Same-wave LDS hand-off: lanes run in lockstep and DS ops retire in order, so only
    // the compiler has to be kept from reordering across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The tile a compositing wave owns when the launch order is kept as per-class lists (ImageState::lpt_*): the unit-th tile
// in class order, or -1 behind the last one.  Wave-uniform; one 4-byte load per lane + one dependent load.
// (E3_LPT_CLASSES = 128, declared with ImageState below: two classes per lane)
__device__ __forceinline__ int lpt_lookup(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ list, uint32_t cap,
                                          uint32_t unit, int lane) {
    const uint2 c = reinterpret_cast<const uint2*>(cnt)[lane];                // classes 2 lane, 2 lane + 1
    uint32_t inc = c.x + c.y;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(unit < inc);   // lanes whose inclusive prefix passes `unit`
    if (hit == 0ull) return -1;
    const int l = __builtin_ctzll(hit);
    const uint32_t excl = (uint32_t)__builtin_amdgcn_readlane((int)(inc - c.x - c.y), l);
    const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)c.x, l);
    uint32_t idx = unit - excl, b = 2u * (uint32_t)l;
    if (idx >= cx) { idx -= cx; b += 1u; }
    return (int)list[(size_t)b * cap + idx];
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
static inline T* carve(char*& p, size_t count) {
    T* r = reinterpret_cast<T*>(p);
    p += align_up(count * sizeof(T), 256);
    return r;
}

// ---------------------------------------------------------------- radix sort / scan
constexpr int SORT_THREADS = 256;
#ifndef E3_SORT_ITEMS
#define E3_SORT_ITEMS 16
#endif
constexpr int SORT_ITEMS = E3_SORT_ITEMS;            // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS; // 4096 keys per workgroup
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

static inline size_t sort_blocks(size_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
static inline size_t scan_blocks(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
// scratch (in uint32) needed to sort n pairs: per-workgroup digit histograms (with the 256 extra workgroups of a
// segment-aligned last pass), the 257 segment boundaries, the 64-bit descriptors of the chained scan (+ ticket)
static inline size_t sort_scratch_words(size_t n) {
    size_t h = (sort_blocks(n) + 256) * 256;
    return h + 260 + 2 * (scan_blocks(h) + 1) + 64;
}
// words (uint32) of the descriptors of a chained scan over n elements (64-bit descriptors + ticket)
static inline size_t scan_desc_words(size_t n) { return 2 * (scan_blocks(n) + 1); }
// ... and for the depth sort's three 11-bit passes (launch_depth_sort_wide: 2048-bin histograms per workgroup); the
// geometry scratch holds whichever of the two is larger
static inline size_t depth_sort_scratch_words(size_t n) {
    const size_t hw = sort_blocks(n) * 2048;
    const size_t wide = hw + 2 + 2 * (scan_blocks(hw) + 1) + 64;
    const size_t narrow = sort_scratch_words(n);
    return wide > narrow ? wide : narrow;
}

// ---------------------------------------------------------------- scratch layouts
// Splats per binning wave = 1 << shift: 64 when there are plenty of splats, fewer when the launch would otherwise
// not fill the chip (bin_kernel).  Both halves of the forward derive the wave count from this.
static inline int e3_bin_group_shift(size_t Q, int adaptive) {
    int sh = 6;
    if (!adaptive) return sh;
    while (sh > 0 && ((Q + ((size_t)1 << sh) - 1) >> sh) < 8192) --sh;
    return sh;
}
// run_reduce: below this many splats one WAVE sums a splat's records (lanes stride over the run) instead of one thread
constexpr size_t E3_RUN_REDUCE_WAVE_MAX = 262144;

// geometry state: everything sized by P.  The first three members are what backward reads.
struct GeomState {
    float4* rec;       // 3 x float4 per Gaussian = one 48-B record the compositing kernels gather:
                       //   [0] (x, y, conic.x, conic.y)  [1] (conic.z, opacity, r, g)  [2] (b, pmin, -, -)
                       //   pmin: alpha >= 1/255 needs power >= pmin = -(ln(255 o) + margin)
    uint32_t* clamped; // SH clamp bitmask (bit ch)
    uint2* run;        // (first slot, kept instance count) of the splat at each DEPTH-SORTED position (ids in
                       // ord0): a splat's instances are emitted contiguously, slot = emission position, so the
                       // gradient records of the splats of a workgroup (in depth order) form one contiguous range
    uint2* rect;       // packed tile rectangle: .x = xmin | ymin<<16, .y = xmax | ymax<<16
    uint32_t* key0;    // depth keys (ping)
    uint32_t* key1;    // (pong)
    uint32_t* ord0;    // Gaussian ids (ping)
    uint32_t* ord1;    // (pong)
    uint32_t* tiles;   // tiles touched, in depth-sorted order
    uint32_t* offsets; // inclusive scan of `tiles`
    uint32_t* scratch; // sort/scan scratch
    uint32_t* total;   // [1] instance count (device copy)
    float4* binrec;    // 2 x float4 per DEPTH-SORTED position: the bin record the count pass derived (centre, conic,
                       // threshold, candidate box), read back -- coalesced -- by the emission pass
    uint32_t* nvis;    // [1] splats the projection kept = length of the depth-sorted order (key0 / ord0): the depth sort
                       // drops the culled ones in its first pass, and every kernel behind it reads its count here
    static size_t required(size_t P) {
        char* p = nullptr;
        from(p, P);
        return (size_t)p + 256;
    }
    static GeomState from(char*& p, size_t P) {
        GeomState g;
        size_t n = P ? P : 1;
        g.rec = carve<float4>(p, 3 * n);
        g.clamped = carve<uint32_t>(p, n);
        g.run = carve<uint2>(p, n);
        g.rect = carve<uint2>(p, n);
        g.key0 = carve<uint32_t>(p, n);
        g.key1 = carve<uint32_t>(p, n);
        g.ord0 = carve<uint32_t>(p, n);
        g.ord1 = carve<uint32_t>(p, n);
        g.tiles = carve<uint32_t>(p, n);
        g.offsets = carve<uint32_t>(p, n + 64);      // [0] = 0, then one entry per binning wave (<= n of them)
        g.scratch = carve<uint32_t>(p, depth_sort_scratch_words(n) + scan_desc_words(n) + 64);
        g.total = carve<uint32_t>(p, 64);
        g.nvis = g.total + 16;
        g.binrec = carve<float4>(p, 2 * n);
        return g;
    }
};

// binning state: everything sized by the instance count I.
struct BinningState {
    uint32_t* perm;       // sorted instance list (tile-major, depth order) as SLOT indices     [read by fwd + bwd]
    uint32_t* vals_alt;
    uint32_t* keys;       // tile ids
    uint32_t* keys_alt;
    uint32_t* emit_gid;   // splat id of each slot (= emission position)                       [read by fwd + bwd]
    uint8_t* strip_mask;  // per LIST POSITION: bit k = the forward evaluated this entry on the tile's 16x4 pixel strip k
                          // (written by render_fwd_kernel for the entries it visited, read by render_bwd_kernel)
    uint8_t* touched;     // per SLOT: 1 = the forward evaluated the instance on some strip, i.e. the compositing backward writes
                          // its gradient record (zeroed at emission, set by render_fwd_kernel, read by run_reduce_kernel: the
                          // other ~55 % of the records are neither written nor read)
    uint32_t* scratch;
    static size_t required(size_t I) {
        char* p = nullptr;
        from(p, I);
        return (size_t)p + 256;
    }
    static BinningState from(char*& p, size_t I) {
        BinningState b;
        size_t n = I ? I : 1;
        b.perm = carve<uint32_t>(p, n);
        b.vals_alt = carve<uint32_t>(p, n);
        b.keys = carve<uint32_t>(p, n);
        b.keys_alt = carve<uint32_t>(p, n);
        b.emit_gid = carve<uint32_t>(p, n);
        b.strip_mask = carve<uint8_t>(p, n + 64);
        b.scratch = carve<uint32_t>(p, sort_scratch_words(n));
        b.touched = carve<uint8_t>(p, n + 64);
        return b;
    }
};

// XCD-aware launch order of the compositing kernels: the dispatcher is observed to place workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, workgroup dispatch), each XCD has its own 4 MiB L2, and a splat's record is gathered by every
// tile it covers.  Tiles are therefore dealt to 8 regions in B x B-tile blocks, and the launch order is built so that the
// workgroups landing on XCD r hold tiles of region r only (longest lists first inside a region).
__host__ __device__ static inline int e3_xcd_region(int tile, int tiles_per_view, int gx, int B) {
    const int view = tile / tiles_per_view, lt = tile - view * tiles_per_view;
    const int ty = lt / gx, tx = lt - ty * gx;
    return ((tx / B) + 3 * (ty / B) + 5 * view) & 7;
}

// Launch order of the compositing kernels without an ordering kernel (E3_LPT_BUCKETS): the kernel that LEARNS a tile's
// cost appends the tile to one of 128 cost classes -- log scale, 8 per octave, class 0 = heaviest -- with one atomic:
// the last pass of the tile sort knows every list length when it writes the tile ranges (forward order), the forward
// compositing kernel knows the backward walk's cost when a tile finishes (backward order).  The consumer's wave u finds
// "the u-th tile in class order" itself: two counts per lane, a wave prefix sum, one ballot (lpt_lookup).  That removes two
// single-workgroup launches (14-20 us each, alone on the GPU) and their four kernel boundaries from every iteration.  The
// order inside a class is whatever the atomics produced: it only affects scheduling, never a result.
#ifndef E3_LPT_SUB
#define E3_LPT_SUB 3                                   // log2(classes per octave)
#endif
constexpr int E3_LPT_CLASSES = 128;                    // two per lane of the consumer's wave (lpt_lookup)
__host__ __device__ static inline uint32_t e3_lpt_class(uint32_t cost) {
    constexpr uint32_t S = E3_LPT_SUB, LIN = 1u << (S + 1);       // costs below LIN: one class each
    if (cost < LIN) return (uint32_t)(E3_LPT_CLASSES - 1) - cost;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t e = 31u - (uint32_t)__builtin_clz(cost);
#else
    uint32_t e = 0; while ((cost >> (e + 1)) != 0u) ++e;
#endif
    const uint32_t asc = LIN + ((e - (S + 1)) << S) + ((cost >> (e - S)) & ((1u << S) - 1u));
    return asc >= (uint32_t)E3_LPT_CLASSES ? 0u : (uint32_t)(E3_LPT_CLASSES - 1) - asc;
}

struct ImageState {
    uint2* ranges;       // per tile [start, end) into BinningState::perm
    float* final_T;      // per pixel
    uint32_t* n_contrib; // per pixel
    uint32_t* order;     // tiles by descending list length (launch order of the forward compositing kernel)
    uint32_t* work;      // per tile: entries the backward walk has to visit (max n_contrib), written by forward
    uint32_t* order_bwd; // tiles by descending `work` (launch order of the backward compositing kernel)
    uint32_t* lpt_cnt;   // [2][E3_LPT_CLASSES] tiles per cost class: [0] forward order, [1] backward order (zeroed by preprocess)
    uint32_t* lpt_list;  // [2][E3_LPT_CLASSES][ntiles] the tiles of each class
    static size_t required(size_t npix, size_t ntiles) {
        char* p = nullptr;
        from(p, npix, ntiles);
        return (size_t)p + 256;
    }
    static ImageState from(char*& p, size_t npix, size_t ntiles) {
        ImageState s;
        s.ranges = carve<uint2>(p, ntiles ? ntiles : 1);
        s.final_T = carve<float>(p, npix ? npix : 1);
        s.n_contrib = carve<uint32_t>(p, npix ? npix : 1);
        s.order = carve<uint32_t>(p, 2 * ntiles + 64);        // XCD-partitioned orders leave holes (launch_tile_order)
        s.work = carve<uint32_t>(p, ntiles ? ntiles : 1);
        s.order_bwd = carve<uint32_t>(p, 2 * ntiles + 64);
        s.lpt_cnt = carve<uint32_t>(p, 2 * E3_LPT_CLASSES);
        s.lpt_list = carve<uint32_t>(p, 2 * (size_t)E3_LPT_CLASSES * (ntiles ? ntiles : 1));
        return s;
    }
};

// per-view constants: matrices stay on the device (they arrive as torch tensors), scalars by value
struct ViewParams {
    const float* view;   // 16, row-vector layout
    const float* proj;   // 16
    const float* campos; // 3
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy;
};

// A call renders n <= E3_MAX_VIEWS views of the SAME Gaussians at the SAME resolution (the three renders of
// an event iteration, train.py:144,159,161; n = 1 for the plain operator).  Everything per (Gaussian, view)
// -- a "splat" -- is indexed  q = i * n + v  (Gaussian-major, so one Gaussian's splats and gradient-record
// runs are adjacent), tiles are numbered  v * ntiles + ty * gx + tx,  and every kernel of the pipeline runs
// ONCE over all views: n times fewer launches, n times larger (better filled) grids, parameters read once.
struct ViewSet {
    int n;
    ViewParams v[E3_MAX_VIEWS];
};

// host-side description of the views of one call (arrays of per-view values)
struct ViewBatch {
    int n;
    const float* view[E3_MAX_VIEWS];
    const float* proj[E3_MAX_VIEWS];
    const float* campos[E3_MAX_VIEWS];
    float tanfovx[E3_MAX_VIEWS], tanfovy[E3_MAX_VIEWS];
};
static inline ViewSet make_view_set(const ViewBatch& b, int W, int H, float scale_modifier) {
    ViewSet vs;
    vs.n = b.n;
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        const int u = v < b.n ? v : 0;         // unused slots mirror view 0 (never used: v >= n)
        ViewParams& w = vs.v[v];
        w.view = b.view[u]; w.proj = b.proj[u]; w.campos = b.campos[u];
        w.tanfovx = b.tanfovx[u]; w.tanfovy = b.tanfovy[u];
        w.focal_x = (float)W / (2.0f * b.tanfovx[u]);
        w.focal_y = (float)H / (2.0f * b.tanfovy[u]);
        w.scale_modifier = scale_modifier;
        w.W = W; w.H = H;
        w.gx = (W + E3_TILE - 1) / E3_TILE;
        w.gy = (H + E3_TILE - 1) / E3_TILE;
    }
    return vs;
}

// Views whose pixel gradient is rank 1: dL/dC(pixel) = s(pixel) * w with one weight vector per view (a loss on a luminance:
// the contrast renders of an event iteration, every --gray loss).  render_bwd_body<RANK1> in backward.hip.
struct Rank1Views {
    float w[E3_MAX_VIEWS][3];   // weight vector of a rank-1 view
    uint32_t mask;              // bit v: view v's pixel gradient is plane 0 of its (3, H, W) block times w[v]
};

// Adam's betas cross the C ABI as fp32, but torch.optim.Adam evaluates 1 - beta, beta^step and their roots on the Python
// double the caller wrote (0.9, 0.999: short decimals).  (float)0.999 = 0.99900001287..., and 1 - that is 1.3e-5 away
// from the fp32 value of 1 - 0.999 torch multiplies g^2 with.  The host therefore recovers the decimal (7 digits: exact
// for anything a caller writes as a literal) and derives every constant from it, as torch does.
// The decimal is only taken when it IS what the caller wrote, i.e. when it converts back to the same fp32 value, and when
// it lies in [0, 1): a beta that is not a 7-digit decimal (a schedule, 1 - 1/k) or within 5e-8 of 1 keeps its own value
// (rounding 0.99999997f to 1.0 would make both bias corrections 0 and the step inf / NaN).
static inline double e3_beta_double(float b) {
    const double d = __builtin_nearbyint((double)b * 1e7) / 1e7;
    return ((float)d == b && d >= 0.0 && d < 1.0) ? d : (double)b;
}
static inline float e3_one_minus_beta(float b) { return (float)(1.0 - e3_beta_double(b)); }

// the options of one call, resolved from its flags word (capi.hip holds the process-wide defaults)
struct CallOpts {
    int cull;          // 0: reference rectangle binning, 1: exact culling, 3: exact culling without the tight box
    int small_paths;   // work decomposition adapts to few splats
    int fast_exp;      // hardware exp2 in compositing (tolerance mode)
};
CallOpts e3_call_opts(int flags);
// What `begin` resolved for a geometry scratch, kept on the HOST per scratch address (capi.hip): `finish` and the backward
// derive scratch offsets and launch shapes from the option bits and the splat count, so a caller that hands them other
// bits than it gave `begin` would make the kernels write out of bounds.  remember() is called by begin, check() by every
// later half: non-zero (hipErrorInvalidValue + message, nothing launched) on a mismatch.
void e3_geom_opts_remember(const void* geom, size_t Q, const CallOpts& o);
int e3_geom_opts_check(const void* geom, size_t Q, const CallOpts& o, const char* who);

// host drivers (forward.hip / backward.hip), called by the C ABI wrappers in capi.hip
typedef char* (*e3_alloc_fn)(void*, size_t);
int e3_forward_begin_impl(e3_alloc_fn geom_alloc, void* geom_user, e3_alloc_fn img_alloc, void* img_user,
                          const ViewBatch& views, int P, int D, int M, int W, int H, const float* means3D,
                          const float* shs, const float* colors, const float* opac, const float* scales,
                          float scale_modifier, const float* rots, const float* cov_pre, int* radii, int debug,
                          int flags, int* count_host, hipStream_t s);
struct DeferredColour {          // inputs of colour_kernel (E3_FLAG_DEFER_COLOR), handed to `finish`
    ViewBatch views;
    int D, M, flags;
    const float* means3D;
    const float* shs;
    void (*before)(void*);      // optional: called on the host right before colour_kernel is enqueued
    void* user;
};
int e3_forward_finish_impl(e3_alloc_fn bin_alloc, void* bin_user, int nviews, int P, int W, int H,
                           const float* background, char* geom_buffer, char* image_buffer, int num_rendered,
                           float* out_color, int debug, int flags, hipStream_t s, const DeferredColour* dc = nullptr,
                           int count_on_device = 0);
int e3_backward_impl(const ViewBatch& views, int P, int D, int M, int num_rendered, const float* background, int W,
                     int H, const float* means3D, const float* shs, const float* colors, const float* opacities,
                     const float* scales, float scale_modifier, const float* rots, const float* cov_pre,
                     const int* radii, const char* geom_buffer, const char* binning_buffer,
                     const char* image_buffer, const float* dL_dpix, float* grad_acc, float* dL_dmean2D,
                     float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                     float* dL_dscale, float* dL_drot, int debug, int flags, hipStream_t s,
                     float* dL_dcolour_views = nullptr, const float* dL_dpix_stats = nullptr,
                     const struct Rank1Views* rank1 = nullptr);
int e3_sh_grad_views_impl(int P, int nranks, int views_per_rank, int D, int M, const float* means3D, const float* packed,
                          size_t rank_stride, float scale, float* dL_dsh, int flags, hipStream_t s);

int e3_sh_adam_views_impl(int P, int nranks, int views_per_rank, int D, int M, const float* means3D, const float* packed,
                          size_t rank_stride, float scale, float* sh, float* exp_avg, float* exp_avg_sq, float lr_dc,
                          float lr_rest, float b1, float b2, float eps, int step, int flags, hipStream_t s, float* dmean = nullptr);

// launchers implemented in scan_sort.hip (0 or a hipError_t code; the text is in e3dgs_last_error())
int launch_scan_chained_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* desc_zeroed, bool inclusive,
                            hipStream_t s, int* total_host = nullptr, uint32_t* total_dev = nullptr);
                            // one launch; desc_zeroed: scan_desc_words(n) words, 8-byte aligned, zero on entry;
                            // total_host: mapped host word / total_dev: device word that receives the grand total
int launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* scratch, bool inclusive,
                              hipStream_t s);
// Stable LSD radix sort of (key,val) pairs on bits [0,nbits).  The result is returned in (*keys_out,*vals_out), which
// are one of the two provided buffer pairs.
// identity_payload: v0 is NOT read; the payload of element i is i (saves writing and reading the index array).
// drop_count_dev:   (u32 keys) keys equal to 0xFFFFFFFF are dropped by the first pass; the number of kept keys is stored
//                   there and the later passes sort those only.
// n_dev:            the element count lives in device memory (n is then the host's upper bound, which sizes the grids).
// ranges_out:       after the sort, ranges_out[k] = [first, last + 1) positions of key k (k < nranges); keys without an
//                   element keep what they held (an empty range).  A two-pass sort derives them inside its last pass and
//                   writes no sorted keys (*keys_out = NULL).
// lpt_cnt / lpt_list: (a two-pass sort with ranges_out) the last pass also appends every key value k < nranges to the list
//                   of its cost class e3_lpt_class(run length): lpt_list[class * nranges + lpt_cnt[class]++] = k  (ImageState).
int launch_radix_sort_pairs(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                            uint32_t* scratch, uint32_t** keys_out, uint32_t** vals_out, hipStream_t s,
                            bool identity_payload = false, uint32_t* drop_count_dev = nullptr,
                            const uint32_t* n_dev = nullptr, uint2* ranges_out = nullptr, uint32_t nranges = 0,
                            uint32_t* lpt_cnt = nullptr, uint32_t* lpt_list = nullptr);
// the depth sort as three 11-bit passes (scan_sort.hip): identity payload, all-ones keys dropped, order in v0
int launch_depth_sort_wide(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, uint32_t* scratch,
                           hipStream_t s, uint32_t* kept_dev);
int launch_radix_sort_pairs_u16(uint16_t* k0, uint16_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                                uint32_t* scratch, uint16_t** keys_out, uint32_t** vals_out, hipStream_t s,
                                bool identity_payload = false, const uint32_t* n_dev = nullptr,
                                uint2* ranges_out = nullptr, uint32_t nranges = 0,
                                uint32_t* lpt_cnt = nullptr, uint32_t* lpt_list = nullptr);
// whether a forward with these sizes keeps its launch orders as per-class lists (both halves of the forward and the
// backward must agree): a two-pass tile sort that derives the ranges, and no XCD-partitioned order requested
bool e3_use_lpt_lists(int ntiles, int num_rendered, int P);
static inline int radix_passes(int nbits) { return (nbits + 7) / 8; }

// ---------------------------------------------------------------- optional event profiler (capi.hip)
enum ProfSlot { PS_PREPROCESS = 0, PS_SORT_DEPTH, PS_SCAN_EMIT, PS_SORT_TILE, PS_RANGES, PS_RENDER_FWD, PS_RENDER_BWD,
                PS_GEOM_BWD, PS_COUNT };
#include <atomic>
extern std::atomic<unsigned> g_prof_mask;  // slots being timed (0: off); process-wide -- autograd's worker threads are timed too
int prof_begin(int slot, hipStream_t s);   // -> index of the reserved event pair, or -1
void prof_end(int slot, int pair, hipStream_t s);
struct ProfScope {
    int slot, pair; hipStream_t s;
    ProfScope(int slot_, hipStream_t s_) : slot(slot_), pair(-1), s(s_) {
        if ((g_prof_mask.load(std::memory_order_relaxed) >> slot) & 1u) pair = prof_begin(slot, s);
    }
    ~ProfScope() { if (pair >= 0) prof_end(slot, pair, s); }
};
