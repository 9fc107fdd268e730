// forward.hip -- forward rasterisation kernels (preprocess, binning, compositing).
//
// Re-creates, MI355X-first, the forward half of the reference's absent CUDA op
// (SURVEY 2.1 / 8a rows a13-a18 [UPSTREAM]; call site gaussian_renderer/__init__.py:89-97):
//
//   preprocess_kernel     a13   cull, project, Sigma3, EWA Sigma2, conic, radius, tile rect, colour
//   (radix sort A)        ---   visible Gaussians by fp32 depth bits (stable -> ties keep index order)
//   gather_tiles_kernel   a14   tiles touched in depth order (then inclusive scan -> offsets, I)
//   emit_kernel           a15   (tile id, Gaussian id) instances, already depth-ordered
//   (radix sort B)        a16   stable sort by tile id only (ceil(log2 T) bits, 2 passes at 1080p)
//   tile_ranges_kernel    a17   [start,end) of each tile
//   render_fwd_kernel     a18   ONE WAVE PER 16x16 TILE, 4 pixels per lane, no workgroup barriers
//
// The two-level sort (depth on P Gaussians, then tile on I instances) yields exactly the order
// of the reference's single 64-bit (tile<<32 | depth) stable sort with 6x less sort traffic.
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------ SH colour
// element (k, ch) of one Gaussian's SH block lives at sh[(k*3 + ch) * st]: st = 1 for the reference's
// (P,M,3) layout, st = P for the coefficient-major layout (E3_FLAG_SH_PLANAR, coalesced across lanes)
// The degree is a template argument so that ALL coefficient loads of a splat are issued back to back before the first
// use: with a run-time degree every `if (D > n)` block ends in its own wait on its own loads -- four dependent
// memory round trips per thread, which made the projection kernel latency-bound (80 % of its wave cycles waiting).
template <int D, bool REGS = false>
__device__ __forceinline__ void sh_to_rgb_t(const float* sh, size_t st, float mx, float my, float mz,
                                            const float* __restrict__ campos, float rgb[3], uint32_t& clamped) {
    const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
    const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f,
                C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                C36 = -0.5900435899266435f;
    // (degree 4, 75 coefficients, is not batched: it would cost every launch half its occupancy for a rarely used degree)
    constexpr int NC = 3 * (D + 1) * (D + 1);
    constexpr bool BATCH = D <= 3 && !REGS;     // REGS: `sh` already is a register array (sh_load_regs)
    float cf[BATCH ? NC : 1];
    if (BATCH) {
#pragma unroll
        for (int k = 0; k < NC; ++k) cf[k] = sh[(size_t)k * st];
    }
#define SHC(k) (BATCH ? cf[BATCH ? (k) : 0] : sh[(size_t)(k) * st])
    float dx = mx - campos[0], dy = my - campos[1], dz = mz - campos[2];
    float len = __builtin_sqrtf(FMA(dx, dx, FMA(dy, dy, dz * dz)));
    float x = dx / len, y = dy / len, z = dz / len;
    clamped = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float r = SH_C0 * SHC(0 * 3 + ch);
        if (D > 0) {
            r = FMA(-(SH_C1 * y), SHC(1 * 3 + ch), r);
            r = FMA(SH_C1 * z, SHC(2 * 3 + ch), r);
            r = FMA(-(SH_C1 * x), SHC(3 * 3 + ch), r);
            if (D > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = FMA(C20 * xy, SHC(4 * 3 + ch), r);
                r = FMA(C21 * yz, SHC(5 * 3 + ch), r);
                r = FMA(C22 * (FMA(2.0f, zz, -xx) - yy), SHC(6 * 3 + ch), r);
                r = FMA(C23 * xz, SHC(7 * 3 + ch), r);
                r = FMA(C24 * (xx - yy), SHC(8 * 3 + ch), r);
                if (D > 2) {
                    r = FMA(C30 * y * FMA(3.0f, xx, -yy), SHC(9 * 3 + ch), r);
                    r = FMA(C31 * xy * z, SHC(10 * 3 + ch), r);
                    r = FMA(C32 * y * (FMA(4.0f, zz, -xx) - yy), SHC(11 * 3 + ch), r);
                    r = FMA(C33 * z * (FMA(2.0f, zz, -(3.0f * xx)) - 3.0f * yy), SHC(12 * 3 + ch), r);
                    r = FMA(C34 * x * (FMA(4.0f, zz, -xx) - yy), SHC(13 * 3 + ch), r);
                    r = FMA(C35 * z * (xx - yy), SHC(14 * 3 + ch), r);
                    r = FMA(C36 * x * FMA(-3.0f, yy, xx), SHC(15 * 3 + ch), r);
                    if (D > 3) {     // utils/sh_utils.py:97-110 (25 coefficients)
                        const float C40 = 2.5033429417967046f, C41 = -1.7701307697799304f, C42 = 0.9461746957575601f,
                                    C43 = -0.6690465435572892f, C44 = 0.10578554691520431f, C45 = -0.6690465435572892f,
                                    C46 = 0.47308734787878004f, C47 = -1.7701307697799304f, C48 = 0.6258357354491761f;
                        r = FMA(C40 * xy * (xx - yy), SHC(16 * 3 + ch), r);
                        r = FMA(C41 * yz * FMA(3.0f, xx, -yy), SHC(17 * 3 + ch), r);
                        r = FMA(C42 * xy * FMA(7.0f, zz, -1.0f), SHC(18 * 3 + ch), r);
                        r = FMA(C43 * yz * FMA(7.0f, zz, -3.0f), SHC(19 * 3 + ch), r);
                        r = FMA(C44 * FMA(zz, FMA(35.0f, zz, -30.0f), 3.0f), SHC(20 * 3 + ch), r);
                        r = FMA(C45 * xz * FMA(7.0f, zz, -3.0f), SHC(21 * 3 + ch), r);
                        r = FMA(C46 * (xx - yy) * FMA(7.0f, zz, -1.0f), SHC(22 * 3 + ch), r);
                        r = FMA(C47 * xz * FMA(-3.0f, yy, xx), SHC(23 * 3 + ch), r);
                        r = FMA(C48 * (xx * FMA(-3.0f, yy, xx) - yy * FMA(3.0f, xx, -yy)), SHC(24 * 3 + ch), r);
                    }
                }
            }
        }
        r = r + 0.5f;
        if (r < 0.0f) clamped |= (1u << ch);
        rgb[ch] = fmaxf(r, 0.0f);
    }
#undef SHC
}

__device__ __forceinline__ void sh_to_rgb(int D, const float* __restrict__ sh, size_t st, float mx, float my, float mz,
                                          const float* __restrict__ campos, float rgb[3], uint32_t& clamped) {
    switch (D) {
    case 0: sh_to_rgb_t<0>(sh, st, mx, my, mz, campos, rgb, clamped); break;
    case 1: sh_to_rgb_t<1>(sh, st, mx, my, mz, campos, rgb, clamped); break;
    case 2: sh_to_rgb_t<2>(sh, st, mx, my, mz, campos, rgb, clamped); break;
    case 3: sh_to_rgb_t<3>(sh, st, mx, my, mz, campos, rgb, clamped); break;
    default: sh_to_rgb_t<4>(sh, st, mx, my, mz, campos, rgb, clamped); break;
    }
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* s3, float mod, const float* q, float cov[6]) {
    float sx = mod * s3[0], sy = mod * s3[1], sz = mod * s3[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.0f - 2.0f * FMA(y, y, z * z), R01 = 2.0f * FMA(x, y, -(r * z)), R02 = 2.0f * FMA(x, z, r * y);
    float R10 = 2.0f * FMA(x, y, r * z), R11 = 1.0f - 2.0f * FMA(x, x, z * z), R12 = 2.0f * FMA(y, z, -(r * x));
    float R20 = 2.0f * FMA(x, z, -(r * y)), R21 = 2.0f * FMA(y, z, r * x), R22 = 1.0f - 2.0f * FMA(x, x, y * y);
    float L00 = R00 * sx, L01 = R01 * sy, L02 = R02 * sz;
    float L10 = R10 * sx, L11 = R11 * sy, L12 = R12 * sz;
    float L20 = R20 * sx, L21 = R21 * sy, L22 = R22 * sz;
    cov[0] = FMA(L00, L00, FMA(L01, L01, L02 * L02));
    cov[1] = FMA(L00, L10, FMA(L01, L11, L02 * L12));
    cov[2] = FMA(L00, L20, FMA(L01, L21, L02 * L22));
    cov[3] = FMA(L10, L10, FMA(L11, L11, L12 * L12));
    cov[4] = FMA(L10, L20, FMA(L11, L21, L12 * L22));
    cov[5] = FMA(L20, L20, FMA(L21, L21, L22 * L22));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------ preprocess
// One thread per GAUSSIAN; its views are a loop.  Everything the views share -- mean, covariance, opacity and the SH
// coefficients (81 % of the parameter bytes) -- is loaded once, in ONE batch of independent loads at the top of the
// thread, and the per-view work (projection, EWA, rectangle, colour) then runs from registers.  (Round 1 ran one
// thread per splat with the SH loads inside the nested degree tests behind the culling tests: five dependent memory
// round trips per thread, 80 % of the wave cycles waiting, 0.15 ms for 0.43 GB.)  The coefficients of Gaussians that
// turn out to be culled in every view are read for nothing (contiguous rows, ~1/4 of them on the benchmark scene):
// cheaper than a second round trip.  The SH degree is a template argument (the host picks the instantiation): the
// coefficient array then lives in registers with static indices.
#ifndef E3_PRE_WAVES
#define E3_PRE_WAVES 5
#endif
template <int DEG, bool VEC4 /* (P, M, 3) coefficient rows, 16-byte aligned: loaded as float4 */>
__global__ __launch_bounds__(256, E3_PRE_WAVES) void preprocess_kernel(
    int P, int M, const float* __restrict__ means, const float* __restrict__ shs,
    const float* __restrict__ colors, const float* __restrict__ opac, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ cov_pre, ViewSet vs, int flags,
    int* __restrict__ radii,
    float4* __restrict__ rec, uint32_t* __restrict__ clamped,
    uint2* __restrict__ rect, uint32_t* __restrict__ key,
    uint2* __restrict__ ranges_to_zero, int nranges, uint32_t* __restrict__ offsets0,
    uint32_t* __restrict__ scan_desc, int ndesc, uint32_t* __restrict__ lpt_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * E3_LPT_CLASSES) lpt_cnt[i] = 0u;       // class counters of both launch orders (ImageState::lpt_cnt)
    // housekeeping that would otherwise be memset commands (each costs a barrier packet on the queue): the tile ranges,
    // offsets[0], and the descriptors of the single-launch scan behind the binning count pass
    for (int t = i; t < nranges; t += gridDim.x * blockDim.x) ranges_to_zero[t] = make_uint2(0u, 0u);
    for (int t = i; t < ndesc; t += gridDim.x * blockDim.x) scan_desc[t] = 0u;
    if (i == 0) *offsets0 = 0u;
    if (i >= P) return;
    const int nv = vs.n;
    constexpr int NCF = 3 * (DEG + 1) * (DEG + 1);
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    // colours of all views first, branch-free (independent chains the scheduler can interleave), so that the coefficient
    // registers are free again before the per-view geometry: views that turn out culled cost ~100 wasted VALU slots
    // each, the kernel gets 6 instead of 4 waves per SIMD for its long dependent division / sqrt chains
    float rgbv[E3_MAX_VIEWS][3];
    uint32_t clv[E3_MAX_VIEWS];
    if (shs && !(flags & E3_FLAG_DEFER_COLOR)) {
        const bool planar = (flags & E3_FLAG_SH_PLANAR) != 0;
        const float* __restrict__ sh = planar ? shs + i : shs + (size_t)i * M * 3;
        const size_t st = planar ? (size_t)P : (size_t)1;
        float cf[NCF];
        if (VEC4) {
            // the reference's (P, M, 3) layout: a lane's coefficients are one 192-byte row -> 16-byte loads (a quarter of
            // the load instructions and cache-line requests of dword loads at a 192-byte lane stride)
            const float4* __restrict__ s4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
            for (int k = 0; k < NCF / 4; ++k) {
                const float4 q = s4[k];
                cf[4 * k] = q.x; cf[4 * k + 1] = q.y; cf[4 * k + 2] = q.z; cf[4 * k + 3] = q.w;
            }
#pragma unroll
            for (int k = NCF / 4 * 4; k < NCF; ++k) cf[k] = sh[k];
        } else {
#pragma unroll
            for (int k = 0; k < NCF; ++k) cf[k] = sh[(size_t)k * st];
        }
#pragma unroll
        for (int v = 0; v < E3_MAX_VIEWS; ++v) {
            if (v >= nv) break;
            sh_to_rgb_t<DEG, true>(cf, 1, mx, my, mz, vs.v[v].campos, rgbv[v], clv[v]);
        }
    } else {
#pragma unroll
        for (int v = 0; v < E3_MAX_VIEWS; ++v) {
            clv[v] = 0u;
            if (shs) { rgbv[v][0] = rgbv[v][1] = rgbv[v][2] = 0.0f; }      // E3_FLAG_DEFER_COLOR: colour_kernel fills colour + clamp mask later
            else { rgbv[v][0] = colors[3 * (size_t)i]; rgbv[v][1] = colors[3 * (size_t)i + 1]; rgbv[v][2] = colors[3 * (size_t)i + 2]; }
        }
    }
    float S[6];
    if (cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = cov_pre[6 * (size_t)i + k];
    } else {
        float sc[3], qn[4], qinv;
        act_load_scale_rot(scales + 3 * (size_t)i, rots + 4 * (size_t)i, (flags & E3_FLAG_PREACT) != 0, sc, qn, qinv);
        cov3d_from_scale_rot(sc, vs.v[0].scale_modifier, qn, S);
    }
    const float o_ = (flags & E3_FLAG_PREACT) ? act_sigmoid(opac[i]) : opac[i];
    // strip-skip bound used by the compositing kernels: alpha >= 1/255 needs power >= pmin
    // (in the forward the same fp32 `power` feeds both tests, so the margin only has to cover exp/log rounding, ~1e-6;
    // backward evaluates `power` in another order, which moves it by ~1e-7 of its largest term: 2e-4 covers that too)
    const float pmin = -(logf(255.0f * o_) + 2e-4f);
    // (unrolled with a uniform guard: a run-time index into the by-value ViewSet would move it to scratch memory)
#pragma unroll
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        if (v >= nv) break;
        const ViewParams& vp = vs.v[v];
        const size_t q = (size_t)i * nv + v;
        const float* V = vp.view;
        const float* Pm = vp.proj;
        int radius_out = 0;
        uint2 rect_out = make_uint2(0u, 0u);
        uint32_t key_out = 0xFFFFFFFFu;
        float vx = XFORM(V, 0, mx, my, mz), vy = XFORM(V, 1, mx, my, mz), vz = XFORM(V, 2, mx, my, mz);
        if (vz > E3_NEAR_CULL_Z) {
            float hx = XFORM(Pm, 0, mx, my, mz), hy = XFORM(Pm, 1, mx, my, mz), hw = XFORM(Pm, 3, mx, my, mz);
            float pw = 1.0f / (hw + E3_W_EPS);
            float ndcx = hx * pw, ndcy = hy * pw;
            // EWA: T = J * Wr, Sigma2 = T Sigma3 T^T
            float limx = E3_GUARD_BAND * vp.tanfovx, limy = E3_GUARD_BAND * vp.tanfovy;
            float txtz = vx / vz, tytz = vy / vz;
            float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
            float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
            float tz = vz;
            float J00 = vp.focal_x / tz, J02 = -(vp.focal_x * tx) / (tz * tz);
            float J11 = vp.focal_y / tz, J12 = -(vp.focal_y * ty) / (tz * tz);
            float T00 = FMA(J00, V[0], J02 * V[2]), T01 = FMA(J00, V[4], J02 * V[6]), T02 = FMA(J00, V[8], J02 * V[10]);
            float T10 = FMA(J11, V[1], J12 * V[2]), T11 = FMA(J11, V[5], J12 * V[6]), T12 = FMA(J11, V[9], J12 * V[10]);
            float u0 = FMA(S[0], T00, FMA(S[1], T01, S[2] * T02));
            float u1 = FMA(S[1], T00, FMA(S[3], T01, S[4] * T02));
            float u2 = FMA(S[2], T00, FMA(S[4], T01, S[5] * T02));
            float w0 = FMA(S[0], T10, FMA(S[1], T11, S[2] * T12));
            float w1 = FMA(S[1], T10, FMA(S[3], T11, S[4] * T12));
            float w2 = FMA(S[2], T10, FMA(S[4], T11, S[5] * T12));
            float a = FMA(T00, u0, FMA(T01, u1, T02 * u2));
            float b = FMA(T10, u0, FMA(T11, u1, T12 * u2));
            float c = FMA(T10, w0, FMA(T11, w1, T12 * w2));
            a = a + E3_DILATION;
            c = c + E3_DILATION;
            float det = FMA(a, c, -(b * b));
            if (det != 0.0f) {
                float det_inv = 1.0f / det;
                float conx = c * det_inv, cony = -b * det_inv, conz = a * det_inv;
                float mid = 0.5f * (a + c);
                float disc = __builtin_sqrtf(fmaxf(E3_EIGEN_FLOOR, FMA(mid, mid, -det)));
                float lam1 = mid + disc, lam2 = mid - disc;
                int radius = (int)__builtin_ceilf(3.0f * __builtin_sqrtf(fmaxf(lam1, lam2)));
                // ndc2Pix in double, single final rounding (gaussian_renderer/__init__.py:238-241)
                float px = (float)((((double)ndcx + 1.0) * (double)vp.W - 1.0) * 0.5);
                float py = (float)((((double)ndcy + 1.0) * (double)vp.H - 1.0) * 0.5);
                float fr = (float)radius;
                int xmin = clampi((int)((px - fr) / (float)E3_TILE), 0, vp.gx);
                int ymin = clampi((int)((py - fr) / (float)E3_TILE), 0, vp.gy);
                int xmax = clampi((int)((((px + fr) + (float)E3_TILE) - 1.0f) / (float)E3_TILE), 0, vp.gx);
                int ymax = clampi((int)((((py + fr) + (float)E3_TILE) - 1.0f) / (float)E3_TILE), 0, vp.gy);
                if ((xmax - xmin) * (ymax - ymin) != 0) {
                    const float* rgb = rgbv[v];
                    const uint32_t cl = clv[v];
                    rec[3 * q] = make_float4(px, py, conx, cony);
                    rec[3 * q + 1] = make_float4(conz, o_, rgb[0], rgb[1]);
                    // .z: above pmin + 4e-4 alpha >= 1/255 holds whatever the rounding of power / exp (backward's band)
                    // .w: the tile rectangle once more, 8 bits per bound (grids up to 255 x 255 tiles): the binning passes
                    // then need ONE gather per splat (this record) instead of two (bin_kernel)
                    const uint32_t prect = (uint32_t)xmin | ((uint32_t)ymin << 8) | ((uint32_t)xmax << 16) | ((uint32_t)ymax << 24);
                    rec[3 * q + 2] = make_float4(rgb[2], pmin, pmin + 4e-4f, __uint_as_float(prect));
                    clamped[q] = cl;
                    radius_out = radius;
                    rect_out = make_uint2((uint32_t)xmin | ((uint32_t)ymin << 16), (uint32_t)xmax | ((uint32_t)ymax << 16));
                    key_out = __float_as_uint(vz);
                }
            }
        }
        radii[(size_t)v * P + i] = radius_out;
        rect[q] = rect_out;
        key[q] = key_out;          // the depth sort's payload is q itself (identity_payload): no id array written
    }
}

// E3_FLAG_DEFER_COLOR: SH -> RGB of the visible splats as a kernel of its own, launched right before compositing.
// The SH coefficients are 81 % of the parameter bytes; everything in front of the compositing kernel (projection,
// both sorts, binning) does not need them, so a trainer can still be averaging / updating them (previous iteration's
// all-reduce + Adam on another stream) while the next iteration builds its lists.  Same sh_to_rgb() as the
// preprocess kernel: bit-identical colours.
__global__ __launch_bounds__(256) void colour_kernel(int P, int nv, int D, int M, const float* __restrict__ means,
                                                     const float* __restrict__ shs, ViewSet vs, int flags,
                                                     const uint2* __restrict__ rect, float4* __restrict__ rec,
                                                     uint32_t* __restrict__ clamped) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = nv == 1 ? q : q / nv;
    if (i >= P) return;
    const uint2 r = rect[q];
    if (((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16)) == 0u) return;      // culled splat
    const int v = q - i * nv;
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    const bool planar = (flags & E3_FLAG_SH_PLANAR) != 0;
    float rgb[3];
    uint32_t cl = 0;
    sh_to_rgb(D, planar ? shs + i : shs + (size_t)i * M * 3, planar ? (size_t)P : (size_t)1, mx, my, mz, vs.v[v].campos,
              rgb, cl);
    float* rp = reinterpret_cast<float*>(rec + 3 * (size_t)q);
    *reinterpret_cast<float2*>(rp + 6) = make_float2(rgb[0], rgb[1]);       // rec[1].zw
    rp[8] = rgb[2];                                                         // rec[2].x
    clamped[q] = cl;
}

__device__ __forceinline__ uint32_t rect_area(uint2 r) {
    uint32_t w = (r.y & 0xFFFFu) - (r.x & 0xFFFFu), h = (r.y >> 16) - (r.x >> 16);
    return w * h;
}

// ------------------------------------------------------------------------------------ binning
// Conservative "does this Gaussian reach any pixel of this tile" test.  A pixel receives a
// contribution only if alpha = o*exp(power) >= 1/255, i.e. q(d) = A dx^2 + 2B dx dy + C dy^2
// <= 2 ln(255 o).  The minimum of the convex quadratic q over the tile's pixel box is either 0
// (centre inside) or attained on one of the four edges (1-D clamped minimiser).  Instances whose
// minimum exceeds the bound by a safety margin (1e-4 of the summed |terms| + an absolute slack folded
// into `thr` by the caller, orders of magnitude above fp32 rounding) contribute to no pixel, so dropping them leaves image and gradients
// bit-identical while shrinking the sorted lists (~1.8x on the benchmark scene).
__device__ __forceinline__ bool tile_touched(float x0, float y0, float A, float B, float C, float thr, float iC,
                                             float iA, int tx, int ty) {
    // iC = -B / C, iA = -B / A: computed once per Gaussian by the caller (three IEEE divisions per candidate
    // tile were a third of the binning kernels' instruction count)
    if (!(A > 0.0f) || !(C > 0.0f)) return true;   // degenerate / NaN conic: never cull
    const float pxl = (float)(tx * E3_TILE), pyl = (float)(ty * E3_TILE);
    const float dxlo = x0 - (pxl + (float)(E3_TILE - 1)), dxhi = x0 - pxl;
    const float dylo = y0 - (pyl + (float)(E3_TILE - 1)), dyhi = y0 - pyl;
    // q is convex with its minimum (0) at d = 0.  With the centre outside the box the minimum over the box lies on an edge
    // that FACES the centre -- walking from any point of a far edge towards the centre lowers q and crosses a near edge
    // first -- so at most two of the four edges need their 1-D clamped minimiser evaluated (half the instructions of
    // the four-edge form; this test is ~45 % of the binning passes' VALU work).
    const bool out_x = dxlo > 0.0f || dxhi < 0.0f, out_y = dylo > 0.0f || dyhi < 0.0f;
    const float ex = dxlo > 0.0f ? dxlo : dxhi;          // the x-edge nearest to the centre (meaningful if out_x)
    const float ey = dylo > 0.0f ? dylo : dyhi;
    const float B2 = 2.0f * B;
    auto qadj = [&](float dx, float dy) {
        float t0 = A * dx * dx, t1 = B2 * dx * dy, t2 = C * dy * dy;
        return (t0 + t1 + t2) - 1e-4f * (fabsf(t0) + fabsf(t1) + fabsf(t2));
    };
    const float qx = qadj(ex, fminf(dyhi, fmaxf(dylo, iC * ex)));
    const float qy = qadj(fminf(dxhi, fmaxf(dxlo, iA * ey)), ey);
    // centre inside the box in both axes -> 0; an axis whose range contains the centre has no facing edge
    const float qmin = fminf(out_x ? qx : (out_y ? INFINITY : 0.0f), out_y ? qy : (out_x ? INFINITY : 0.0f));
    return !(qmin > thr);
}

// Flattened, wave-cooperative binning.  A wave owns 64 consecutive depth-sorted Gaussians; the
// union of their tile rectangles is walked 64 candidate (Gaussian, tile) items at a time, so lanes
// stay busy whatever the individual splat sizes are.  EMIT=false counts the kept items per wave
// (-> exclusive scan -> total instance count I); EMIT=true replays the identical walk and writes
// (tile id, Gaussian id) compacted with a ballot prefix.  Emission order = depth order of the
// Gaussians, row-major inside a rectangle -- exactly the order the reference's per-Gaussian loop
// produces, so a stable sort on the tile id alone finishes the job.
// With few splats (the first iterations after a point-cloud initialisation: thousands of Gaussians, each covering
// hundreds of tiles) 64 splats per wave would leave most of the chip idle, so a wave then owns only 1 << gshift of
// them (e3_bin_group_shift: at least 8192 waves whenever there are that many splats).
constexpr int BIN_WAVES = 4;

// LDS tables of one binning wave: what the candidate walk needs of its (up to) 64 splats
struct BinTables {
    float4 A[WAVE];       // x, y, conic.x, conic.y
    float4 B[WAVE];       // conic.z, thr, xmin|ymin<<16, width
    float4 D[WAVE];       // -B/C, -B/A, 1/width, first tile id of the splat's view
    uint32_t excl[WAVE];  // exclusive scan of the candidate counts (first candidate item of the splat)
    uint32_t id[WAVE];    // splat ids
    uint32_t cnt[WAVE];   // kept instances per splat (emission)
    uint32_t flag[WAVE];  // candidate walk: "a splat starts at this item of the round"
};
// The tables hold only the splats WITH candidates, in order (rank r = position among them): the walk finds the splat of
// candidate item m as "number of splats that start at or before m", from start flags + a ballot, instead of a 6-step
// binary search through LDS (6 dependent LDS round trips per 64 items: what bounded the walk).  WalkLane: what a lane
// keeps of ITS splat for that (rank, first item, number of ranked splats).
struct WalkLane { uint32_t rank, excl, nnz; bool nonempty; };

// Gathers the records of the wave's splats (depth-sorted positions s of `order`), derives the candidate rectangles and
// fills the wave's tables.  Returns the number of candidate (splat, tile) items of the wave (wave-uniform).
// MODE 0: gather and derive.  MODE 1 (count pass): also store the derived 32-byte bin record at the splat's SORTED position
// (`binrec`, coalesced).  MODE 2 (emission pass): load that record instead of gathering -- the gather of a 48-byte record
// at a random address per splat (1.25 lines of 128 bytes) is what bounds a binning pass, and the emission pass repeats
// the count pass's walk over exactly the same splats.
template <int MODE>
__device__ __forceinline__ uint32_t bin_load_tables(BinTables& T, int lane, bool mine, int s, int nviews, int ntiles,
                                                    const uint32_t* __restrict__ order, const uint2* __restrict__ rect,
                                                    int packed_rect, const float4* __restrict__ rec, int cull,
                                                    float4* __restrict__ binrec, WalkLane& wl) {
    uint32_t n = 0, g = 0;
    float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
    if (MODE == 2) {
        if (mine) {
            g = order[s];
            a = binrec[2 * (size_t)s];
            b = binrec[2 * (size_t)s + 1];
            const uint32_t wh = __float_as_uint(b.w);
            n = (wh & 0xFFFFu) * (wh >> 16);
            b.w = __uint_as_float(wh & 0xFFFFu);
        }
    } else if (mine) {
        g = order[s];
        // ONE gather of the 48-byte record delivers the rectangle too (8 bits per bound in its spare word, grids up to
        // 255 x 255 tiles; the 16-bit array otherwise): the rectangle array used to cost a second 128-byte line per
        // splat and pass.
        uint2 r;
        a = rec[3 * (size_t)g];
        const float4 t = rec[3 * (size_t)g + 1];
        if (packed_rect) {
            const uint32_t p = __float_as_uint(rec[3 * (size_t)g + 2].w);
            r = make_uint2((p & 255u) | (((p >> 8) & 255u) << 16), ((p >> 16) & 255u) | ((p >> 24) << 16));
        } else {
            r = rect[g];
        }
        uint32_t w = (r.y & 0xFFFFu) - (r.x & 0xFFFFu), h = (r.y >> 16) - (r.x >> 16);
        n = w * h;
        if (n) {
            // 2 ln(255 o) with the safety margin folded in; o <= 0 -> -inf -> nothing kept
            // margins: see tile_touched(); the Lambda term bounds the rounding of the compositing kernels'
            // own `power` at any pixel of the tile (|terms| <= 2(|terms at the minimiser| + Lambda*15^2*2))
            float thr = 2.0f * logf(255.0f * t.y);
            thr = thr + fabsf(thr) * 1e-4f + 1e-3f + 2e-3f * (fabsf(a.z) + 2.0f * fabsf(a.w) + fabsf(t.x));
            uint32_t xy0 = r.x;
            // Candidate rectangle: the reference walks every tile of the 3-sigma square; tile_touched() then keeps those the
            // ellipse q <= thr can reach.  Nearly half of the candidates fail that test, and every one costs the binning
            // walk a search + test.  The axis-aligned box of the (inflated) ellipse is known in closed form --
            // |dx| <= sqrt(thr' C / det), |dy| <= sqrt(thr' A / det) -- so only its tiles are walked.  thr' carries the
            // relative slack tile_touched() subtracts (1e-4 of the summed |terms|, bounded over the rectangle's pixels) and the
            // box a pixel of margin: every tile outside it fails tile_touched(), i.e. the kept set is unchanged
            // (tests/test_hip_parity.py::test_tight_candidate_box_keeps_the_same_instances).
            if (cull == 1) {                            // (cull == 3: exact culling without the tight box, for the fuzz test)
                const float A = a.z, B = a.w, C = t.x, det = A * C - B * B;
                if (A > 0.0f && C > 0.0f && det > 0.0f) {
                    const int x0r = (int)(r.x & 0xFFFFu), y0r = (int)(r.x >> 16), x1r = (int)(r.y & 0xFFFFu), y1r = (int)(r.y >> 16);
                    // largest |dx|, |dy| between the centre and a pixel of the (grid-clamped) rectangle: for a centre off
                    // the screen that is more than the rectangle's own extent
                    const float ex = fmaxf(fabsf(a.x - 16.0f * (float)x0r), fabsf(a.x - (16.0f * (float)x1r - 1.0f)));
                    const float ey = fmaxf(fabsf(a.y - 16.0f * (float)y0r), fabsf(a.y - (16.0f * (float)y1r - 1.0f)));
                    const float thr2 = thr + 2e-4f * (A * ex * ex + C * ey * ey) * 1.01f + 1e-3f;
                    if (thr2 > 0.0f) {
                        const float hx = __builtin_sqrtf(thr2 * C / det) * 1.001f + 1.0f;
                        const float hy = __builtin_sqrtf(thr2 * A / det) * 1.001f + 1.0f;
                        // pixels of tile t span [16 t, 16 t + 15]
                        const int tx0 = max(x0r, (int)__builtin_floorf((a.x - hx - 15.0f) * (1.0f / 16.0f))),
                                  tx1 = min(x1r, (int)__builtin_floorf((a.x + hx) * (1.0f / 16.0f)) + 1),
                                  ty0 = max(y0r, (int)__builtin_floorf((a.y - hy - 15.0f) * (1.0f / 16.0f))),
                                  ty1 = min(y1r, (int)__builtin_floorf((a.y + hy) * (1.0f / 16.0f)) + 1);
                        if (tx1 > tx0 && ty1 > ty0) {
                            w = (uint32_t)(tx1 - tx0); h = (uint32_t)(ty1 - ty0);
                            xy0 = (uint32_t)tx0 | ((uint32_t)ty0 << 16);
                        } else { w = 0u; h = 0u; }
                        n = w * h;
                    } else n = 0u;                      // 255 o < 1 everywhere (with slack): nothing can be kept
                }
            }
            b = make_float4(t.x, thr, __uint_as_float(xy0), __uint_as_float(w));
        }
        if (MODE == 1) {
            binrec[2 * (size_t)s] = a;
            binrec[2 * (size_t)s + 1] = make_float4(b.x, b.y, b.z, __uint_as_float(n ? (w | (h << 16)) : 0u));
        }
    }
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    // (made scalar explicitly: a trip count that sits in a VGPR makes the candidate loop a divergent loop)
    const uint32_t total = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
    const unsigned long long nzmask = __ballot(n != 0u);
    const uint32_t r = (uint32_t)__popcll(nzmask & ((1ull << lane) - 1ull));
    wl.rank = r; wl.excl = incl - n; wl.nnz = (uint32_t)__popcll(nzmask); wl.nonempty = n != 0u;
    T.cnt[lane] = 0;
    if (n != 0u) {
        const uint32_t tbase = nviews > 1 ? (g % (uint32_t)nviews) * (uint32_t)ntiles : 0u;
        const float fw = (float)__float_as_uint(b.w);
        T.A[r] = a; T.B[r] = b; T.excl[r] = incl - n; T.id[r] = g;
        T.D[r] = make_float4(-a.w / b.x, -a.w / a.z, 1.0f / fw, __uint_as_float(tbase));
    }
    wave_sync();
    return total;
}

// The flattened candidate walk of one wave: 64 candidate (splat, tile) items at a time.  EMIT = false counts the kept
// items; EMIT = true replays the identical walk and writes (tile id, splat id) compacted with a ballot prefix at
// out_base + (position among the wave's kept items).  Emission order = depth order of the splats, row-major inside a
// rectangle -- exactly the order the reference's per-Gaussian loop produces, so a stable sort on the tile id alone
// finishes the job.  Returns the number of kept items (wave-uniform).
template <bool EMIT>
__device__ __forceinline__ uint32_t bin_walk(BinTables& T, const WalkLane& wl, int lane, uint32_t total, int gx, int cull,
                                             uint32_t out_base, void* __restrict__ keys, int keys16,
                                             uint32_t* __restrict__ emit_gid, uint8_t* __restrict__ touched) {
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint64_t le_mask = lt_mask | (1ull << lane);
    uint32_t count = 0;
    uint32_t started = 0;                       // (scalar) ranked splats whose first item lies before this round
    for (uint32_t m0 = 0; m0 < total; m0 += WAVE) {
        const uint32_t m = m0 + lane;
        const bool active = m < total;
        // the splat of item m = (number of ranked splats that start at or before m) - 1: lane r flags the item its splat
        // starts at when that item belongs to this round; one ballot of the flags, one popcount per lane.  (DS operations
        // of a wave retire in order: clear, set and read need no waits in between.)
        T.flag[lane] = 0u;
        wave_sync();
        {
            const uint32_t p = wl.excl - m0;    // (unsigned: a start before the round wraps to a large number)
            if (wl.nonempty && p < (uint32_t)WAVE) T.flag[p] = 1u;
        }
        wave_sync();
        const unsigned long long starts = __ballot(T.flag[lane] != 0u);
        const int j = (int)(started + (uint32_t)__popcll(starts & le_mask)) - 1;
        started += (uint32_t)__popcll(starts);
        const uint32_t excl = T.excl[j];
        const float4 A4 = T.A[j], B4 = T.B[j], D4 = T.D[j];
        const uint32_t k = m - excl, w = __float_as_uint(B4.w), xy0 = __float_as_uint(B4.z);
        uint32_t row = (uint32_t)(((float)k + 0.5f) * D4.z);       // w, k < 2^24: exact after the fix-up
        if (row * w > k) --row;
        if ((row + 1) * w <= k) ++row;
        const int tx = (int)((xy0 & 0xFFFFu) + (k - row * w)), ty = (int)((xy0 >> 16) + row);
        const bool keep = active && (!cull || tile_touched(A4.x, A4.y, A4.z, A4.w, B4.x, B4.y, D4.x, D4.y, tx, ty));
        const uint64_t mask = __ballot(keep);
        if (EMIT && keep) {
            // emission position = depth order (what the stable tile sort preserves).  It is also the instance's
            // SLOT: where backward parks its gradient record; the kept instances of one splat are contiguous.
            // The sort payload is this position itself, so no payload array is written (identity_payload)
            const uint32_t pos = out_base + count + (uint32_t)__popcll(mask & lt_mask);
            const uint32_t tile_id = __float_as_uint(D4.w) + (uint32_t)ty * (uint32_t)gx + (uint32_t)tx;
            if (keys16) reinterpret_cast<uint16_t*>(keys)[pos] = (uint16_t)tile_id;      // (<= 65536 tiles: 16-bit sort keys)
            else reinterpret_cast<uint32_t*>(keys)[pos] = tile_id;
            emit_gid[pos] = T.id[j];
            touched[pos] = 0;                       // (render_fwd_kernel sets it for the instances it evaluates)
            atomicAdd(&T.cnt[j], 1u);
        }
        count += (uint32_t)__popcll(mask);
    }
    return count;
}

// run of each splat of the wave, stored at its DEPTH-SORTED position: (first slot, kept instances)
__device__ __forceinline__ void bin_store_runs(BinTables& T, const WalkLane& wl, int lane, bool mine, int s,
                                               uint32_t out_base, uint2* __restrict__ run_sorted) {
    wave_sync();
    const uint32_t c = wl.nonempty ? T.cnt[wl.rank] : 0u;      // (the tables are indexed by rank)
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (mine) run_sorted[s] = make_uint2(out_base + inc - c, c);
}

// Two passes: EMIT=false counts the kept items per wave (-> inclusive scan -> total instance count I), EMIT=true replays
// the walk and writes.  (A single launch -- gather once, count, chain the counts across the workgroups with decoupled
// look-back, emit from the tables still in LDS -- was measured slower: the chain serialises the workgroups,
// profiles/EXPERIMENTS.md.)
template <bool EMIT>
__global__ __launch_bounds__(BIN_WAVES * WAVE) void bin_kernel(int P, int gshift, int nviews, int ntiles,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ nvis /* [1]: length of `order` */,
                                                              const uint2* __restrict__ rect, int packed_rect,
                                                              const float4* __restrict__ rec, int gx, int cull,
                                                              const uint32_t* __restrict__ wave_offsets,
                                                              uint32_t* __restrict__ wave_counts,
                                                              void* __restrict__ keys, int keys16,
                                                              uint32_t* __restrict__ emit_gid,
                                                              uint2* __restrict__ run_sorted,
                                                              uint8_t* __restrict__ touched,
                                                              const uint32_t* __restrict__ total_dev, uint32_t capacity,
                                                              float4* __restrict__ binrec /* or null: both passes gather */) {
    __shared__ BinTables tabs[BIN_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * BIN_WAVES + wave;
    const int s = (gw << gshift) + lane;
    if ((gw << gshift) >= P) return;
    // `order` holds the splats the projection kept, by depth: the depth sort dropped the culled ones (their count is
    // only known on the device).  Waves behind the end have nothing to walk.
    const int nv = (int)__builtin_amdgcn_readfirstlane(*nvis);
    // (buffers sized before the count was known: more instances than they hold -> nothing is written, see
    // e3_forward_finish_impl)
    if (EMIT && total_dev && *total_dev > capacity) return;
    if ((gw << gshift) >= nv) {
        if (!EMIT && lane == 0) wave_counts[gw] = 0u;
        return;
    }
    const bool mine = lane < (1 << gshift) && s < nv;
    BinTables& T = tabs[wave];
    uint32_t total;
    WalkLane wl;
    if (!binrec) total = bin_load_tables<0>(T, lane, mine, s, nviews, ntiles, order, rect, packed_rect, rec, cull, nullptr, wl);
    else total = bin_load_tables<EMIT ? 2 : 1>(T, lane, mine, s, nviews, ntiles, order, rect, packed_rect, rec, cull, binrec, wl);
    const uint32_t out_base = EMIT ? __builtin_amdgcn_readfirstlane(wave_offsets[gw]) : 0u;
    const uint32_t count = bin_walk<EMIT>(T, wl, lane, total, gx, cull, out_base, keys, keys16, emit_gid, touched);
    if (!EMIT) {
        if (lane == 0) wave_counts[gw] = count;
    } else {
        bin_store_runs(T, wl, lane, mine, s, out_base, run_sorted);
    }
}

// After the stable tile sort the [start, end) of each tile comes out of the sort itself (scan_sort.hip: the last
// pass of a two-pass sort derives the ranges from its scanned histogram).  The sort moved (tile id, emission index)
// pairs; the compositing kernels translate emission index -> Gaussian id themselves (`emit_gid`, a small L2-resident
// gather that rides in their staging pipeline) and backward stores each instance's gradient record at its EMISSION
// position, where the records of one Gaussian are contiguous -- that turns the reference's float atomics into
// plain stores plus a per-Gaussian run reduction (deterministic, and ~2x faster: the atomics were 44 % of the
// compositing backward on this chip).

// ------------------------------------------------------------------------------------ compositing
// One wave per 16x16 tile.  Lane l owns column (l & 15) and rows (l >> 4) + 4k, k = 0..3, so the
// x-dependent half of the quadratic form is shared by its four pixels.  Each round the wave
// gathers 64 list entries (one per lane) into its private LDS slice and every lane then walks
// them with broadcast ds_read_b128; the next round's gathers are in flight meanwhile.
#ifndef E3_RENDER_WAVES
#define E3_RENDER_WAVES 2      // waves (= tiles) per workgroup: 4 -> 2 measured -2.4 % on render_fwd_kernel (1: the same); wave slots are refilled at a finer grain
#endif
constexpr int RENDER_WAVES = E3_RENDER_WAVES;

// Launch order of the compositing kernels: longest lists first (LPT).  Workgroups are dispatched in index
// order as slots free up, so heavy tiles start at t = 0 and the short ones fill the tail; without this the
// last long tiles run alone on their SIMDs at single-wave (latency-bound) speed.  Single workgroup:
// block max -> 1024-bucket histogram in LDS -> scan -> scatter.  Order inside a bucket is arbitrary, which
// only affects scheduling, never results.
// The last pass of the tile sort leaves tiles without an instance with an empty range start == end (the position
// where their run would sit); the documented state -- what the reference's identifyTileRanges leaves, and what the
// parity tests compare -- is (0, 0).  The forward ordering kernel is the first reader of the ranges and rewrites them.
__device__ __forceinline__ uint2 normalise_empty_range(uint2* __restrict__ ranges, int t) {
    uint2 r = ranges[t];
    if (r.x == r.y && r.x != 0u) { r = make_uint2(0u, 0u); ranges[t] = r; }
    return r;
}

__global__ __launch_bounds__(1024) void tile_order_kernel(int ntiles, uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ work,
                                                          uint32_t* __restrict__ order) {
    auto key = [&](int t) -> uint32_t { if (work) return work[t]; uint2 r = ranges[t]; return r.y - r.x; };
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t m = 0;
    if (!work)      // (forward order: this kernel is the first reader of the ranges the tile sort wrote)
        for (int t = tid; t < ntiles; t += 1024) normalise_empty_range(ranges, t);
    for (int t = tid; t < ntiles; t += 1024) m = max(m, key(t));
    m = wave_max_u32(m);
    if (lane == 0) wsum[wave] = m;
    hist[tid] = 0;
    __syncthreads();
    if (tid == 0) { uint32_t x = 0; for (int w = 0; w < 16; ++w) x = max(x, wsum[w]); smax = x; }
    __syncthreads();
    const uint32_t width = smax / 1024u + 1u;
    for (int t = tid; t < ntiles; t += 1024) atomicAdd(&hist[1023u - min(1023u, key(t) / width)], 1u);
    __syncthreads();
    // exclusive scan of hist (one bucket per thread)
    uint32_t v = hist[tid], inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    hist[tid] = woff + inc - v;
    __syncthreads();
    for (int t = tid; t < ntiles; t += 1024) {
        uint32_t pos = atomicAdd(&hist[1023u - min(1023u, key(t) / width)], 1u);
        order[pos] = (uint32_t)t;
    }
}

// Same, for up to 32 K tiles (3 x 1080p = 24 480): every thread keeps its (up to) 32 keys in registers, so global
// memory is read once instead of three times and the three phases only touch LDS (30 -> ~10 us; this kernel sits
// alone on the GPU twice per iteration).
constexpr int ORDER_ITEMS = 32;
__global__ __launch_bounds__(1024) void tile_order_reg_kernel(int ntiles, uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ work,
                                                              uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t key[ORDER_ITEMS];
    uint32_t m = 0;
    // unconditional loads with a clamped index, all in flight together: a load under `if (t < ntiles)` waits for its
    // own round trip in every unrolled iteration (32 dependent memory latencies in a kernel that runs alone on the GPU)
    if (work) {
#pragma unroll
        for (int i = 0; i < ORDER_ITEMS; ++i) {
            const int t = tid + i * 1024;
            key[i] = work[t < ntiles ? t : ntiles - 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < ORDER_ITEMS; ++i) {
            const int t = tid + i * 1024;
            const uint2 r = ranges[t < ntiles ? t : ntiles - 1];
            key[i] = r.y - r.x;
            if (t < ntiles && r.x == r.y && r.x != 0u) ranges[t] = make_uint2(0u, 0u);      // normalise_empty_range
        }
    }
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i) {
        if (tid + i * 1024 >= ntiles) key[i] = 0;
        m = max(m, key[i]);
    }
    m = wave_max_u32(m);
    if (lane == 0) wsum[wave] = m;
    hist[tid] = 0;
    __syncthreads();
    if (tid == 0) { uint32_t x = 0; for (int w = 0; w < 16; ++w) x = max(x, wsum[w]); smax = x; }
    __syncthreads();
    const uint32_t width = smax / 1024u + 1u;
    // bucket = 1023 - min(1023, key / width) with the division as a multiplication by the rounded-up reciprocal
    // (exact for the 32-bit keys and widths that occur: checked against the division below in debug builds is not
    // needed -- an off-by-one bucket only changes the launch order, never a result)
    const float inv_width = 1.0f / (float)width;
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i) {
        const uint32_t b = min(1023u, (uint32_t)((float)key[i] * inv_width));
        key[i] = 1023u - b;                                  // from here on key[] holds the bucket
        if (tid + i * 1024 < ntiles) atomicAdd(&hist[key[i]], 1u);
    }
    __syncthreads();
    uint32_t v = hist[tid], inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    hist[tid] = woff + inc - v;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i) {
        const int t = tid + i * 1024;
        if (t < ntiles) {
            uint32_t pos = atomicAdd(&hist[key[i]], 1u);
            order[pos] = (uint32_t)t;
        }
    }
}

// XCD-partitioned variant: bucket = region * 128 + (127 - work / width); slot of the k-th heaviest tile of region r is
// (k / 4) * 32 + r * 4 + (k % 4), i.e. the 4-tile workgroup w = slot / 4 belongs to region w % 8.  Unused slots hold ~0u.
__global__ __launch_bounds__(1024) void tile_order_xcd_kernel(int ntiles, int nslots, int tiles_per_view, int gx, int B,
                                                              uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ work,
                                                              uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t start[8];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t key[ORDER_ITEMS];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i) {
        const int t = tid + i * 1024;
        uint32_t k = 0;
        if (t < ntiles) {
            if (work) k = work[t];
            else { const uint2 r = normalise_empty_range(ranges, t); k = r.y - r.x; }
        }
        key[i] = k;
        m = max(m, k);
    }
    m = wave_max_u32(m);
    if (lane == 0) wsum[wave] = m;
    hist[tid] = 0;
    for (int j = tid; j < nslots; j += 1024) order[j] = 0xFFFFFFFFu;
    __syncthreads();
    if (tid == 0) { uint32_t x = 0; for (int w = 0; w < 16; ++w) x = max(x, wsum[w]); smax = x; }
    __syncthreads();
    const uint32_t width = smax / 128u + 1u;
    auto bucket = [&](int t, uint32_t k) {
        return (uint32_t)e3_xcd_region(t, tiles_per_view, gx, B) * 128u + (127u - min(127u, k / width));
    };
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i)
        if (tid + i * 1024 < ntiles) atomicAdd(&hist[bucket(tid + i * 1024, key[i])], 1u);
    __syncthreads();
    uint32_t v = hist[tid], inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    hist[tid] = woff + inc - v;
    if ((tid & 127) == 0) start[tid >> 7] = woff + inc - v;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ORDER_ITEMS; ++i) {
        const int t = tid + i * 1024;
        if (t < ntiles) {
            const uint32_t b = bucket(t, key[i]);
            const uint32_t k = atomicAdd(&hist[b], 1u) - start[b >> 7];
            order[(k >> 2) * 32u + (b >> 7) * 4u + (k & 3u)] = (uint32_t)t;
        }
    }
}

static bool bin_handoff() {      // E3DGS_BIN_HANDOFF=0: both binning passes gather the splat records (A/B switch)
    static const bool v = [] { const char* e = getenv("E3DGS_BIN_HANDOFF"); return !(e && e[0] == '0'); }();
    return v;
}
static int xcd_block() {
    static const int v = [] { const char* e = getenv("E3DGS_XCD_BLOCK"); return e ? atoi(e) : 0; }();
    return v;
}

static bool lpt_lists_enabled() {      // E3DGS_LPT_LISTS=0: ordering kernels as before (A/B switch)
    static const bool v = [] { const char* e = getenv("E3DGS_LPT_LISTS"); return !(e && e[0] == '0'); }();
    return v;
}
bool e3_use_lpt_lists(int ntiles, int num_rendered, int P) {
    if (!lpt_lists_enabled() || xcd_block() > 0 || num_rendered <= 0 || P <= 0) return false;
    // a TWO-pass tile sort (257 ... 65536 tiles per call) derives the ranges -- and the forward order -- in its last pass
    return ntiles > 256 && ntiles <= 65536;
}

// Returns the number of launch slots (== ntiles unless the XCD-partitioned order is on).
int launch_tile_order(int ntiles, int tiles_per_view, int gx, uint2* ranges, const uint32_t* work, uint32_t* order,
                      hipStream_t s) {
    const int B = xcd_block();
    if (B > 0 && ntiles <= ORDER_ITEMS * 1024 && ntiles >= 64) {
        static thread_local int c_key[4] = {0, 0, 0, 0}, c_slots = 0;
        if (c_key[0] != ntiles || c_key[1] != tiles_per_view || c_key[2] != gx || c_key[3] != B) {
            int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0;
            for (int t = 0; t < ntiles; ++t) ++cnt[e3_xcd_region(t, tiles_per_view, gx, B)];
            for (int r = 0; r < 8; ++r) mx = cnt[r] > mx ? cnt[r] : mx;
            c_key[0] = ntiles; c_key[1] = tiles_per_view; c_key[2] = gx; c_key[3] = B;
            c_slots = ((mx + 3) / 4) * 32;
        }
        if (c_slots <= 2 * ntiles + 64) {
            tile_order_xcd_kernel<<<dim3(1), dim3(1024), 0, s>>>(ntiles, c_slots, tiles_per_view, gx, B, ranges, work, order);
            return c_slots;
        }
    }
    if (ntiles <= ORDER_ITEMS * 1024)
        tile_order_reg_kernel<<<dim3(1), dim3(1024), 0, s>>>(ntiles, ranges, work, order);
    else
        tile_order_kernel<<<dim3(1), dim3(1024), 0, s>>>(ntiles, ranges, work, order);
    return ntiles;
}

unsigned long long* g_trace = nullptr;   // debug: per-tile {start, end, entries, hw_id} (tools/trace_fwd.py)

// Strip pre-test of the forward compositing kernel, LANE-PARALLEL: at staging time lane j holds the record of list entry j
// of the round, and decides for ITS entry which of the tile's four 16x4 pixel strips the entry can contribute to at all
// -- i.e. whether some LIVE pixel of the strip can have power >= pmin (alpha >= 1/255 needs that, preprocess_kernel).  The
// entry loop then skips a dead strip with one scalar bit test, where it used to spend 6 VALU instructions per (entry,
// strip) on every one of the wave's 64 lanes to find the same thing out: 4 x 6 wave-instructions per entry become
// ~200 / 64 = 3.
// "Live pixel": most of the entries a tile walks lie behind pixels that have already saturated, so the test runs against
// the bounding box of the strip's live pixels at the start of the round (columns [cmin, cmax] of the lane mask alive[k],
// rows with a live pixel -- scalar bit arithmetic, once per round), not against the whole strip.  Per pixel row r (dy
// fixed, exact) the maximum of the concave quadratic
//   power(dx) = -C dy^2 / 2 - dx (A dx / 2 + B dy)      over the continuous column range dx in [dxlo, dxhi]
// is attained at clamp(-B dy / A); a strip is live when one of its live rows reaches pmin - margin.  The margin covers
// the rounding of this evaluation AND of the kernel's own `power` (a few ulp of the largest term each; bounded once per
// entry over the tile's box, times a safety factor of ~5): a strip is only ever skipped when no live pixel of it can pass
// the kernel's alpha >= 1/255 test, so image, final_T and n_contrib are unchanged bit for bit.  Conics that are not safely
// concave in dx (A <= 0, NaN) keep every strip.
__device__ __forceinline__ void strip_pretest(const float4 ra, const float4 rb, const float4 rc, float X0, float Y0,
                                              const unsigned long long alive[4], unsigned long long m[4],
                                              unsigned long long& unsafe) {
    const float A = ra.z, B = ra.w, C = rb.x, pmin = rc.y;
    // Entries whose two per-pixel guards are provably no-ops -- the bulk -- skip them (render_fwd_body, E3_FWD_STRIP):
    //  * `power > 0` (rejected by the reference) cannot come out of the kernel's arithmetic for a conic that is safely
    //    positive definite: power = -(0.5 a + 0.5 c + b) (a = A dx^2, b = B dx dy, c = C dy^2) is evaluated with an error
    //    below 3 u (0.5 a + 0.5 c + |b|) <= 3 u lmax r^2 (u = 2^-24) against a true value <= -0.5 lmin r^2, so the
    //    computed sign is right once lmin / lmax > 6 u; det / tr^2 <= lmin / lmax, and 1e-5 leaves a factor of 28 (and
    //    covers the rounding of det itself, <= u tr^2 / 2).  r = 0 gives +-0, which is not > 0.
    //  * min(0.99, o G) is o G when o <= 0.99: G = exp_det(power <= 0) <= 1 (p(f) = fma(q, f, 1) with q > 0 for the
    //    reduced argument f <= 0; 2^n p <= p(0.5) / 2 < 1 below that).
    // NaNs fail the comparisons: unsafe.
    const float det = A * C - B * B, tr = A + C;
    unsafe = __builtin_amdgcn_ballot_w64(!(det > 1e-5f * (tr * tr)) || !(A > 0.0f) || !(rb.y <= E3_ALPHA_CLAMP));
    const float xr = ra.x - X0;                              // dx = xr - column
    const float y0r = ra.y - Y0;                             // dy of row r = y0r - r
    const float nBA = -B / A;
    const float hA = 0.5f * A, hC = 0.5f * C;
    const float DX = fmaxf(fabsf(xr), fabsf(xr - 15.0f)), DY = fmaxf(fabsf(y0r), fabsf(y0r - 15.0f));
    const float margin = 2e-6f * (hA * DX * DX + fabsf(B) * DX * DY + hC * DY * DY) + 1e-6f;
    const float thr = pmin - margin;
    const bool keep_all = !(A > 0.0f) || !(thr == thr) || !(fabsf(nBA) < 3e38f);
    const unsigned long long all = __builtin_amdgcn_ballot_w64(keep_all);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long al = alive[k];
        if (al == 0ull) { m[k] = 0ull; continue; }           // (wave-uniform: every pixel of the strip has finished)
        // columns that still hold a live pixel in any of the strip's rows (lane l = row (l >> 4), column (l & 15))
        const uint32_t cols = (uint32_t)((al | (al >> 16) | (al >> 32) | (al >> 48)) & 0xFFFFull);
        const float dxlo = xr - (float)(31 - __builtin_clz(cols)), dxhi = xr - (float)__builtin_ctz(cols);
        float pk = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (((al >> (16 * r)) & 0xFFFFull) == 0ull) continue;      // (wave-uniform) no live pixel in this row
            const float dy = y0r - (float)(4 * k + r);
            const float dxc = __builtin_amdgcn_fmed3f(dy * nBA, dxlo, dxhi);
            const float u = FMA(hA, dxc, B * dy);
            const float p = FMA(-(hC * dy), dy, -(dxc * u));
            pk = fmaxf(pk, p);
        }
        m[k] = __builtin_amdgcn_fcmpf(pk, thr, 3 /* OGE */) | all;
    }
}

#ifndef E3_FWD_ASM
#define E3_FWD_ASM 1
#endif
#ifndef E3_FWD_GUARDS
#define E3_FWD_GUARDS 1
#endif
#ifndef E3_FWD_WAVES
#define E3_FWD_WAVES 7      // 72 VGPRs: the lane-parallel strip pre-test at the top of a round must not cost the loop a wave per SIMD
#endif
// FAST (E3DGS_FLAG_FAST_EXP, tolerance mode): G = v_exp_f32(power log2 e) -- 2 instead of 10 VALU instructions on the
// most executed path -- instead of the bit-reproducible polynomial.  Everything in front of compositing (radii, lists,
// ranges) is untouched; alpha moves by <= 1 ulp, which can flip the alpha >= 1/255 / T < 1e-4 decisions at a handful of
// pixels per frame (tests/test_hip_parity.py::test_fast_exp_mode counts them).
template <bool FAST>
__device__ __forceinline__ void render_fwd_body(
    unsigned long long* __restrict__ trace, int ntiles, int tiles_per_view, const uint32_t* __restrict__ order, int gx,
    int W, int H, const uint2* __restrict__ ranges, const uint32_t* __restrict__ perm,
    const uint32_t* __restrict__ emit_gid,
    const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_work, uint8_t* __restrict__ strip_mask,
    uint8_t* __restrict__ touched /* per slot, see BinningState */,
    uint32_t* __restrict__ lpt_cnt /* order == NULL: launch orders as per-class lists (ImageState::lpt_*) */,
    uint32_t* __restrict__ lpt_list, uint32_t lpt_cap) {
    __shared__ float4 sA[RENDER_WAVES][WAVE];
    __shared__ float4 sB[RENDER_WAVES][WAVE];
    __shared__ float4 sC[RENDER_WAVES][WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * RENDER_WAVES + wave;
    if (unit >= ntiles) return;                 // ntiles = launch slots (holes of an XCD-partitioned order hold ~0u)
    // (wave-uniform values are made scalar explicitly: the compiler cannot see that `wave` is uniform, and a loop whose
    // trip count sits in a VGPR is compiled as a divergent loop -- exec-mask bookkeeping per iteration, and every
    // scalar the loop carries copied to VGPRs at the latch)
    // global tile id: view * tiles_per_view + local tile
    const int tile = __builtin_amdgcn_readfirstlane(order ? (int)order[unit] : lpt_lookup(lpt_cnt, lpt_list, lpt_cap, (uint32_t)unit, lane));
    if (tile < 0) return;
    const unsigned long long t_start = trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = trace ? __builtin_readcyclecounter() : 0ull;     // s_memtime: shader cycles
    int processed = 0;
    const int view = tile / tiles_per_view, ltile = tile - view * tiles_per_view;
    const int tx = ltile % gx, ty = ltile / gx;
    const int px = tx * E3_TILE + (lane & 15);
    const int py0 = ty * E3_TILE + (lane >> 4);
    const float pfx = (float)px;
    const float tile_x0 = (float)(tx * E3_TILE), tile_y0 = (float)(ty * E3_TILE);
    // Which pixels are still live is wave-level state: alive[k] holds, as a lane mask in an SGPR pair, the pixels of
    // strip k that are inside the image and have not reached T < 1e-4.  The strip test ANDs it with one compare
    // (no per-pixel "T > 0" compare per entry and strip), and a finished pixel simply leaves the mask.
    float pfy[4], T[4], C0[4], C1[4], C2[4];
    uint32_t last[4];
    bool inside[4];
    unsigned long long alive[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int py = py0 + 4 * k;
        pfy[k] = (float)py;
        inside[k] = (px < W) && (py < H);
        T[k] = 1.0f;
        alive[k] = __builtin_amdgcn_ballot_w64(inside[k]);
        C0[k] = C1[k] = C2[k] = 0.0f; last[k] = 0;
        // (opaque to the compiler: it would otherwise re-materialise the int -> float conversion in every strip test of
        // every entry to save four registers -- 4 of the ~50 VALU instructions per entry)
        asm volatile("" : "+v"(pfy[k]));
    }
    uint2 range = ranges[tile];
    range.x = __builtin_amdgcn_readfirstlane(range.x); range.y = __builtin_amdgcn_readfirstlane(range.y);
    const int n = (int)(range.y - range.x);
    int live_strips = 0;      // wave-uniform (SALU): (entry, 16x4 strip) pairs that were evaluated

    // staging pipeline, three deep: round r+2's emission indices, round r+1's Gaussian ids (emit_gid gather) and
    // round r+1's records are all in flight under round r's math, so each round costs one exposed memory round
    // trip at most (at the tile start) instead of index -> id -> record
    float4 ra = make_float4(0, 0, 0, 0), rb = make_float4(0, 0, 0, 0), rc = make_float4(0, 0, 0, 0);
    uint32_t id_next = 0, e_next2 = 0;
    if (lane < n) {
        const uint32_t id = emit_gid[perm[range.x + lane]];
        ra = rec[3 * (size_t)id]; rb = rec[3 * (size_t)id + 1]; rc = rec[3 * (size_t)id + 2];
    }
    if (WAVE + lane < n) id_next = emit_gid[perm[range.x + WAVE + lane]];
    if (2 * WAVE + lane < n) e_next2 = perm[range.x + 2 * WAVE + lane];
    for (int base = 0; base < n; base += WAVE) {
        if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0ull) break;
        const int cnt = min(WAVE, n - base);
        processed += cnt;
        sA[wave][lane] = ra; sB[wave][lane] = rb; sC[wave][lane] = rc;
        // which strips each of the round's entries can contribute to at all (lane j decides for entry j; bit j of gm[k]):
        // tested against the live pixels' bounding box; a strip whose pixels have all finished drops out of gm for good
        unsigned long long gm[4], um;
        strip_pretest(ra, rb, rc, tile_x0, tile_y0, alive, gm, um);
        wave_sync();
        if (base + WAVE + lane < n) {
            ra = rec[3 * (size_t)id_next]; rb = rec[3 * (size_t)id_next + 1]; rc = rec[3 * (size_t)id_next + 2];
        }
        if (base + 2 * WAVE + lane < n) id_next = emit_gid[e_next2];
        if (base + 3 * WAVE + lane < n) e_next2 = perm[range.x + base + 3 * WAVE + lane];
        // Which 16x4 strips evaluated which entry lets the backward walk skip its own per-strip liveness tests (and
        // whole entries): an entry that reached no pixel of a strip here (all its pixels finished, or below 1/255)
        // composited nothing there.  Bit j of sm[k] (an SGPR pair per strip) = entry j of this round evaluated strip k:
        // one scalar bit-set per evaluated strip, and one byte per entry stored per round.
        unsigned long long sm[4] = {0ull, 0ull, 0ull, 0ull};
        float exp_k1 = 0.009671698324382305f;       // second coefficient of exp_det's polynomial, in a VGPR for v_fmamk (asm strip)
        asm volatile("" : "+v"(exp_k1));
        // 1-based list position of the entry, kept in a VECTOR register on purpose: the select that records a pixel's
        // last contributor needs it there, and a scalar copy is re-materialised with a v_mov in every live strip
        uint32_t contributor = (uint32_t)base;
        unsigned long long jbit = 1ull;             // 1 << j, shifted along (a 64-bit shift amount drags a 64-bit counter)
        auto entry = [&](const int j) __attribute__((always_inline)) {
            const float4 a = sA[wave][j];
            const float4 b = sB[wave][j];
            const float c_x = sC[wave][j].x;
            contributor += 1u;
            asm volatile("" : "+v"(contributor));
            const float dx = a.x - pfx;
            const float cxdx = a.z * dx;
            const float qx = cxdx * dx;
            const float cydx = a.w * dx;
            const unsigned long long uj = um & jbit;    // (scalar) this entry keeps the `power > 0` / min(0.99, .) guards
            // lanes of one k form a 16x4 pixel strip: skipped with one scalar bit test when
            // the staging-time pre-test (strip_pretest) found that no pixel of it can reach alpha >= 1/255, or when all its
            // pixels have finished
#define E3_FWD_STRIP(k)                                                                                                     \
            {                                                                                                               \
                if ((gm[k] & jbit) == 0ull) goto skip##k;                                                                   \
                const float dy = a.y - pfy[k];                                                                              \
                const float q = FMA(b.x * dy, dy, qx);                                                                      \
                const float power = FMA(-0.5f, q, -(cydx * dy));                                                            \
                const float G = FAST ? __builtin_amdgcn_exp2f(power * 1.4426950408889634f) : exp_det_noclamp(power);        \
                float alpha = b.y * G;                                                                                      \
                /* lane masks in SGPR pairs: valid = alive & !(alpha < 1/255) [& !(power > 0)], alpha = min(0.99, alpha):   \
                   the two bracketed guards only for the rare entries they can change anything for (strip_pretest: uj; the  \
                   block is laid out of line) -- v_cmp / v_min issue at half the rate of v_fma (tools/ubench/valu_rate) */  \
                unsigned long long valid = alive[k] & __builtin_amdgcn_fcmpf(alpha, E3_ALPHA_SKIP, 11 /* UGE */);           \
                if (!E3_FWD_GUARDS || __builtin_expect(uj != 0ull, 0)) {                                                    \
                    valid &= __builtin_amdgcn_fcmpf(power, 0.0f, 13 /* ULE */);                                             \
                    alpha = fminf(E3_ALPHA_CLAMP, alpha);                                                                   \
                }                                                                                                           \
                const float w = alpha * T[k];                                                                               \
                const float test_T = T[k] - w;                                                                              \
                const unsigned long long stop = valid & __builtin_amdgcn_fcmpf(test_T, E3_T_STOP, 4 /* OLT */);             \
                /* app = valid & ~stop: the pixels the entry is composited on; the strip-mask bit for the backward is set   \
                   only when there is one (s_andn2 leaves "result != 0" in SCC: one s_cselect, no compare) */               \
                unsigned long long app, bit;                                                                                \
                asm("s_andn2_b64 %0, %2, %3\n\ts_cselect_b64 %1, %4, 0"                                                     \
                    : "=&s"(app), "=s"(bit) : "s"(valid), "s"(stop), "s"(jbit) : "scc");                                    \
                const bool apply = __builtin_amdgcn_inverse_ballot_w64(app);                                                \
                /* one select instead of five: a zero weight leaves C and T bit-unchanged (x + 0*c == x, x - 0 == x).       \
                   (The five updates under EXEC = app instead -- s_and_saveexec, v_fmac x3, v_mov x2, s_mov exec: two SALU   \
                   for two v_cndmask -- measured +2.5 % on the kernel: profiles/EXPERIMENTS.md, round 4.) */                \
                const float we = apply ? w : 0.0f;                                                                          \
                C0[k] = FMA(b.z, we, C0[k]);                                                                                \
                C1[k] = FMA(b.w, we, C1[k]);                                                                                \
                C2[k] = FMA(c_x, we, C2[k]);                                                                                \
                last[k] = apply ? contributor : last[k];                                                                    \
                T[k] = T[k] - we;                     /* a pixel that stops here keeps its T and leaves the mask */         \
                sm[k] |= bit;                                                                                               \
                /* alive &= ~stop; when the strip's last pixel has finished no entry evaluates it again (gm = 0) */         \
                asm("s_andn2_b64 %0, %0, %2\n\ts_cselect_b64 %1, %1, 0" : "+s"(alive[k]), "+s"(gm[k]) : "s"(stop) : "scc");  \
            }                                                                                                               \
            skip##k:;
            // The same strip, instruction for instruction, as ONE asm block for the exact (polynomial exp) kernel.  The
            // forward is bound by the CU's single scalar unit (bench.py: scalar_unit_frac 0.99; +2 dummy SALU per strip:
            // +4.3 %), and hipcc cannot branch on the SCC an s_and / s_andn2 has just produced (it re-compares: s_and +
            // s_cmp + s_cbranch; `asm goto` is dropped in device code).  Hand-placed branches take SCC directly and the
            // common "no pixel of the strip stops here" case skips the stop bookkeeping: 7 scalar instructions per live
            // strip instead of 12-13, 2 per skipped strip instead of 3: -2 % on the kernel.  The strip's instructions run
            // with EXEC = alive[k] (the pixels that have finished are inactive lanes; results are the same -- their
            // compares come out 0 -- and two more SALU instructions per strip buy another -2 %, measured: the kernel runs
            // power-limited at ~1.8 GHz, and inactive lanes draw none).  Same VALU operations in the same order as the C
            // form above (which stays as the tolerance-mode kernel's body and as the readable statement of the algorithm).
            // (EXEC is all ones on entry -- the entry loop runs in wave-uniform control flow, every lane of the 128-thread
            // workgroup alive -- and the block leaves it all ones.)
#define E3_FWD_STRIP_ASM(k)                                                                                                 \
            {                                                                                                               \
                float t0_, t1_, t2_, t3_;                                                                                   \
                unsigned long long s0_, s1_, s2_;                                                                           \
                asm volatile(                                                                                               \
                    "s_and_b64 %[s0], %[gm], %[jb]\n\t"                                                                     \
                    "s_cbranch_scc0 .Lskip_%=\n\t"                                                                          \
                    "s_mov_b64 exec, %[al]\n\t"                     /* pixels that have finished: out of EXEC (below) */   \
                    "v_sub_f32 %[t0], %[ay], %[pfy]\n\t"              /* dy */                                               \
                    "v_mul_f32 %[t1], %[bx], %[t0]\n\t"                                                                     \
                    "v_fma_f32 %[t1], %[t1], %[t0], %[qx]\n\t"        /* q = fma(b.x dy, dy, qx) */                          \
                    "v_mul_f32 %[t2], %[cydx], %[t0]\n\t"                                                                   \
                    "v_fma_f32 %[t2], -0.5, %[t1], -%[t2]\n\t"        /* power = fma(-0.5, q, -(cydx dy)) */                 \
                    "v_mul_f32 %[t1], 0x3fb8aa3b, %[t2]\n\t"          /* exp_det_noclamp(power): t = power log2 e */         \
                    "v_rndne_f32 %[t0], %[t1]\n\t"                                                                          \
                    "v_sub_f32 %[t1], %[t1], %[t0]\n\t"                                                                     \
                    "v_fmamk_f32 %[t3], %[t1], 0x3aad4281, %[k1]\n\t"                                                       \
                    "v_fmaak_f32 %[t3], %[t3], %[t1], 0x3d635d55\n\t"                                                       \
                    "v_cvt_i32_f32 %[t0], %[t0]\n\t"                                                                        \
                    "v_fmaak_f32 %[t3], %[t3], %[t1], 0x3e75fcdb\n\t"                                                       \
                    "v_fmaak_f32 %[t3], %[t3], %[t1], 0x3f317213\n\t"                                                       \
                    "v_fma_f32 %[t3], %[t3], %[t1], 1.0\n\t"                                                                \
                    "v_ldexp_f32 %[t3], %[t3], %[t0]\n\t"             /* G */                                                \
                    "v_mul_f32 %[t3], %[by], %[t3]\n\t"               /* alpha = o G */                                      \
                    "s_and_b64 %[s0], %[um], %[jb]\n\t"               /* guards only for the entries that need them */      \
                    "s_cbranch_scc0 .Lsafe_%=\n\t"                                                                          \
                    "v_cmp_nlt_f32 vcc, 0, %[t2]\n\t"                 /* !(power > 0) */                                     \
                    "s_and_b64 %[s1], vcc, %[al]\n\t"                                                                       \
                    "v_max_f32 %[t3], %[t3], %[t3]\n\t"                                                                     \
                    "v_min_f32 %[t3], 0x3f7d70a4, %[t3]\n\t"          /* min(0.99, alpha) */                                 \
                    "v_cmp_ngt_f32 vcc, 0x3b808081, %[t3]\n\t"        /* !(alpha < 1/255) */                                 \
                    "s_and_b64 %[s1], %[s1], vcc\n\t"                                                                       \
                    "s_branch .Ljoin_%=\n"                                                                                  \
                    ".Lsafe_%=:\n\t"                                                                                        \
                    "v_cmp_ngt_f32 vcc, 0x3b808081, %[t3]\n\t"                                                              \
                    "s_and_b64 %[s1], vcc, %[al]\n"                   /* valid; SCC = valid != 0 */                          \
                    ".Ljoin_%=:\n\t"                                                                                        \
                    "s_cselect_b64 %[s0], %[jb], 0\n\t"               /* strip-mask bit if no pixel stops below */           \
                    "v_mul_f32 %[t0], %[T], %[t3]\n\t"                /* w = alpha T */                                      \
                    "v_sub_f32 %[t1], %[T], %[t0]\n\t"                /* test_T */                                           \
                    "v_cmp_gt_f32 vcc, 0x38d1b717, %[t1]\n\t"         /* test_T < 1e-4 */                                    \
                    "s_and_b64 %[s2], vcc, %[s1]\n\t"                 /* stop; SCC = stop != 0 */                            \
                    "s_cbranch_scc0 .Lnostop_%=\n\t"                                                                        \
                    "s_andn2_b64 %[s1], %[s1], %[s2]\n\t"             /* app = valid & ~stop */                              \
                    "s_cselect_b64 %[s0], %[jb], 0\n\t"                                                                     \
                    "s_andn2_b64 %[al], %[al], %[s2]\n\t"             /* alive &= ~stop */                                   \
                    "s_cselect_b64 %[gm], %[gm], 0\n"                 /* strip finished: no entry evaluates it again */     \
                    ".Lnostop_%=:\n\t"                                                                                      \
                    "s_or_b64 %[sm], %[sm], %[s0]\n\t"                                                                      \
                    "s_mov_b64 exec, %[s1]\n\t"                       /* the five updates on the pixels the entry is */     \
                    "v_fmac_f32 %[c0], %[bz], %[t0]\n\t"              /* composited on: plain full-rate instructions */     \
                    "v_fmac_f32 %[c1], %[bw], %[t0]\n\t"              /* under EXEC = app instead of two v_cndmask   */     \
                    "v_fmac_f32 %[c2], %[cx], %[t0]\n\t"              /* selects (-1.7 % on the kernel)               */     \
                    "v_mov_b32 %[la], %[co]\n\t"                                                                            \
                    "v_mov_b32 %[T], %[t1]\n\t"                       /* T - w (= test_T) */                                \
                    "s_mov_b64 exec, -1\n"                                                                                  \
                    ".Lskip_%=:"                                                                                            \
                    : [gm] "+s"(gm[k]), [al] "+s"(alive[k]), [sm] "+s"(sm[k]), [T] "+v"(T[k]), [c0] "+v"(C0[k]),            \
                      [c1] "+v"(C1[k]), [c2] "+v"(C2[k]), [la] "+v"(last[k]), [t0] "=&v"(t0_), [t1] "=&v"(t1_),             \
                      [t2] "=&v"(t2_), [t3] "=&v"(t3_), [s0] "=&s"(s0_), [s1] "=&s"(s1_), [s2] "=&s"(s2_)                   \
                    : [ay] "v"(a.y), [pfy] "v"(pfy[k]), [bx] "v"(b.x), [qx] "v"(qx), [cydx] "v"(cydx), [by] "v"(b.y),       \
                      [bz] "v"(b.z), [bw] "v"(b.w), [cx] "v"(c_x), [co] "v"(contributor), [k1] "v"(exp_k1),                 \
                      [jb] "s"(jbit), [um] "s"(um)                                                                          \
                    : "vcc", "scc");                                                                                        \
            }
            if (!FAST && E3_FWD_ASM) { E3_FWD_STRIP_ASM(0) E3_FWD_STRIP_ASM(1) E3_FWD_STRIP_ASM(2) E3_FWD_STRIP_ASM(3) }
            else { E3_FWD_STRIP(0) E3_FWD_STRIP(1) E3_FWD_STRIP(2) E3_FWD_STRIP(3) }
#undef E3_FWD_STRIP
#undef E3_FWD_STRIP_ASM
            jbit <<= 1;
        };
        // two entries per trip: halves the loop bookkeeping (counter, LDS address, branch) of an issue-bound loop
        int j = 0;
        for (; j + 1 < cnt; j += 2) { entry(j); entry(j + 1); }
        if (j < cnt) entry(j);
        // (cost model of tile_work below: an evaluated strip ~ 2 entries)
        live_strips += 2 * (__builtin_popcountll(sm[0]) + __builtin_popcountll(sm[1]) + __builtin_popcountll(sm[2]) +
                            __builtin_popcountll(sm[3]));
        if (lane < cnt) {
            const uint32_t mine = (__builtin_amdgcn_inverse_ballot_w64(sm[0]) ? 1u : 0u) |
                                  (__builtin_amdgcn_inverse_ballot_w64(sm[1]) ? 2u : 0u) |
                                  (__builtin_amdgcn_inverse_ballot_w64(sm[2]) ? 4u : 0u) |
                                  (__builtin_amdgcn_inverse_ballot_w64(sm[3]) ? 8u : 0u);
            strip_mask[range.x + base + lane] = (uint8_t)mine;
            if (mine) touched[perm[range.x + base + lane]] = 1;     // (slot of the entry; zero from the emission pass otherwise)
        }
        wave_sync();
    }
    {
        // cost model of the backward walk over this tile: a fixed part per visited entry plus a part per
        // evaluated strip (measured: dense tiles cost ~2.4x more per entry than sparse ones)
        uint32_t mw = max(max(last[0], last[1]), max(last[2], last[3]));
        mw = wave_max_u32(mw);
        const uint32_t cost = 2u * mw + (uint32_t)live_strips;
        if (lane == 0) {
            tile_work[tile] = cost;
            if (!order) {              // the backward's launch order: the tile joins the list of its cost class
                const uint32_t cls = e3_lpt_class(cost);
                const uint32_t pos = atomicAdd(&lpt_cnt[E3_LPT_CLASSES + cls], 1u);
                lpt_list[((size_t)E3_LPT_CLASSES + cls) * lpt_cap + pos] = (uint32_t)tile;
            }
        }
    }
    if (trace && lane == 0) {
        trace[E3_TRACE_WORDS * (size_t)tile + 0] = t_start;
        trace[E3_TRACE_WORDS * (size_t)tile + 1] = wall_clock64();
        trace[E3_TRACE_WORDS * (size_t)tile + 2] = ((unsigned long long)n << 32) | (unsigned)processed;
        trace[E3_TRACE_WORDS * (size_t)tile + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) |
                                      (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        trace[E3_TRACE_WORDS * (size_t)tile + 4] = c_start;                    // effective clock of the kernel = cycles / wall time
        trace[E3_TRACE_WORDS * (size_t)tile + 5] = __builtin_readcyclecounter();
    }
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;
    out += (size_t)view * 3 * HW;               // (nviews, 3, H, W)
    final_T += (size_t)view * HW;
    n_contrib += (size_t)view * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (inside[k]) {
            size_t pix = (size_t)(py0 + 4 * k) * W + px;
            final_T[pix] = T[k];
            n_contrib[pix] = last[k];
            out[pix] = FMA(T[k], bg0, C0[k]);
            out[HW + pix] = FMA(T[k], bg1, C1[k]);
            out[2 * HW + pix] = FMA(T[k], bg2, C2[k]);
        }
    }
}

#define E3_RENDER_FWD_PARAMS                                                                                             \
    unsigned long long *__restrict__ trace, int ntiles, int tiles_per_view, const uint32_t *__restrict__ order, int gx,  \
        int W, int H, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ perm,                               \
        const uint32_t *__restrict__ emit_gid, const float4 *__restrict__ rec, const float *__restrict__ bg,            \
        float *__restrict__ out, float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,                          \
        uint32_t *__restrict__ tile_work, uint8_t *__restrict__ strip_mask, uint8_t *__restrict__ touched,              \
        uint32_t *__restrict__ lpt_cnt, uint32_t *__restrict__ lpt_list, uint32_t lpt_cap
#define E3_RENDER_FWD_ARGS                                                                                               \
    trace, ntiles, tiles_per_view, order, gx, W, H, ranges, perm, emit_gid, rec, bg, out, final_T, n_contrib, tile_work, \
        strip_mask, touched, lpt_cnt, lpt_list, lpt_cap
// (plain kernels around the one body: the profiles, the bench line and the reviews name `render_fwd_kernel`)
__global__ __launch_bounds__(RENDER_WAVES * WAVE, E3_FWD_WAVES) void render_fwd_kernel(E3_RENDER_FWD_PARAMS) {
    render_fwd_body<false>(E3_RENDER_FWD_ARGS);
}
__global__ __launch_bounds__(RENDER_WAVES * WAVE, E3_FWD_WAVES) void render_fwd_fast_kernel(E3_RENDER_FWD_PARAMS) {
    render_fwd_body<true>(E3_RENDER_FWD_ARGS);
}

// background fill for P == 0 or empty scenes is handled by the same kernel (ranges are zero).

// ------------------------------------------------------------------------------------ mark_visible
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means,
                                                           const float* __restrict__ view,
                                                           uint8_t* __restrict__ present) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float vz = XFORM(view, 2, mx, my, mz);
    present[i] = vz > E3_NEAR_CULL_Z ? 1 : 0;
}

// ------------------------------------------------------------------------------------ host driver
extern thread_local char g_err[512];
int e3_fail(hipError_t e, const char* what);
#define HIP_OK(expr)                                          \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return e3_fail(_e, #expr);      \
    } while (0)
#define KERNEL_OK(name)                                       \
    do {                                                      \
        hipError_t _e = hipGetLastError();                    \
        if (_e != hipSuccess) return e3_fail(_e, name);       \
        if (debug) {                                          \
            _e = hipStreamSynchronize(s);                     \
            if (_e != hipSuccess) return e3_fail(_e, name);   \
        }                                                     \
    } while (0)

static int ceil_log2(uint32_t v) {
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

// Forward is split in two enqueue-only halves around the instance count: `begin` ends with an async copy
// of the count to host memory, the caller synchronises and `finish` sizes the binning buffers with it
// (the one-call e3dgs_rasterize_forward = begin + stream sync + finish).
int e3_forward_begin_impl(e3_alloc_fn geom_alloc, void* geom_user, e3_alloc_fn img_alloc, void* img_user,
                          const ViewBatch& views, int P, int D, int M, int W, int H, const float* means3D,
                          const float* shs, const float* colors, const float* opac, const float* scales,
                          float scale_modifier, const float* rots, const float* cov_pre, int* radii, int debug,
                          int flags, int* count_host, hipStream_t s) {
    const ViewSet vs = make_view_set(views, W, H, scale_modifier);
    const int nv = vs.n;
    const int ntiles = vs.v[0].gx * vs.v[0].gy;
    const size_t npix = (size_t)W * H;
    const size_t Q = (size_t)P * nv;             // splats

    char* gp = geom_alloc(geom_user, GeomState::required(Q));
    char* ip = img_alloc(img_user, ImageState::required(npix * nv, (size_t)ntiles * nv));
    if (!gp || !ip) return e3_fail(hipErrorOutOfMemory, "scratch allocation callback returned NULL");
    e3_geom_opts_remember(gp, Q, e3_call_opts(flags));      // (before from(): it advances the pointer)
    GeomState geom = GeomState::from(gp, Q);
    ImageState img = ImageState::from(ip, npix * nv, (size_t)ntiles * nv);
    if (P <= 0) HIP_OK(hipMemsetAsync(img.ranges, 0, (size_t)ntiles * nv * sizeof(uint2), s));
    if (!(flags & E3_FLAG_COUNT_MAPPED) || P <= 0) *count_host = 0;      // (mapped: the caller armed a sentinel)
    if (P > 0) {
        const unsigned pb = (unsigned)(((size_t)P + 255) / 256);      // one thread per Gaussian (its views are a loop)
        // descriptors of the scan behind the count pass: the tail of the scratch, which the depth sort does not use
        uint32_t* bin_scan_desc = geom.scratch + depth_sort_scratch_words(Q);
        {
        ProfScope ps(PS_PREPROCESS, s);
        // (P, M, 3) coefficient rows that start on 16-byte boundaries are loaded as float4 (the planar layout is coalesced
        // across lanes as it is)
        const bool vec4 = shs && !(flags & (E3_FLAG_SH_PLANAR | E3_FLAG_DEFER_COLOR)) &&
                          ((reinterpret_cast<uintptr_t>(shs) | (uintptr_t)((size_t)M * 12)) & 15) == 0;
        auto kern = vec4 ? (D <= 0 ? preprocess_kernel<0, true> : D == 1 ? preprocess_kernel<1, true>
                            : D == 2 ? preprocess_kernel<2, true> : D == 3 ? preprocess_kernel<3, true>
                                                                            : preprocess_kernel<4, true>)
                         : (D <= 0 ? preprocess_kernel<0, false> : D == 1 ? preprocess_kernel<1, false>
                            : D == 2 ? preprocess_kernel<2, false> : D == 3 ? preprocess_kernel<3, false>
                                                                            : preprocess_kernel<4, false>);
        kern<<<dim3(pb), dim3(256), 0, s>>>(P, M, means3D, shs, colors, opac, scales, rots, cov_pre, vs, flags, radii,
                                            geom.rec, geom.clamped, geom.rect, geom.key0, img.ranges, ntiles * nv,
                                            geom.offsets, bin_scan_desc, (int)scan_desc_words(Q), img.lpt_cnt);
        }
        KERNEL_OK("preprocess_kernel");
        uint32_t *keys_sorted, *order;
        {
        ProfScope ps(PS_SORT_DEPTH, s);
        // culled splats carry the all-ones key: the first pass drops them, geom.nvis receives the kept count
        // E3DGS_DEPTH_SORT_WIDE=1: three passes of 11 + 11 + 10 bits (launch_depth_sort_wide) instead of four of 8.  Built and
        // measured in round 6, NOT the default: 0.188 ms against 0.115 ms for the benchmark's 3 M splats -- the 2048-bin
        // passes scatter two keys per (workgroup, digit) instead of sixteen (profiles/EXPERIMENTS.md, round 6)
        static const bool wide = getenv("E3DGS_DEPTH_SORT_WIDE") && getenv("E3DGS_DEPTH_SORT_WIDE")[0] == '1';
        if (wide) {
            // (the sorted keys themselves are not produced: nobody reads them)
            const int rc = launch_depth_sort_wide(geom.key0, geom.key1, geom.ord0, geom.ord1, Q, geom.scratch, s, geom.nvis);
            if (rc) return rc;
            order = geom.ord0; keys_sorted = geom.key0;
        } else {
        const int rc = launch_radix_sort_pairs(geom.key0, geom.key1, geom.ord0, geom.ord1, Q, 32, geom.scratch,
                                               &keys_sorted, &order, s, true, geom.nvis);
        if (rc) return rc;
        }
        }
        KERNEL_OK("radix sort (depth)");
        if (order != geom.ord0 || keys_sorted != geom.key0)
            return e3_fail(hipErrorUnknown, "internal: depth order not in ord0 / key0");
        const CallOpts opt = e3_call_opts(flags);
        const int gshift = e3_bin_group_shift(Q, opt.small_paths);
        const unsigned nwaves = (unsigned)((Q + ((size_t)1 << gshift) - 1) >> gshift);
        const unsigned bb = (nwaves + BIN_WAVES - 1) / BIN_WAVES;
        {
        ProfScope ps(PS_SCAN_EMIT, s);
        bin_kernel<false><<<dim3(bb), dim3(BIN_WAVES * WAVE), 0, s>>>((int)Q, gshift, nv, ntiles, order, geom.nvis,
                                                                     geom.rect, vs.v[0].gx <= 255 && vs.v[0].gy <= 255,
                                                                     geom.rec, vs.v[0].gx, opt.cull, nullptr, geom.tiles,
                                                                     nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0u,
                                                                     bin_handoff() ? geom.binrec : nullptr);
        // inclusive scan of the per-wave counts; offsets[w] = end of wave w, so start = offsets[w-1]
        const int rc = launch_scan_chained_u32(geom.tiles, geom.offsets + 1, (size_t)nwaves, bin_scan_desc, true, s,
                                               (flags & E3_FLAG_COUNT_MAPPED) ? count_host : nullptr);
        if (rc) return rc;
        }
        KERNEL_OK("bin count + scan");
        // the instance count sizes the binning buffers: the op's single device->host read-back
        if (!(flags & E3_FLAG_COUNT_MAPPED))      // (mapped: the scan's last thread stored the total into count_host)
            HIP_OK(hipMemcpyAsync(count_host, geom.offsets + nwaves, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    return 0;
}

// count_on_device: `num_rendered` is the CAPACITY the caller sized the binning buffers for before the instance count was
// known (a training loop: the previous iteration's count plus a margin); the kernels read the count itself from device
// memory, so no host wait sits between the two halves of the forward.  If the count turns out larger than the capacity
// nothing is emitted or sorted (every tile list stays empty) and the caller, who gets the count through its mapped host
// word as usual, repeats the call with larger buffers.
int e3_forward_finish_impl(e3_alloc_fn bin_alloc, void* bin_user, int nviews, int P, int W, int H,
                           const float* background, char* geom_buffer, char* image_buffer, int num_rendered,
                           float* out_color, int debug, int flags, hipStream_t s, const DeferredColour* dc,
                           int count_on_device) {
    const CallOpts opt = e3_call_opts(flags);      // (the caller passes the option bits it gave to `begin`)
    const int gx = (W + E3_TILE - 1) / E3_TILE, gy = (H + E3_TILE - 1) / E3_TILE;
    const int tiles_per_view = gx * gy;
    const int ntiles = tiles_per_view * nviews;
    const size_t Q = (size_t)P * nviews;
    const uint32_t I = (uint32_t)num_rendered;
    char* gp = geom_buffer;
    char* ip = image_buffer;
    if (const int rc = e3_geom_opts_check(gp, Q, opt, "forward finish")) return rc;
    GeomState geom = GeomState::from(gp, Q);
    ImageState img = ImageState::from(ip, (size_t)W * H * nviews, ntiles);
    char* bp = bin_alloc(bin_user, BinningState::required(I));
    if (!bp) return e3_fail(hipErrorOutOfMemory, "binning allocation callback returned NULL");
    BinningState bin = BinningState::from(bp, I);
    const bool use_lpt = e3_use_lpt_lists(ntiles, num_rendered, P);
    if (I > 0 && P > 0) {
        // at least one pass even for a single tile: the first pass is what materialises the identity payload
        const int tile_bits = ntiles > 1 ? ceil_log2((uint32_t)ntiles) : 1;
        const int passes = radix_passes(tile_bits);
        const int keys16 = ntiles <= 65536 ? 1 : 0;           // 16-bit sort keys: a third of the sort's bytes less per pass
        // choose the emit target so that the final sorted values (slot indices) land in bin.perm
        uint32_t *k0 = bin.keys, *k1 = bin.keys_alt, *v0 = bin.perm, *v1 = bin.vals_alt;
        if (passes & 1) { uint32_t* t = v0; v0 = v1; v1 = t; }
        const int gshift = e3_bin_group_shift(Q, opt.small_paths);
        const unsigned nwaves = (unsigned)((Q + ((size_t)1 << gshift) - 1) >> gshift);
        const unsigned bb = (nwaves + BIN_WAVES - 1) / BIN_WAVES;
        // the instance count on the device: the last element of the inclusive scan of the per-wave counts
        const uint32_t* count_dev = count_on_device ? geom.offsets + nwaves : nullptr;
        {
        ProfScope ps(PS_SCAN_EMIT, s);
        bin_kernel<true><<<dim3(bb), dim3(BIN_WAVES * WAVE), 0, s>>>((int)Q, gshift, nviews, tiles_per_view, geom.ord0,
                                                                    geom.nvis, geom.rect, gx <= 255 && gy <= 255, geom.rec, gx,
                                                                    opt.cull, geom.offsets, nullptr, k0, keys16,
                                                                    bin.emit_gid, geom.run, bin.touched, count_dev, I,
                                                                    bin_handoff() ? geom.binrec : nullptr);
        }
        KERNEL_OK("bin emit");
        uint32_t* vs;
        {
        // stable sort by tile id; the [start, end) of every tile comes out of the same launches (img.ranges: zeroed by
        // the preprocess kernel, tiles without an instance stay / are rewritten as (0, 0)), and so does the forward
        // launch order when it is kept as per-class lists
        ProfScope ps(PS_SORT_TILE, s);
        int rc;
        uint32_t* lc = use_lpt ? img.lpt_cnt : nullptr;
        uint32_t* ll = use_lpt ? img.lpt_list : nullptr;
        if (keys16) {
            uint16_t* ks16;
            rc = launch_radix_sort_pairs_u16(reinterpret_cast<uint16_t*>(k0), reinterpret_cast<uint16_t*>(k1), v0, v1,
                                             (size_t)I, tile_bits, bin.scratch, &ks16, &vs, s, true, count_dev, img.ranges,
                                             (uint32_t)ntiles, lc, ll);
        } else {
            uint32_t* ks;
            rc = launch_radix_sort_pairs(k0, k1, v0, v1, (size_t)I, tile_bits, bin.scratch, &ks, &vs, s, true, nullptr,
                                         count_dev, img.ranges, (uint32_t)ntiles, lc, ll);
        }
        if (rc) return rc;
        }
        KERNEL_OK("radix sort (tile)");
        if (vs != bin.perm) return e3_fail(hipErrorUnknown, "internal: sorted list not in perm");
    }
    int nslots = ntiles;
    if (!use_lpt) {
    ProfScope ps(PS_RANGES, s);
    nslots = launch_tile_order(ntiles, tiles_per_view, gx, img.ranges, nullptr, img.order, s);
    }
    KERNEL_OK("tile_order_kernel");
    if (dc && P > 0) {
        // deferred SH -> RGB (E3_FLAG_DEFER_COLOR).  `before` lets the caller make this stream wait for whatever
        // still owns the SH coefficients (e.g. the previous iteration's optimizer step on another stream).
        if (dc->before) dc->before(dc->user);
        ProfScope ps(PS_PREPROCESS, s);
        const ViewSet vs = make_view_set(dc->views, W, H, 1.0f);
        colour_kernel<<<dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s>>>(P, nviews, dc->D, dc->M, dc->means3D, dc->shs,
                                                                             vs, dc->flags, geom.rect, geom.rec,
                                                                             geom.clamped);
        KERNEL_OK("colour_kernel");
    }
    ProfScope ps_render(PS_RENDER_FWD, s);
    (opt.fast_exp ? render_fwd_fast_kernel : render_fwd_kernel)<<<dim3((nslots + RENDER_WAVES - 1) / RENDER_WAVES), dim3(RENDER_WAVES * WAVE), 0, s>>>(
        g_trace, nslots, tiles_per_view, use_lpt ? nullptr : img.order, gx, W, H, img.ranges, bin.perm, bin.emit_gid, geom.rec,
        background, out_color, img.final_T, img.n_contrib, img.work, bin.strip_mask, bin.touched, img.lpt_cnt, img.lpt_list,
        (uint32_t)ntiles);
    KERNEL_OK("render_fwd_kernel");
    return 0;
}

int e3_mark_visible_impl(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    if (P <= 0) return 0;
    mark_visible_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, means3D, view, present);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "mark_visible_kernel");
}
