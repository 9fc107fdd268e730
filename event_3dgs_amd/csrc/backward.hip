// backward.hip -- gradient kernels (SURVEY 8a rows a19-a21, Appendix B [UPSTREAM]).
//
//   render_bwd_kernel  a19  one wave per tile, 4 pixels per lane, back-to-front walk over the same
//                           sorted list; per list entry the 9 partial sums (mean2D 2, conic 3,
//                           opacity 1, colour 3) are reduced across the wave through LDS and stored as
//                           ONE 48-byte record at the instance's slot -- plain stores, no atomics
//                           (the reference adds every value of every pixel with atomicAdd).
//   run_reduce_kernel  ---  per-splat sum of its (contiguous) instance records, fixed order.
//   geom_bwd_kernel    a20+a21 fused: conic -> Sigma2 -> Sigma3 and view-space mean, NDC mean through
//                           the projection, SH backward, Sigma3 -> scale / quaternion
//                           (geom_bwd_multi_kernel: all views of an iteration in one pass).
#include "common.h"
#include <stdlib.h>
#include <string.h>

#ifndef E3_BWD_WG_WAVES
#define E3_BWD_WG_WAVES 4
#endif
constexpr int BWD_WAVES = E3_BWD_WG_WAVES;

extern unsigned long long* g_trace;
int launch_tile_order(int ntiles, int tiles_per_view, int gx, uint2* ranges, const uint32_t* work, uint32_t* order,
                      hipStream_t s);
#ifndef E3_BWD_WAVES
#define E3_BWD_WAVES 6
#endif
#ifndef E3_R1_TWO_READS
#define E3_R1_TWO_READS 1     // rank-1 body: two 16-byte LDS broadcasts per entry instead of three (pmin staged beside colour . w)
#endif
#ifndef E3_BWD_STATS_WAVES
#define E3_BWD_STATS_WAVES 5      // render_bwd_stats_kernel (second gradient chain: +16 VGPRs)
#endif
// STATS (shared-pose iterations that also collect densification statistics, EventTrainer.compute_gradients): view 0 is
// the render that train.py:144 and :159 both produce; its backward carries the SUM of the two loss terms' pixel
// gradients, while the statistics (train.py:145,317-320) want the screen-space mean gradient under the intensity term's
// pixel gradient `dL_dpix2` ALONE.  The backward is linear in the pixel gradient, so the tiles of view 0 run a second
// dL/dalpha chain on dL_dpix2 next to the first -- sharing G, alpha, T and the keep decisions -- and store its two
// screen-space sums per instance in `part2` (2 floats at the instance's slot).  The default instantiation is unchanged.
// FAST: the forward ran with E3DGS_FLAG_FAST_EXP -- its alpha is min(0.99, o v_exp_f32(power log2 e)), which this kernel
// computes with the same instruction on the same bits: the forward's keep decisions are reproduced by one compare, without
// the band logic the polynomial exp of the exact mode needs.
// RANK1 (e3dgs_rasterize_backward_multi_rank1): the pixel gradient of this tile's view is s(pixel) * w with ONE weight
// vector w per view -- what every loss on a luminance gives: the two contrast renders of an event iteration
// (dL/dC = s (0.4124, 0.35758, 0.1804), utils/loss_utils.py:24-28,234-249) and every --gray loss (s (0.299, 0.587, 0.114),
// :18-23,40-48).  Then c . dL/dC = (c . w) s, where c . w is per ENTRY (computed once at staging, lane-parallel), and the
// three colour sums are w times ONE sum of alpha T s: one multiplication instead of three FMAs for the colour chain and
// one FMA instead of three for the colour sums per evaluated (pixel, entry), seven values instead of nine through the
// per-entry LDS transposition (no ninth-value chain: -2 ds_write, -1 ds_read, -3 DPP adds, -1 parked store per entry),
// a third of the pixel-gradient loads.  Same record layout out (the commit lanes multiply the one sum by w).
struct StagedRec { float4 a, b, c; };
template <bool STATS>
struct BwdShared {
    // the 64 staged records of a round, 48 B each: one LDS address per entry, the three 16-B broadcasts are immediate
    // offsets of it (three separate arrays cost two more address adds per entry); 48-B stride keeps the staging stores
    // conflict-free (8 lanes x 16 B per LDS cycle land on 32 distinct banks)
    StagedRec sRec[BWD_WAVES][WAVE];
    // Gradient reduction through LDS.  DPP adds cost ~9 cycles per wave-instruction on this chip (plain adds 2.7),
    // so instead of a 54-op (or 26-op transposed) DPP butterfly the nine per-lane sums of an entry are transposed
    // through the wave's LDS slice: 9 conflict-free ds_write_b32, then lane (v,p) = (lane>>3, lane&7) reads the 8
    // floats [8p,8p+8) of value v as two ds_read_b128 and adds them, and three DPP steps finish the 8-lane groups.
    // The LDS pipe is otherwise idle in this kernel.  Reduced sums are parked in sPart and committed every 16
    // entries by lanes 0..15 as three 16-byte stores into the instance's own record (no atomics: the per-Gaussian
    // sum over its instances happens in geom_bwd_kernel, in a fixed order -> deterministic gradients).
    __attribute__((aligned(16))) float sRed[BWD_WAVES][STATS ? 11 : 9][68];   // per-entry transpose buffer: [value][lane], rows padded to 68
    __attribute__((aligned(16))) float sPart[BWD_WAVES][16][16];     // [entry & 15][16 floats]: reduced sums parked until the commit
    float sPart2[STATS ? BWD_WAVES : 1][16][2];                      // STATS: the two extra sums of an entry
};
// SLDS: the STATS flavour of the kernel's LDS block (a stats kernel runs the plain body in the tiles of its other views)
template <bool SLDS, bool STATS, bool FAST, bool RANK1>
__device__ __forceinline__ void render_bwd_body(
    BwdShared<SLDS>& sh, const int tile,
    unsigned long long* __restrict__ trace, int tiles_per_view, int gx,
    int W, int H, const uint2* __restrict__ ranges, const uint32_t* __restrict__ emit_gid,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ perm, const uint8_t* __restrict__ strip_mask, const float* __restrict__ dL_dpix,
    float* __restrict__ part /* (I,9) per-instance records at their SLOTS, packed: mx my A B C o c0 c1 c2 */,
    const float* __restrict__ dL_dpix2 /* STATS: (3,H,W) second pixel gradient of view 0 */,
    float* __restrict__ part2 /* STATS: (I,2) NDC-unit screen-space mean gradient under dL_dpix2, view-0 slots */,
    const float rw0, const float rw1, const float rw2 /* RANK1: the view's weight vector (wave-uniform) */) {
    auto& sRec = sh.sRec;
    auto& sRed = sh.sRed;
    auto& sPart = sh.sPart;
    auto& sPart2 = sh.sPart2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f4_t LdsF4;
    struct F3 { float x, y, z; };               // 12-byte store unit of the packed 36-byte records (4-byte aligned)
    // LDS byte address of this wave's staged records, as a scalar (the per-entry address is then scalar arithmetic)
    const uint32_t recs_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)&sRec[wave][0].a.x);
    uint32_t recs_base_v = recs_base;
    asm volatile("" : "+v"(recs_base_v));       // the same address kept in a VGPR for the per-entry v_mad
    const unsigned long long t_start = trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = trace ? __builtin_readcyclecounter() : 0ull;     // s_memtime: shader cycles
    const int view = tile / tiles_per_view, ltile = tile - view * tiles_per_view;
    const int tx = ltile % gx, ty = ltile / gx;
    const int px = tx * E3_TILE + (lane & 15);
    const int py0 = ty * E3_TILE + (lane >> 4);
    const float pfx = (float)px;
    const size_t HW = (size_t)H * W;
    final_T += (size_t)view * HW;               // per-view planes of the batch
    n_contrib += (size_t)view * HW;
    dL_dpix += (size_t)view * 3 * HW;           // (nviews, 3, H, W)
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // Per-pixel state of the back-to-front walk.  With C = sum_j c_j a_j T_j + T_final bg and
    // T_j = prod_{i<j}(1 - a_i):   dC/da_g = c_g T_g - (sum_{j>g} c_j a_j T_j + T_final bg) / (1 - a_g).
    // Contracting with dL/dC first leaves ONE scalar running sum per pixel,
    //   Q_g = T_final (bg . dL/dC) + sum_{j>g} a_j T_j (c_j . dL/dC),
    // instead of the reference's accumulated colour / last colour / last alpha (7 registers -> 1).
    float pfy[4], T[4], Q[4], dp0[4], dp1[4], dp2[4];
    float dq0[4], dq1[4], dq2[4], Q2[4];                       // STATS: the second chain's pixel gradient and running sum
    const bool sv = STATS && view == 0;                        // (wave-uniform)
    uint32_t last[4];
    uint32_t maxc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int py = py0 + 4 * k;
        pfy[k] = (float)py;
        bool inside = (px < W) && (py < H);
        size_t pix = (size_t)py * W + px;
        T[k] = inside ? final_T[pix] : 0.0f;
        last[k] = inside ? n_contrib[pix] : 0u;
        dp0[k] = inside ? dL_dpix[pix] : 0.0f;              // RANK1: the scalar field s
        dp1[k] = (!RANK1 && inside) ? dL_dpix[HW + pix] : 0.0f;
        dp2[k] = (!RANK1 && inside) ? dL_dpix[2 * HW + pix] : 0.0f;
        Q[k] = RANK1 ? T[k] * (FMA(bg0, rw0, FMA(bg1, rw1, bg2 * rw2)) * dp0[k])
                     : T[k] * FMA(bg0, dp0[k], FMA(bg1, dp1[k], bg2 * dp2[k]));
        if (STATS) {
            dq0[k] = (sv && inside) ? dL_dpix2[pix] : 0.0f;
            dq1[k] = (sv && inside) ? dL_dpix2[HW + pix] : 0.0f;
            dq2[k] = (sv && inside) ? dL_dpix2[2 * HW + pix] : 0.0f;
            Q2[k] = T[k] * FMA(bg0, dq0[k], FMA(bg1, dq1[k], bg2 * dq2[k]));
        }
        maxc = last[k] > maxc ? last[k] : maxc;
    }
    maxc = wave_max_u32(maxc);
    uint2 range = ranges[tile];
    range.x = __builtin_amdgcn_readfirstlane(range.x); range.y = __builtin_amdgcn_readfirstlane(range.y);
    const int n = __builtin_amdgcn_readfirstlane((int)maxc);   // entries [0, n) of the tile list can contribute
    const unsigned long long t_loop = trace ? wall_clock64() : 0ull;

    // walk entries n-1 ... 0; round r covers list positions n-1-r*64-lane
    // staging pipeline as in the forward kernel (three deep: emission index -> Gaussian id -> record); the strip-mask
    // byte of an entry is indexed by its LIST position, so it rides one stage ahead without a gather.
    // The forward recorded, per entry, which of the tile's four 16x4 strips it evaluated at all (render_fwd_kernel:
    // strip_mask).  An entry it evaluated nowhere composited nothing -- more than half of the walked entries, whose
    // four liveness tests (5 VALU + 3 SALU each) and accumulator resets were most of this kernel's instructions:
    // they are skipped with scalar bit tests, and the walk visits only the entries with a non-zero mask (s_ff1).
    float4 ra = make_float4(0, 0, 0, 0), rb = make_float4(0, 0, 0, 0), rc = make_float4(0, 0, 0, 0);
    uint32_t re = 0, e_next = 0, id_next = 0, e_next2 = 0, rm = 0, m_next = 0;
    const uint8_t* __restrict__ tmask = strip_mask + range.x;
    if (lane < n) {
        re = perm[range.x + (uint32_t)(n - 1 - lane)];
        rm = tmask[n - 1 - lane];
        const uint32_t rid = emit_gid[re];
        ra = rec[3 * (size_t)rid]; rb = rec[3 * (size_t)rid + 1]; rc = rec[3 * (size_t)rid + 2];
    }
    if (WAVE + lane < n) {
        e_next = perm[range.x + (uint32_t)(n - 1 - (WAVE + lane))];
        m_next = tmask[n - 1 - (WAVE + lane)];
        id_next = emit_gid[e_next];
    }
    if (2 * WAVE + lane < n) e_next2 = perm[range.x + (uint32_t)(n - 1 - (2 * WAVE + lane))];
    for (int base = 0; base < n; base += WAVE) {
        const int cnt = min(WAVE, n - base);
        rc.w = __uint_as_float(re);                 // the record's spare word carries the emission index (where its gradient record goes)
        if (RANK1) {
            rb.z = FMA(rb.z, rw0, FMA(rb.w, rw1, rc.x * rw2));      // colour . w of the entry, in the red slot
            if (E3_R1_TWO_READS) rb.w = rc.y;                        // ... and pmin beside it: the entry loop reads a, b only
        }
        sRec[wave][lane].a = ra; sRec[wave][lane].b = rb; sRec[wave][lane].c = rc;
        const uint32_t mvec = lane < cnt ? rm : 0u;
        wave_sync();
        if (base + WAVE + lane < n) {
            re = e_next; rm = m_next;
            ra = rec[3 * (size_t)id_next]; rb = rec[3 * (size_t)id_next + 1]; rc = rec[3 * (size_t)id_next + 2];
        }
        if (base + 2 * WAVE + lane < n) {
            e_next = e_next2; id_next = emit_gid[e_next2];
            m_next = tmask[n - 1 - (base + 2 * WAVE + lane)];
        }
        if (base + 3 * WAVE + lane < n) e_next2 = perm[range.x + (uint32_t)(n - 1 - (base + 3 * WAVE + lane))];
        const unsigned long long nz = __builtin_amdgcn_ballot_w64(mvec != 0u);
        int nrem = n - base;
        asm volatile("" : "+s"(nrem));           // (opaque: otherwise re-associated into two scalar ops per entry)
        for (int jb = 0; jb < cnt; jb += 16) {          // 16 entries share one commit
            uint32_t ng = (uint32_t)(nz >> jb) & 0xFFFFu;
            while (ng != 0u) {
                const int j = jb + __builtin_ctz(ng);
                ng &= ng - 1u;
                const uint32_t em = (uint32_t)__builtin_amdgcn_readlane((int)mvec, j);
                // (32-bit LDS pointer arithmetic: through the generic pointer the compiler forms the address with a 64-bit mad)
                // one VALU op (24-bit multiply-add with the base in a vector register) instead of s_mul + s_add + v_mov
                uint32_t rec_addr;
                asm("v_mad_u32_u24 %0, %1, 48, %2" : "=v"(rec_addr) : "s"(j), "v"(recs_base_v));
                LdsF4* vp = (LdsF4*)(uintptr_t)rec_addr;
                const f4_t va = vp[0], vb = vp[1];
                const float4 a = make_float4(va.x, va.y, va.z, va.w);
                const float4 b = make_float4(vb.x, vb.y, vb.z, vb.w);
                float4 c;
                if (RANK1 && E3_R1_TWO_READS) {
                    // the rank-1 body needs neither green nor blue: pmin rides in b.w (staging), and the band bound is the
                    // sum preprocess_kernel stored -- pmin + 4e-4f, the same single fp32 addition: the same bits
                    c = make_float4(0.0f, vb.w, vb.w + 4e-4f, 0.0f);
                } else {
                    const f4_t vc = vp[2];
                    c = make_float4(vc.x, vc.y, vc.z, vc.w);
                }
                const uint32_t contributor = (uint32_t)(nrem - j);         // 1-based position in the list
                const float dx = a.x - pfx;
                // power in the FORWARD's operation order (q = fma(C dy, dy, (A dx) dx); power = fma(-0.5, q, -((B dx) dy))):
                // the same bits as the forward / the oracle.  A Horner form in dy is 2 VALU cheaper per strip, but for a
                // large anisotropic splat the three terms cancel (|terms| ~ 1e3 against a sum of ~5 along the major axis), so
                // a different rounding order moves G by 1e-4 at exactly the far pixels that dominate the conic sums (weights
                // dx^2, dy^2): measured 5e-5 on the sums and 1.4e-3 on one scale gradient per million Gaussians.  With
                // identical bits the backward's G is the forward's up to the 1-ulp v_exp_f32.
                const float qx = (a.z * dx) * dx;
                const float cydx = a.w * dx;
                // per-lane partial sums over its (up to) 4 pixels, with u = G dL/dalpha (so that dL/dG * G = o u):
                //   So = sum u, Uy = sum u dy, Uyy = sum u dy^2, Sc* = sum alpha T dL/dC*.
                // A lane's four pixels share dx, and the opacity o is the entry's: the x moments are dx So, dx^2 So and
                // dx Uy (three multiplications per entry instead of five more operations per pixel), and o is applied
                // to the reduced sums by the commit lanes.
                float Uy = 0.0f, Uyy = 0.0f, So = 0.0f, Sc0 = 0.0f, Sc1 = 0.0f, Sc2 = 0.0f;
                float So2 = 0.0f, Uy2 = 0.0f;               // STATS: sum u', sum u' dy of the second chain
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (((em >> k) & 1u) == 0u) continue;   // scalar: the forward did not evaluate this strip
                    const float dy = a.y - pfy[k];
                    const float power = FMA(-0.5f, FMA(b.x * dy, dy, qx), -(cydx * dy));
                    // All decisions are lane masks in SGPR pairs (compare intrinsics) combined with scalar ops; the one
                    // divergent branch takes its mask through inverse_ballot (an s_and_saveexec, no VALU).
                    // c.y = pmin: below it alpha < 1/255 whatever the rounding (preprocess_kernel)
                    const unsigned long long live = __builtin_amdgcn_uicmp(contributor, last[k], 37 /* ULE */) &
                                                    __builtin_amdgcn_fcmpf(power, c.y, 3 /* OGE */);
                    // (no `live != 0` early-out: the forward sets a strip's mask bit only where the entry was composited on
                    // some pixel of the strip, and that pixel is live here -- its n_contrib is >= this position, `power` has
                    // the forward's bits -- so the test would cost two scalar instructions per strip and never fire)
                    {
                        // The two skip decisions of the forward: power > 0 -- exact here, `power` has the forward's bits --
                        // and alpha < 1/255, which the forward took with its polynomial exp.  Away from that threshold the
                        // outcome cannot depend on the exp: power >= c.z = pmin + 4e-4 keeps.  Inside the band the forward's
                        // own exp is redone, so that backward differentiates exactly the set of (pixel, Gaussian) pairs the
                        // forward composited.
                        // hardware exp2 (1 ulp) instead of the forward's bit-reproducible polynomial: the VALUES of
                        // backward are tolerance-checked, and 2 issue slots replace 10 on the most executed path.
                        // Issued before the mask algebra so that its latency overlaps the compares.
                        float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
                        asm volatile("" : "+v"(G));          // keep it here: the compiler would sink it into the branch
                        const unsigned long long nonpos = __builtin_amdgcn_fcmpf(power, 0.0f, 13 /* ULE: !(power > 0) */);
                        unsigned long long keep;
                        if (FAST) {
                            const float a_f = fminf(E3_ALPHA_CLAMP, b.y * G);
                            keep = live & nonpos & __builtin_amdgcn_fcmpf(a_f, E3_ALPHA_SKIP, 11 /* UGE: !(alpha < 1/255) */);
                        } else {
                            const unsigned long long hi = __builtin_amdgcn_fcmpf(power, c.z, 3 /* OGE */);
                            keep = live & hi & nonpos;
                            const unsigned long long near = live & ~hi & nonpos;
                            if (near != 0ull) {                  // rare (<1 % of the live strips)
                                const float af = fminf(E3_ALPHA_CLAMP, b.y * exp_det_noclamp(power));
                                keep |= near & __builtin_amdgcn_fcmpf(af, E3_ALPHA_SKIP, 11 /* UGE: !(af < 1/255) */);
                            }
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(keep)) {
                            const float alpha = fminf(E3_ALPHA_CLAMP, b.y * G);
                            // 1-ulp v_rcp_f32
                            const float inv_one_m = __builtin_amdgcn_rcpf(1.0f - alpha);
                            T[k] = T[k] * inv_one_m;                        // transmittance in front of this entry
                            const float cd = RANK1 ? b.z * dp0[k] : FMA(b.z, dp0[k], FMA(b.w, dp1[k], c.x * dp2[k]));
                            const float w = alpha * T[k];
                            const float dL_dalpha = FMA(T[k], cd, -(Q[k] * inv_one_m));
                            Q[k] = FMA(w, cd, Q[k]);
                            Sc0 = FMA(w, dp0[k], Sc0);                      // RANK1: sum alpha T s (x w at the commit)
                            if (!RANK1) {
                                Sc1 = FMA(w, dp1[k], Sc1);
                                Sc2 = FMA(w, dp2[k], Sc2);
                            }
                            const float u = G * dL_dalpha;                  // straight-through min(0.99, .)
                            const float uy = u * dy;
                            So += u; Uy += uy;
                            Uyy = FMA(uy, dy, Uyy);
                            if (STATS && sv) {                              // the same formulas on the second pixel gradient
                                const float cd2 = FMA(b.z, dq0[k], FMA(b.w, dq1[k], c.x * dq2[k]));
                                const float dL_dalpha2 = FMA(T[k], cd2, -(Q2[k] * inv_one_m));
                                Q2[k] = FMA(w, cd2, Q2[k]);
                                const float u2 = G * dL_dalpha2;
                                const float uy2 = u2 * dy;
                                So2 += u2; Uy2 += uy2;
                            }
                        }
                    }
                }
                {   // (every visited entry has a contributing pixel: see above)
                    float* r = &sRed[wave][0][0];
                    float* part_base_ptr = &sPart[wave][0][0];
                    const float Sx = dx * So;                  // all of these still lack the factor o (commit)
                    r[0 * 68 + lane] = Sx;  r[1 * 68 + lane] = Uy;  r[2 * 68 + lane] = dx * Sx;
                    r[3 * 68 + lane] = dx * Uy; r[4 * 68 + lane] = Uyy; r[5 * 68 + lane] = So;
                    r[6 * 68 + lane] = Sc0;
                    if (!RANK1) { r[7 * 68 + lane] = Sc1; r[8 * 68 + lane] = Sc2; }
                    if (STATS && sv) { r[9 * 68 + lane] = dx * So2; r[10 * 68 + lane] = Uy2; }
                    wave_sync();
                    // lane (v, p) = (lane >> 3, lane & 7) adds the 8 floats [8p, 8p + 8) of value v (two ds_read_b128) and ONE
                    // float of the ninth value (Sc2): its 64 floats are spread over all lanes instead of a masked second
                    // pass by eight of them.  Three fused DPP adds finish the 8-lane groups (the last one is a row
                    // shift: only the group's first lane, which does the write, needs the right partner); the
                    // eight per-group partials of Sc2 are added by the commit lanes.
                    const int rv = lane >> 3, rp = lane & 7;
                    const float4 q0 = *reinterpret_cast<const float4*>(r + rv * 68 + 8 * rp);
                    const float4 q1 = *reinterpret_cast<const float4*>(r + rv * 68 + 8 * rp + 4);
                    // (RANK1: seven values -- the lanes of rv == 7 add a stale row nobody reads, and there is no ninth value)
                    float s8 = RANK1 ? 0.0f : r[8 * 68 + lane];
                    float s = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
                    s += dpp_f<DPP_QUAD_XOR1>(s);  if (!RANK1) s8 += dpp_f<DPP_QUAD_XOR1>(s8);
                    s += dpp_f<DPP_QUAD_XOR2>(s);  if (!RANK1) s8 += dpp_f<DPP_QUAD_XOR2>(s8);
                    s += dpp_bc_f<DPP_ROW_SHL4>(s); if (!RANK1) s8 += dpp_bc_f<DPP_ROW_SHL4>(s8);  // lanes 0-3 / 8-11 of a row: + the next quad
                    if (RANK1) asm volatile("" : "+v"(s));
                    else asm volatile("" : "+v"(s), "+v"(s8));     // keep the adds here (fused v_add_f32_dpp) instead of sunk behind the branch
                    if (rp == 0) {                           // slots 0..7 = Sx Sy Sxx Sxy Syy So Sc0 Sc1, 8..15 = partials of Sc2
                        float* pe = reinterpret_cast<float*>(part_base_ptr + (size_t)(j & 15) * 16);
                        pe[rv] = s;
                        if (!RANK1) pe[8 + rv] = s8;
                    }
                    if (STATS && sv) {                       // rows 9, 10 (Sx', Sy'): the same 8-float reads + three DPP adds
                        const int rr = 9 + (rv & 1);
                        const float4 t0 = *reinterpret_cast<const float4*>(r + rr * 68 + 8 * rp);
                        const float4 t1 = *reinterpret_cast<const float4*>(r + rr * 68 + 8 * rp + 4);
                        float t = ((t0.x + t0.y) + (t0.z + t0.w)) + ((t1.x + t1.y) + (t1.z + t1.w));
                        t += dpp_f<DPP_QUAD_XOR1>(t);
                        t += dpp_f<DPP_QUAD_XOR2>(t);
                        t += dpp_bc_f<DPP_ROW_SHL4>(t);
                        asm volatile("" : "+v"(t));
                        if (rp == 0 && rv < 2) sPart2[wave][j & 15][rv] = t;
                    }
                    wave_sync();
                }
            }
            if (((nz >> jb) & 0xFFFFull) != 0ull) {
                // commit the (up to) 16 entries of this group.  Records exist only for entries the forward composited
                // somewhere (non-zero mask = BinningState::touched of their slot): the others are neither written nor read
                // (run_reduce_kernel).
                wave_sync();
                if (lane < 16 && ((nz >> (jb + lane)) & 1ull) != 0ull) {
                    const int e = jb + lane;
                    const float4* pp = reinterpret_cast<const float4*>(&sPart[wave][lane][0]);
                    float4 p0 = pp[0], p1 = pp[1];
                    float p2 = 0.0f;
                    if (!RANK1) {
                        const float4 c0 = pp[2], c1 = pp[3];
                        p2 = ((c0.x + c0.y) + (c0.z + c0.w)) + ((c1.x + c1.y) + (c1.z + c1.w));
                    }
                    // p0 = (Sx, Sy, Sxx, Sxy) / o   p1 = (Syy / o, So, Sc0, Sc1)   p2 = Sc2
                    const float4 ea = sRec[wave][e].a;
                    const float4 eb = sRec[wave][e].b;
                    p0.x *= eb.y; p0.y *= eb.y; p0.z *= eb.y; p0.w *= eb.y; p1.x *= eb.y;
                    F3* g = reinterpret_cast<F3*>(part + E3_REC_FLOATS * (size_t)__float_as_uint(sRec[wave][e].c.w));
                    // dG/d(delta) = -G (A dx + B dy, C dy + B dx); d(delta)/d(ndc) = (W/2, H/2)
                    g[0] = F3{-(ea.z * p0.x + ea.w * p0.y) * ddelx_dx, -(eb.x * p0.y + ea.w * p0.x) * ddely_dy, -0.5f * p0.z};
                    g[1] = F3{-p0.w, -0.5f * p1.x, p1.y};
                    g[2] = RANK1 ? F3{rw0 * p1.z, rw1 * p1.z, rw2 * p1.z} : F3{p1.z, p1.w, p2};
                    if (STATS && sv) {                       // the same mean2D formula on the second chain's sums
                        const float sx = sPart2[wave][lane][0] * eb.y, sy = sPart2[wave][lane][1] * eb.y;
                        float* g2 = part2 + 2 * (size_t)__float_as_uint(sRec[wave][e].c.w);
                        g2[0] = -(ea.z * sx + ea.w * sy) * ddelx_dx;
                        g2[1] = -(eb.x * sy + ea.w * sx) * ddely_dy;
                    }
                }
                wave_sync();
            }
        }
        wave_sync();
    }
    // list entries behind the last contributor of every pixel are never walked: zero records for those the forward
    // evaluated (its mask bytes behind its own early exit are stale: a spurious zero record there is never read, the
    // slot's `touched` flag is 0)
    for (uint32_t e = range.x + (uint32_t)n + (uint32_t)lane; e < range.y; e += WAVE) {
        if (strip_mask[e] == 0) continue;
        F3* g = reinterpret_cast<F3*>(part + E3_REC_FLOATS * (size_t)perm[e]);
        const F3 z3 = F3{0.0f, 0.0f, 0.0f};
        g[0] = z3; g[1] = z3; g[2] = z3;
        if (STATS && sv) { part2[2 * (size_t)perm[e]] = 0.0f; part2[2 * (size_t)perm[e] + 1] = 0.0f; }
    }
    if (trace && lane == 0) {
        trace[E3_TRACE_WORDS * (size_t)tile + 0] = t_start;
        trace[E3_TRACE_WORDS * (size_t)tile + 1] = wall_clock64();
        trace[E3_TRACE_WORDS * (size_t)tile + 2] = ((unsigned long long)(range.y - range.x) << 32) | (unsigned)n;
        trace[E3_TRACE_WORDS * (size_t)tile + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) |
                                      (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        trace[E3_TRACE_WORDS * (size_t)tile + 4] = c_start;                    // effective clock of the kernel = cycles / wall time
        trace[E3_TRACE_WORDS * (size_t)tile + 5] = __builtin_readcyclecounter();
        (void)t_loop;
    }
}

#define E3_RENDER_BWD_PARAMS                                                                                             \
    unsigned long long *__restrict__ trace, int ntiles, int tiles_per_view, const uint32_t *__restrict__ order, int gx,  \
        int W, int H, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ emit_gid,                           \
        const float4 *__restrict__ rec, const float *__restrict__ bg, const float *__restrict__ final_T,                 \
        const uint32_t *__restrict__ n_contrib, const uint32_t *__restrict__ perm,                                       \
        const uint8_t *__restrict__ strip_mask, const float *__restrict__ dL_dpix, float *__restrict__ part,             \
        const uint32_t *__restrict__ lpt_cnt, const uint32_t *__restrict__ lpt_list, uint32_t lpt_cap, Rank1Views r1
#define E3_RENDER_BWD_ARGS \
    trace, ntiles, tiles_per_view, order, gx, W, H, ranges, emit_gid, rec, bg, final_T, n_contrib, perm, strip_mask, dL_dpix, part
// One wave = one tile: find it, then run the body its VIEW asks for (wave-uniform): the rank-1 body where the view's pixel
// gradient is a scalar field times a weight vector, the general (STATS: second-chain) body otherwise.  Both share the LDS block.
template <bool STATS, bool FAST, bool ANY_RANK1>
__device__ __forceinline__ void render_bwd_dispatch(
    unsigned long long* __restrict__ trace, int ntiles, int tiles_per_view, const uint32_t* __restrict__ order, int gx,
    int W, int H, const uint2* __restrict__ ranges, const uint32_t* __restrict__ emit_gid,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ perm, const uint8_t* __restrict__ strip_mask, const float* __restrict__ dL_dpix,
    float* __restrict__ part, const float* __restrict__ dL_dpix2, float* __restrict__ part2,
    const uint32_t* __restrict__ lpt_cnt, const uint32_t* __restrict__ lpt_list, uint32_t lpt_cap, const Rank1Views& r1) {
    __shared__ BwdShared<STATS> sh;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * BWD_WAVES + wave;
    if (unit >= ntiles) return;                 // ntiles = launch slots (see render_fwd_kernel)
    // global tile id (made scalar: see render_fwd_kernel)
    const int tile = __builtin_amdgcn_readfirstlane(order ? (int)order[unit] : lpt_lookup(lpt_cnt, lpt_list, lpt_cap, (uint32_t)unit, lane));
    if (tile < 0) return;
    if (ANY_RANK1) {
        const int view = tile / tiles_per_view;
        if ((r1.mask >> view) & 1u) {
            render_bwd_body<STATS, false, FAST, true>(sh, tile, trace, tiles_per_view, gx, W, H, ranges, emit_gid, rec, bg, final_T,
                                                      n_contrib, perm, strip_mask, dL_dpix, part, nullptr, nullptr,
                                                      r1.w[view][0], r1.w[view][1], r1.w[view][2]);
            return;
        }
    }
    render_bwd_body<STATS, STATS, FAST, false>(sh, tile, trace, tiles_per_view, gx, W, H, ranges, emit_gid, rec, bg, final_T,
                                               n_contrib, perm, strip_mask, dL_dpix, part, dL_dpix2, part2, 0.0f, 0.0f, 0.0f);
}
// (plain kernels around the one dispatch: the profiles, the bench line and the reviews name `render_bwd_kernel`.  Both
// bodies are compiled into it; a call without rank-1 views passes mask 0 and every tile takes the general body)
__global__ __launch_bounds__(BWD_WAVES * WAVE, E3_BWD_WAVES) void render_bwd_kernel(E3_RENDER_BWD_PARAMS) {
    render_bwd_dispatch<false, false, true>(E3_RENDER_BWD_ARGS, nullptr, nullptr, lpt_cnt, lpt_list, lpt_cap, r1);
}
__global__ __launch_bounds__(BWD_WAVES * WAVE, E3_BWD_WAVES) void render_bwd_fast_kernel(E3_RENDER_BWD_PARAMS) {
    render_bwd_dispatch<false, true, true>(E3_RENDER_BWD_ARGS, nullptr, nullptr, lpt_cnt, lpt_list, lpt_cap, r1);
}
__global__ __launch_bounds__(BWD_WAVES * WAVE, E3_BWD_STATS_WAVES) void render_bwd_stats_kernel(
    E3_RENDER_BWD_PARAMS, const float* __restrict__ dL_dpix2, float* __restrict__ part2) {
    render_bwd_dispatch<true, false, true>(E3_RENDER_BWD_ARGS, dL_dpix2, part2, lpt_cnt, lpt_list, lpt_cap, r1);
}
__global__ __launch_bounds__(BWD_WAVES * WAVE, E3_BWD_STATS_WAVES) void render_bwd_stats_fast_kernel(
    E3_RENDER_BWD_PARAMS, const float* __restrict__ dL_dpix2, float* __restrict__ part2) {
    render_bwd_dispatch<true, true, true>(E3_RENDER_BWD_ARGS, dL_dpix2, part2, lpt_cnt, lpt_list, lpt_cap, r1);
}

// ------------------------------------------------------------------------------------ per-Gaussian backward
template <bool ACCUM>
__device__ __forceinline__ void put(float* p, float v) { if (ACCUM) *p += v; else *p = v; }

// D (degree) is a template argument: the coefficients then live in a statically indexed register array that is loaded in
// one batch and -- coefficient by coefficient -- overwritten with the gradient it produces, which is stored in one batch
// at the end.  VEC4: the reference's (P, M, 3) layout with 16-byte aligned rows, moved as float4 (a quarter of the
// memory instructions and cache-line requests of dword accesses at a 192-byte lane stride); otherwise element (k, ch)
// lives at [(k * 3 + ch) * st] (st = P: coefficient-major).  ACCUM adds into dsh (VEC4 is not used with it).
template <bool ACCUM, int D, bool VEC4>
__device__ __forceinline__ void sh_backward(int M, const float* __restrict__ sh, float* __restrict__ dsh, size_t st,
                                            float mx, float my, float mz, const float* __restrict__ campos,
                                            uint32_t clamped, const float gin[3], float gmean[3]) {
    constexpr int NC = 3 * (D + 1) * (D + 1);
    float cf[NC];
    if (VEC4) {
        const float4* __restrict__ s4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
        for (int k = 0; k < NC / 4; ++k) {
            const float4 q = s4[k];
            cf[4 * k] = q.x; cf[4 * k + 1] = q.y; cf[4 * k + 2] = q.z; cf[4 * k + 3] = q.w;
        }
#pragma unroll
        for (int k = NC / 4 * 4; k < NC; ++k) cf[k] = sh[k];
    } else {
#pragma unroll
        for (int k = 0; k < NC; ++k) cf[k] = sh[(size_t)k * st];
    }
    const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                         0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    float ox = mx - campos[0], oy = my - campos[1], oz = mz - campos[2];
    float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    float g[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) g[ch] = ((clamped >> ch) & 1u) ? 0.0f : gin[ch];
    float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
    auto term = [&](int k, float Y, float Yx, float Yy, float Yz) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float s = cf[k * 3 + ch] * g[ch];
            cf[k * 3 + ch] = Y * g[ch];                // the coefficient is consumed: its slot now holds dL/dsh[k][ch]
            ddx = FMA(Yx, s, ddx); ddy = FMA(Yy, s, ddy); ddz = FMA(Yz, s, ddz);
        }
    };
    term(0, SH_C0, 0.0f, 0.0f, 0.0f);
    if (D > 0) {
        term(1, -SH_C1 * y, 0.0f, -SH_C1, 0.0f);
        term(2, SH_C1 * z, 0.0f, 0.0f, SH_C1);
        term(3, -SH_C1 * x, -SH_C1, 0.0f, 0.0f);
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            term(4, C2[0] * xy, C2[0] * y, C2[0] * x, 0.0f);
            term(5, C2[1] * yz, 0.0f, C2[1] * z, C2[1] * y);
            term(6, C2[2] * (2.0f * zz - xx - yy), C2[2] * -2.0f * x, C2[2] * -2.0f * y, C2[2] * 4.0f * z);
            term(7, C2[3] * xz, C2[3] * z, 0.0f, C2[3] * x);
            term(8, C2[4] * (xx - yy), C2[4] * 2.0f * x, C2[4] * -2.0f * y, 0.0f);
            if (D > 2) {
                term(9, C3[0] * y * (3.0f * xx - yy), C3[0] * 6.0f * xy, C3[0] * (3.0f * xx - 3.0f * yy), 0.0f);
                term(10, C3[1] * xy * z, C3[1] * yz, C3[1] * xz, C3[1] * xy);
                term(11, C3[2] * y * (4.0f * zz - xx - yy), C3[2] * -2.0f * xy, C3[2] * (4.0f * zz - xx - 3.0f * yy),
                     C3[2] * 8.0f * yz);
                term(12, C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), C3[3] * -6.0f * xz, C3[3] * -6.0f * yz,
                     C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy));
                term(13, C3[4] * x * (4.0f * zz - xx - yy), C3[4] * (4.0f * zz - 3.0f * xx - yy), C3[4] * -2.0f * xy,
                     C3[4] * 8.0f * xz);
                term(14, C3[5] * z * (xx - yy), C3[5] * 2.0f * xz, C3[5] * -2.0f * yz, C3[5] * (xx - yy));
                term(15, C3[6] * x * (xx - 3.0f * yy), C3[6] * (3.0f * xx - 3.0f * yy), C3[6] * -6.0f * xy, 0.0f);
                if (D > 3) {     // utils/sh_utils.py:97-110: partial derivatives of the polynomials as written there
                    const float C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f,
                                         -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f,
                                         0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};
                    term(16, C4[0] * xy * (xx - yy), C4[0] * y * (3.0f * xx - yy), C4[0] * x * (xx - 3.0f * yy), 0.0f);
                    term(17, C4[1] * yz * (3.0f * xx - yy), C4[1] * 6.0f * xy * z, C4[1] * z * (3.0f * xx - 3.0f * yy),
                         C4[1] * y * (3.0f * xx - yy));
                    term(18, C4[2] * xy * (7.0f * zz - 1.0f), C4[2] * y * (7.0f * zz - 1.0f), C4[2] * x * (7.0f * zz - 1.0f),
                         C4[2] * 14.0f * xy * z);
                    term(19, C4[3] * yz * (7.0f * zz - 3.0f), 0.0f, C4[3] * z * (7.0f * zz - 3.0f), C4[3] * y * (21.0f * zz - 3.0f));
                    term(20, C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f), 0.0f, 0.0f, C4[4] * z * (140.0f * zz - 60.0f));
                    term(21, C4[5] * xz * (7.0f * zz - 3.0f), C4[5] * z * (7.0f * zz - 3.0f), 0.0f, C4[5] * x * (21.0f * zz - 3.0f));
                    term(22, C4[6] * (xx - yy) * (7.0f * zz - 1.0f), C4[6] * 2.0f * x * (7.0f * zz - 1.0f),
                         C4[6] * -2.0f * y * (7.0f * zz - 1.0f), C4[6] * 14.0f * z * (xx - yy));
                    term(23, C4[7] * xz * (xx - 3.0f * yy), C4[7] * z * (3.0f * xx - 3.0f * yy), C4[7] * -6.0f * xy * z,
                         C4[7] * x * (xx - 3.0f * yy));
                    term(24, C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)), C4[8] * x * (4.0f * xx - 12.0f * yy),
                         C4[8] * y * (4.0f * yy - 12.0f * xx), 0.0f);
                }
            }
        }
    }
    if (VEC4 && !ACCUM) {
        float4* __restrict__ d4 = reinterpret_cast<float4*>(dsh);
#pragma unroll
        for (int k = 0; k < NC / 4; ++k) d4[k] = make_float4(cf[4 * k], cf[4 * k + 1], cf[4 * k + 2], cf[4 * k + 3]);
#pragma unroll
        for (int k = NC / 4 * 4; k < NC; ++k) dsh[k] = cf[k];
    } else {
#pragma unroll
        for (int k = 0; k < NC; ++k) put<ACCUM>(&dsh[(size_t)k * st], cf[k]);
    }
    if (!ACCUM)
        for (int k = NC; k < 3 * M; ++k) dsh[(size_t)k * st] = 0.0f;
    float dot = x * ddx + y * ddy + z * ddz;
    gmean[0] += (ddx - x * dot) / len;
    gmean[1] += (ddy - y * dot) / len;
    gmean[2] += (ddz - z * dot) / len;
}

// ---- shared pieces of the per-Gaussian backward (one view's contribution, then the view-independent tail)

// Sigma3 = (R diag s)(R diag s)^T from activated scale / unit quaternion.
struct Cov3 {
    float R[3][3];
    float s[3];
    float S[6];
};
__device__ __forceinline__ void build_cov3(const float sact[3], float scale_modifier, float qr, float qx, float qy,
                                           float qz, Cov3& c) {
    const float s0 = scale_modifier * sact[0], s1 = scale_modifier * sact[1], s2 = scale_modifier * sact[2];
    c.s[0] = s0; c.s[1] = s1; c.s[2] = s2;
    float (*R)[3] = c.R;
    R[0][0] = 1.0f - 2.0f * FMA(qy, qy, qz * qz); R[0][1] = 2.0f * FMA(qx, qy, -(qr * qz)); R[0][2] = 2.0f * FMA(qx, qz, qr * qy);
    R[1][0] = 2.0f * FMA(qx, qy, qr * qz); R[1][1] = 1.0f - 2.0f * FMA(qx, qx, qz * qz); R[1][2] = 2.0f * FMA(qy, qz, -(qr * qx));
    R[2][0] = 2.0f * FMA(qx, qz, -(qr * qy)); R[2][1] = 2.0f * FMA(qy, qz, qr * qx); R[2][2] = 1.0f - 2.0f * FMA(qx, qx, qy * qy);
    float L[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { L[a][0] = R[a][0] * s0; L[a][1] = R[a][1] * s1; L[a][2] = R[a][2] * s2; }
    float* S = c.S;
    S[0] = FMA(L[0][0], L[0][0], FMA(L[0][1], L[0][1], L[0][2] * L[0][2]));
    S[1] = FMA(L[0][0], L[1][0], FMA(L[0][1], L[1][1], L[0][2] * L[1][2]));
    S[2] = FMA(L[0][0], L[2][0], FMA(L[0][1], L[2][1], L[0][2] * L[2][2]));
    S[3] = FMA(L[1][0], L[1][0], FMA(L[1][1], L[1][1], L[1][2] * L[1][2]));
    S[4] = FMA(L[1][0], L[2][0], FMA(L[1][1], L[2][1], L[1][2] * L[2][2]));
    S[5] = FMA(L[2][0], L[2][0], FMA(L[2][1], L[2][1], L[2][2] * L[2][2]));
}

// Per-splat sums of the per-instance gradient records.  A splat's records sit contiguously at its emission
// positions, so the 64 depth-consecutive splats of a WAVE own one contiguous slot range: the wave streams it through
// its private LDS slice with coalesced 16-byte loads (128 packed 36-byte records per chunk; no workgroup barrier, the
// four waves of a workgroup run independently) and every lane adds the records of its own run from LDS, in slot
// order (fixed order -> deterministic).  Output: 3 float4 per splat at its INDEX q,
// (mx my A B | C o c0 c1 | c2 - - -), which the per-Gaussian kernels then read coalesced.
constexpr int RR_CHUNK = 128;
#ifndef E3_RR_PACKED_STORES
#define E3_RR_PACKED_STORES 1
#endif
__global__ __launch_bounds__(256) void run_reduce_kernel(uint32_t Q, const uint32_t* __restrict__ order,
                                                         const uint2* __restrict__ run_sorted,
                                                         const float* __restrict__ part, float4* __restrict__ gsum,
                                                         const uint32_t* __restrict__ nvis /* [1]: length of the order */,
                                                         const uint8_t* __restrict__ touched /* per slot */,
                                                         const uint32_t* __restrict__ count_dev, uint32_t capacity) {
    __shared__ float sbuf[4][RR_CHUNK * E3_REC_FLOATS];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t j0 = blockIdx.x * 256u + wave * 64u, j = j0 + lane;
    // the depth-sorted order holds only the splats the projection kept (the depth sort dropped the others): nothing to
    // sum, and no sum to store, for the rest -- the per-Gaussian kernels never read the sums of culled splats
    Q = __builtin_amdgcn_readfirstlane(*nvis);
    if (j0 >= Q) return;
    // (forward with pre-sized buffers whose count did not fit: nothing was emitted, `run_sorted` is stale)
    if (count_dev && *count_dev > capacity) return;
    const uint32_t nl = (Q - j0 < 64u ? Q - j0 : 64u) - 1u;          // last lane with a splat
    const uint2 rn = j < Q ? run_sorted[j] : make_uint2(0u, 0u);
    const uint32_t S0 = __builtin_amdgcn_readfirstlane(rn.x);
    const uint32_t S1 = __builtin_amdgcn_readlane(rn.x + rn.y, nl);
    float a[E3_REC_FLOATS];
#pragma unroll
    for (int k = 0; k < E3_REC_FLOATS; ++k) a[k] = 0.0f;
    float* sb = sbuf[wave];
    const uint32_t end = rn.x + rn.y;
    for (uint32_t c0 = S0; c0 < S1; c0 += RR_CHUNK) {
        const uint32_t nrec = (S1 - c0 < (uint32_t)RR_CHUNK) ? S1 - c0 : (uint32_t)RR_CHUNK;
        // Stage the chunk: lane l brings records l and l + 64.  Only records whose slot is flagged `touched` exist (the
        // compositing backward writes nothing for the ~55 % of the instances the forward never evaluated); the others
        // are staged as zeros without touching memory -- the loads of unflagged lanes are redirected to the chunk's
        // first record (one cache line, unconditional loads: predicated ones serialise) and discarded.
        float v[2][E3_REC_FLOATS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t r = lane + 64u * t;
            const bool f = r < nrec && touched[c0 + r] != 0;
            const float* __restrict__ src = part + E3_REC_FLOATS * (size_t)(c0 + (f ? r : 0u));
#pragma unroll
            for (int k = 0; k < E3_REC_FLOATS; ++k) v[t][k] = src[k];
#pragma unroll
            for (int k = 0; k < E3_REC_FLOATS; ++k) v[t][k] = f ? v[t][k] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < E3_REC_FLOATS; ++k) sb[E3_REC_FLOATS * (lane + 64u * t) + k] = v[t][k];
        wave_sync();
        const uint32_t lo = rn.x > c0 ? rn.x : c0;
        const uint32_t cend = c0 + nrec;
        const uint32_t hi = end < cend ? end : cend;
        for (uint32_t r = lo; r < hi; ++r) {
            const float* q = sb + E3_REC_FLOATS * (r - c0);
#pragma unroll
            for (int k = 0; k < E3_REC_FLOATS; ++k) a[k] += q[k];
        }
        wave_sync();
    }
    // (zeros for splats whose every tile was culled: their radius can still be > 0)
#if E3_RR_PACKED_STORES
    // The sums go to the splat's INDEX q: 48 bytes at a random place.  As three 16-byte stores per lane those are three
    // partial-sector write transactions per splat, and it is their NUMBER that costs (a 4-byte scatter of the same count
    // costs the same, EXPERIMENTS.md round 5).  Through the wave's LDS slice the three float4 of a splat are handed to three
    // CONSECUTIVE lanes: one store instruction then writes 21 whole 48-byte blocks, each as one transaction.
    {
        float4* s4 = reinterpret_cast<float4*>(sb);                 // 64 x 3 float4 (3 KB of the wave's 4.5 KB slice)
        uint32_t* sq = reinterpret_cast<uint32_t*>(sb + 64 * 12);   // the splats' indices (+ 256 B)
        s4[3 * lane] = make_float4(a[0], a[1], a[2], a[3]);
        s4[3 * lane + 1] = make_float4(a[4], a[5], a[6], a[7]);
        s4[3 * lane + 2] = make_float4(a[8], 0.0f, 0.0f, 0.0f);
        sq[lane] = j < Q ? order[j] : 0xFFFFFFFFu;
        wave_sync();
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint32_t idx = lane + 64u * t, sp = idx / 3u, part = idx - 3u * sp;
            const uint32_t q = sq[sp];
            if (q != 0xFFFFFFFFu) gsum[3 * (size_t)q + part] = s4[idx];
        }
    }
#else
    if (j < Q) {
        const size_t q = order[j];
        gsum[3 * q] = make_float4(a[0], a[1], a[2], a[3]);
        gsum[3 * q + 1] = make_float4(a[4], a[5], a[6], a[7]);
        gsum[3 * q + 2] = make_float4(a[8], 0.0f, 0.0f, 0.0f);
    }
#endif
}

// Few splats with long runs (point-cloud initialisation: thousands of Gaussians covering hundreds of tiles each): one
// wave per splat.  Lane l adds records l, l + 64, ... of the run, then a fixed butterfly adds the lanes -- the
// order depends only on the run length, so the result is as reproducible as the streaming kernel's.
__global__ __launch_bounds__(256) void run_reduce_wave_kernel(uint32_t Q, const uint32_t* __restrict__ order,
                                                              const uint2* __restrict__ run_sorted,
                                                              const float* __restrict__ part,
                                                              float4* __restrict__ gsum,
                                                              const uint8_t* __restrict__ touched,
                                                              const uint32_t* __restrict__ nvis,
                                                              const uint32_t* __restrict__ count_dev, uint32_t capacity) {
    const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (j >= *nvis) return;                  // (the order holds the kept splats only)
    if (count_dev && *count_dev > capacity) return;
    const uint2 rn = run_sorted[j];
    const float* __restrict__ p = part + E3_REC_FLOATS * (size_t)rn.x;
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t r = lane; r < rn.y; r += 64u) {
        if (touched[rn.x + r] == 0) continue;            // no record was written for this slot
#pragma unroll
        for (int k = 0; k < E3_REC_FLOATS; ++k) a[k] += p[E3_REC_FLOATS * (size_t)r + k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] += __shfl_xor(a[k], o, 64);
    }
    if (lane == 0) {
        const size_t q = order[j];
        gsum[3 * q] = make_float4(a[0], a[1], a[2], a[3]);
        gsum[3 * q + 1] = make_float4(a[4], a[5], a[6], a[7]);
        gsum[3 * q + 2] = make_float4(a[8], 0.0f, 0.0f, 0.0f);
    }
}

// STATS (render_bwd_stats_kernel): per-splat sums of the second chain's two floats per instance, for the splats of view
// 0 only, into the free components .y / .z of the splat's third sum vector (run_reduce_kernel wrote (c2, 0, 0, 0)
// there: this kernel runs after it).  One thread per kept splat, its run added in slot order (fixed order).
__global__ __launch_bounds__(256) void run_reduce_stats_kernel(const uint32_t* __restrict__ order,
                                                               const uint2* __restrict__ run_sorted,
                                                               const float* __restrict__ part2, float4* __restrict__ gsum,
                                                               const uint32_t* __restrict__ nvis,
                                                               const uint8_t* __restrict__ touched, uint32_t nviews,
                                                               const uint32_t* __restrict__ count_dev, uint32_t capacity) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= *nvis) return;
    if (count_dev && *count_dev > capacity) return;
    const size_t q = order[j];
    if (q % nviews != 0) return;                     // splat q = Gaussian * nviews + view
    const uint2 rn = run_sorted[j];
    float sx = 0.0f, sy = 0.0f;
    for (uint32_t r = rn.x; r < rn.x + rn.y; ++r) {
        if (touched[r] == 0) continue;
        sx += part2[2 * (size_t)r]; sy += part2[2 * (size_t)r + 1];
    }
    float4 v = gsum[3 * q + 2];
    v.y = sx; v.z = sy;
    gsum[3 * q + 2] = v;
}

__device__ __forceinline__ void load_sums(const float4* __restrict__ gsum, size_t q, float g12[9]) {
    const float4 s0 = gsum[3 * q], s1 = gsum[3 * q + 1];
    const float s2 = gsum[3 * q + 2].x;
    g12[0] = s0.x; g12[1] = s0.y; g12[2] = s0.z; g12[3] = s0.w;
    g12[4] = s1.x; g12[5] = s1.y; g12[6] = s1.z; g12[7] = s1.w; g12[8] = s2;
}

// One view: conic -> Sigma2 -> (Sigma3, view-space mean) and NDC mean -> world mean.
// Writes gcov[6] (dL/dSigma3 of this view) and gmean[3] (dL/dmean3D of this view, without the SH term).
__device__ __forceinline__ void view_geom_backward(const ViewParams& vp, float mx, float my, float mz,
                                                   const float S[6], const float g12[9], float gcov[6],
                                                   float gmean[3]) {
    const float* V = vp.view;
    const float* Pm = vp.proj;
    // ---- recompute the EWA intermediates
    float vx = XFORM(V, 0, mx, my, mz), vy = XFORM(V, 1, mx, my, mz), vz = XFORM(V, 2, mx, my, mz);
    float limx = E3_GUARD_BAND * vp.tanfovx, limy = E3_GUARD_BAND * vp.tanfovy;
    float txtz = vx / vz, tytz = vy / vz;
    float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    float tz = vz;
    float xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    float ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
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
    float a = FMA(T00, u0, FMA(T01, u1, T02 * u2)) + E3_DILATION;
    float b = FMA(T10, u0, FMA(T11, u1, T12 * u2));
    float c = FMA(T10, w0, FMA(T11, w1, T12 * w2)) + E3_DILATION;
    // ---- conic -> (a,b,c)
    float gA = g12[2], gB = g12[3], gC = g12[4];
    float det = a * c - b * b;
    float d2inv = 1.0f / (det * det + E3_DET2_EPS);
    float g_a = 0.0f, g_b = 0.0f, g_c = 0.0f;
    if (det != 0.0f) {
        g_a = d2inv * (-c * c * gA + b * c * gB - b * b * gC);
        g_c = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
        g_b = d2inv * (2.0f * b * c * gA - (a * c + b * b) * gB + 2.0f * a * b * gC);
    }
    gcov[0] = T00 * T00 * g_a + T00 * T10 * g_b + T10 * T10 * g_c;
    gcov[3] = T01 * T01 * g_a + T01 * T11 * g_b + T11 * T11 * g_c;
    gcov[5] = T02 * T02 * g_a + T02 * T12 * g_b + T12 * T12 * g_c;
    gcov[1] = 2.0f * T00 * T01 * g_a + (T00 * T11 + T01 * T10) * g_b + 2.0f * T10 * T11 * g_c;
    gcov[2] = 2.0f * T00 * T02 * g_a + (T00 * T12 + T02 * T10) * g_b + 2.0f * T10 * T12 * g_c;
    gcov[4] = 2.0f * T02 * T01 * g_a + (T01 * T12 + T02 * T11) * g_b + 2.0f * T11 * T12 * g_c;
    float gT00 = 2.0f * g_a * u0 + g_b * w0, gT01 = 2.0f * g_a * u1 + g_b * w1, gT02 = 2.0f * g_a * u2 + g_b * w2;
    float gT10 = 2.0f * g_c * w0 + g_b * u0, gT11 = 2.0f * g_c * w1 + g_b * u1, gT12 = 2.0f * g_c * w2 + g_b * u2;
    float gJ00 = V[0] * gT00 + V[4] * gT01 + V[8] * gT02;
    float gJ02 = V[2] * gT00 + V[6] * gT01 + V[10] * gT02;
    float gJ11 = V[1] * gT10 + V[5] * gT11 + V[9] * gT12;
    float gJ12 = V[2] * gT10 + V[6] * gT11 + V[10] * gT12;
    float itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
    float gtx = xmul * (-vp.focal_x * itz2 * gJ02);
    float gty = ymul * (-vp.focal_y * itz2 * gJ12);
    float gtz = -vp.focal_x * itz2 * gJ00 - vp.focal_y * itz2 * gJ11 + (2.0f * vp.focal_x * tx) * itz3 * gJ02 +
                (2.0f * vp.focal_y * ty) * itz3 * gJ12;
    gmean[0] = V[0] * gtx + V[1] * gty + V[2] * gtz;
    gmean[1] = V[4] * gtx + V[5] * gty + V[6] * gtz;
    gmean[2] = V[8] * gtx + V[9] * gty + V[10] * gtz;
    // ---- NDC mean gradient through the projection
    float gm2x = g12[0], gm2y = g12[1];
    float hx = XFORM(Pm, 0, mx, my, mz), hy = XFORM(Pm, 1, mx, my, mz), hw = XFORM(Pm, 3, mx, my, mz);
    float pw = 1.0f / (hw + E3_W_EPS);
    float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
    gmean[0] += (Pm[0] * pw - Pm[3] * mul1) * gm2x + (Pm[1] * pw - Pm[3] * mul2) * gm2y;
    gmean[1] += (Pm[4] * pw - Pm[7] * mul1) * gm2x + (Pm[5] * pw - Pm[7] * mul2) * gm2y;
    gmean[2] += (Pm[8] * pw - Pm[11] * mul1) * gm2x + (Pm[9] * pw - Pm[11] * mul2) * gm2y;
}

// dL/dSigma3 -> (scale, quaternion); with PREACT also through exp / normalize.
__device__ __forceinline__ void cov3_backward(const Cov3& c, const float gcov[6], float scale_modifier, float qr,
                                              float qx, float qy, float qz, bool preact, const float sact[3],
                                              float qinv, float ds[3], float dq[4]) {
    const float (*R)[3] = c.R;
    const float* s = c.s;
    const float Gs[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                            {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                            {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
    float dR[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float dl = 2.0f * (Gs[r][0] * (R[0][j] * s[j]) + Gs[r][1] * (R[1][j] * s[j]) + Gs[r][2] * (R[2][j] * s[j]));
            t += R[r][j] * dl;
            dR[r][j] = s[j] * dl;
        }
        ds[j] = scale_modifier * t;
    }
    dq[0] = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
    dq[1] = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.0f * qx * dR[1][1] - qr * dR[1][2] + qz * dR[2][0] + qr * dR[2][1] - 2.0f * qx * dR[2][2]);
    dq[2] = 2.0f * (-2.0f * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - qr * dR[2][0] + qz * dR[2][1] - 2.0f * qy * dR[2][2]);
    dq[3] = 2.0f * (-2.0f * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2.0f * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
    if (preact) {
        // scales = exp(raw): d/draw = d/dscale * scale;  rotation = raw/|raw|: project out the radial part
        ds[0] *= sact[0]; ds[1] *= sact[1]; ds[2] *= sact[2];
        float dot = qr * dq[0] + qx * dq[1] + qy * dq[2] + qz * dq[3];
        dq[0] = (dq[0] - qr * dot) * qinv; dq[1] = (dq[1] - qx * dot) * qinv;
        dq[2] = (dq[2] - qy * dot) * qinv; dq[3] = (dq[3] - qz * dot) * qinv;
    }
}

// One thread per Gaussian.  ACCUM=false writes every output element (zeros for culled Gaussians, so the
// caller never has to pre-zero); ACCUM=true adds into the outputs for visible Gaussians only, which lets the
// renders of a training iteration accumulate straight into one gradient buffer.
template <bool ACCUM, int DEG, bool VEC4>
__global__ __launch_bounds__(256) void geom_bwd_kernel(
    int P, int M, const float* __restrict__ means, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ opac_in,
    const float* __restrict__ cov_pre, ViewParams vp, int flags, const int* __restrict__ radii,
    const uint32_t* __restrict__ clamped, const float4* __restrict__ gsum,
    float* dL_dmean2D, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool preact = (flags & E3_FLAG_PREACT) != 0;
    if (radii[i] <= 0) {
        if (!ACCUM) {
            if (dL_dmean2D) { dL_dmean2D[3 * (size_t)i] = 0.0f; dL_dmean2D[3 * (size_t)i + 1] = 0.0f; dL_dmean2D[3 * (size_t)i + 2] = 0.0f; }
            if (dL_dopacity) dL_dopacity[i] = 0.0f;
            if (dL_dcolor) { dL_dcolor[3 * (size_t)i] = 0.0f; dL_dcolor[3 * (size_t)i + 1] = 0.0f; dL_dcolor[3 * (size_t)i + 2] = 0.0f; }
            dL_dmean3D[3 * (size_t)i] = 0.0f; dL_dmean3D[3 * (size_t)i + 1] = 0.0f; dL_dmean3D[3 * (size_t)i + 2] = 0.0f;
            if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] = 0.0f;
            if (dL_dsh) {
                const bool pl = (flags & E3_FLAG_SH_PLANAR) != 0;
                float* d0 = pl ? dL_dsh + i : dL_dsh + (size_t)i * M * 3;
                const size_t st0 = pl ? (size_t)P : (size_t)1;
                for (int k = 0; k < 3 * M; ++k) d0[(size_t)k * st0] = 0.0f;
            }
            if (dL_dscale) { dL_dscale[3 * (size_t)i] = 0.0f; dL_dscale[3 * (size_t)i + 1] = 0.0f; dL_dscale[3 * (size_t)i + 2] = 0.0f; }
            if (dL_drot) for (int k = 0; k < 4; ++k) dL_drot[4 * (size_t)i + k] = 0.0f;
        }
        return;
    }
    float g12[9];
    load_sums(gsum, (size_t)i, g12);
    float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    Cov3 cv;
    float qr = 0, qx = 0, qy = 0, qz = 0, qinv = 1.0f;
    float sact[3] = {0, 0, 0};
    if (cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cv.S[k] = cov_pre[6 * (size_t)i + k];
    } else {
        float qn[4];
        act_load_scale_rot(scales + 3 * (size_t)i, rots + 4 * (size_t)i, preact, sact, qn, qinv);
        qr = qn[0]; qx = qn[1]; qy = qn[2]; qz = qn[3];
        build_cov3(sact, vp.scale_modifier, qr, qx, qy, qz, cv);
    }
    float gcov[6], gmean[3];
    view_geom_backward(vp, mx, my, mz, cv.S, g12, gcov, gmean);
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) put<ACCUM>(&dL_dcov3D[6 * (size_t)i + k], gcov[k]);
    }
    if (dL_dmean2D) {   // NDC-unit screen-space gradient (scene/gaussian_model.py:405-407); overwritten
        dL_dmean2D[3 * (size_t)i] = g12[0]; dL_dmean2D[3 * (size_t)i + 1] = g12[1]; dL_dmean2D[3 * (size_t)i + 2] = 0.0f;
    }
    {
        float go = g12[5];
        if (preact) { float o = act_sigmoid(opac_in[i]); go = go * o * (1.0f - o); }
        if (dL_dopacity) put<ACCUM>(&dL_dopacity[i], go);
    }
    if (dL_dcolor) {
        put<ACCUM>(&dL_dcolor[3 * (size_t)i], g12[6]); put<ACCUM>(&dL_dcolor[3 * (size_t)i + 1], g12[7]);
        put<ACCUM>(&dL_dcolor[3 * (size_t)i + 2], g12[8]);
    }
    if (shs) {
        float gcol[3] = {g12[6], g12[7], g12[8]};
        const bool pl = (flags & E3_FLAG_SH_PLANAR) != 0;
        sh_backward<ACCUM, DEG, VEC4>(M, pl ? shs + i : shs + (size_t)i * M * 3,
                                      pl ? dL_dsh + i : dL_dsh + (size_t)i * M * 3, pl ? (size_t)P : (size_t)1, mx, my, mz,
                                      vp.campos, clamped[i], gcol, gmean);
    }
    put<ACCUM>(&dL_dmean3D[3 * (size_t)i], gmean[0]);
    put<ACCUM>(&dL_dmean3D[3 * (size_t)i + 1], gmean[1]);
    put<ACCUM>(&dL_dmean3D[3 * (size_t)i + 2], gmean[2]);
    if (!cov_pre) {
        float ds[3], dq[4];
        cov3_backward(cv, gcov, vp.scale_modifier, qr, qx, qy, qz, preact, sact, qinv, ds, dq);
#pragma unroll
        for (int k = 0; k < 3; ++k) put<ACCUM>(&dL_dscale[3 * (size_t)i + k], ds[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) put<ACCUM>(&dL_drot[4 * (size_t)i + k], dq[k]);
    }
}

// ---- all views of one training iteration in ONE per-Gaussian pass -----------------------------------------
// The reference's loss.backward() (train.py:211) runs the rasteriser backward once per render and autograd
// adds the three results into .grad.  Here one thread per Gaussian loops over the views: parameters are read
// once, the activation / Sigma3 work is shared, dL/dSigma3 and dL/dmean are summed in registers, and every
// gradient element is written exactly once (zeros when no view saw the Gaussian), so the caller neither
// pre-zeroes the gradient buffer nor serialises the views.
struct MultiViews {
    ViewSet vs;
    const int* radii;          // (n, P)
    const uint32_t* clamped;   // per splat q = i * n + v
    const float4* gsum;        // per splat: summed gradient records (run_reduce_kernel)
    int stats;                 // view 0's screen-space output comes from the second chain's sums (gsum[3q+2].y/.z)
};

// real SH basis function k and its gradient w.r.t. the unit direction (same constants as sh_backward)
__device__ __forceinline__ void sh_basis(int k, float x, float y, float z, float& Y, float& Yx, float& Yy, float& Yz) {
    const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
    const float C30 = -0.5900435899266435f, C31 = 2.890611442640554f, C32 = -0.4570457994644658f,
                C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                C36 = -0.5900435899266435f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    switch (k) {
    case 0: Y = SH_C0; Yx = 0.0f; Yy = 0.0f; Yz = 0.0f; break;
    case 1: Y = -SH_C1 * y; Yx = 0.0f; Yy = -SH_C1; Yz = 0.0f; break;
    case 2: Y = SH_C1 * z; Yx = 0.0f; Yy = 0.0f; Yz = SH_C1; break;
    case 3: Y = -SH_C1 * x; Yx = -SH_C1; Yy = 0.0f; Yz = 0.0f; break;
    case 4: Y = C20 * xy; Yx = C20 * y; Yy = C20 * x; Yz = 0.0f; break;
    case 5: Y = C21 * yz; Yx = 0.0f; Yy = C21 * z; Yz = C21 * y; break;
    case 6: Y = C22 * (2.0f * zz - xx - yy); Yx = C22 * -2.0f * x; Yy = C22 * -2.0f * y; Yz = C22 * 4.0f * z; break;
    case 7: Y = C23 * xz; Yx = C23 * z; Yy = 0.0f; Yz = C23 * x; break;
    case 8: Y = C24 * (xx - yy); Yx = C24 * 2.0f * x; Yy = C24 * -2.0f * y; Yz = 0.0f; break;
    case 9: Y = C30 * y * (3.0f * xx - yy); Yx = C30 * 6.0f * xy; Yy = C30 * (3.0f * xx - 3.0f * yy); Yz = 0.0f; break;
    case 10: Y = C31 * xy * z; Yx = C31 * yz; Yy = C31 * xz; Yz = C31 * xy; break;
    case 11: Y = C32 * y * (4.0f * zz - xx - yy); Yx = C32 * -2.0f * xy; Yy = C32 * (4.0f * zz - xx - 3.0f * yy);
             Yz = C32 * 8.0f * yz; break;
    case 12: Y = C33 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); Yx = C33 * -6.0f * xz; Yy = C33 * -6.0f * yz;
             Yz = C33 * (6.0f * zz - 3.0f * xx - 3.0f * yy); break;
    case 13: Y = C34 * x * (4.0f * zz - xx - yy); Yx = C34 * (4.0f * zz - 3.0f * xx - yy); Yy = C34 * -2.0f * xy;
             Yz = C34 * 8.0f * xz; break;
    case 14: Y = C35 * z * (xx - yy); Yx = C35 * 2.0f * xz; Yy = C35 * -2.0f * yz; Yz = C35 * (xx - yy); break;
    case 15: Y = C36 * x * (xx - 3.0f * yy); Yx = C36 * (3.0f * xx - 3.0f * yy); Yy = C36 * -6.0f * xy; Yz = 0.0f; break;
    default: {        // degree 4 (utils/sh_utils.py:97-110): the polynomials and partial derivatives of sh_backward<4>
        const float C40 = 2.5033429417967046f, C41 = -1.7701307697799304f, C42 = 0.9461746957575601f,
                    C43 = -0.6690465435572892f, C44 = 0.10578554691520431f, C45 = -0.6690465435572892f,
                    C46 = 0.47308734787878004f, C47 = -1.7701307697799304f, C48 = 0.6258357354491761f;
        switch (k) {
        case 16: Y = C40 * xy * (xx - yy); Yx = C40 * y * (3.0f * xx - yy); Yy = C40 * x * (xx - 3.0f * yy); Yz = 0.0f; break;
        case 17: Y = C41 * yz * (3.0f * xx - yy); Yx = C41 * 6.0f * xy * z; Yy = C41 * z * (3.0f * xx - 3.0f * yy);
                 Yz = C41 * y * (3.0f * xx - yy); break;
        case 18: Y = C42 * xy * (7.0f * zz - 1.0f); Yx = C42 * y * (7.0f * zz - 1.0f); Yy = C42 * x * (7.0f * zz - 1.0f);
                 Yz = C42 * 14.0f * xy * z; break;
        case 19: Y = C43 * yz * (7.0f * zz - 3.0f); Yx = 0.0f; Yy = C43 * z * (7.0f * zz - 3.0f);
                 Yz = C43 * y * (21.0f * zz - 3.0f); break;
        case 20: Y = C44 * (zz * (35.0f * zz - 30.0f) + 3.0f); Yx = 0.0f; Yy = 0.0f; Yz = C44 * z * (140.0f * zz - 60.0f); break;
        case 21: Y = C45 * xz * (7.0f * zz - 3.0f); Yx = C45 * z * (7.0f * zz - 3.0f); Yy = 0.0f;
                 Yz = C45 * x * (21.0f * zz - 3.0f); break;
        case 22: Y = C46 * (xx - yy) * (7.0f * zz - 1.0f); Yx = C46 * 2.0f * x * (7.0f * zz - 1.0f);
                 Yy = C46 * -2.0f * y * (7.0f * zz - 1.0f); Yz = C46 * 14.0f * z * (xx - yy); break;
        case 23: Y = C47 * xz * (xx - 3.0f * yy); Yx = C47 * z * (3.0f * xx - 3.0f * yy); Yy = C47 * -6.0f * xy * z;
                 Yz = C47 * x * (xx - 3.0f * yy); break;
        default: Y = C48 * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)); Yx = C48 * x * (4.0f * xx - 12.0f * yy);
                 Yy = C48 * y * (4.0f * yy - 12.0f * xx); Yz = 0.0f; break;
        }
    } break;
    }
}

// DEFER: E3_FLAG_DEFER_SH_MEAN without dL_dsh -- no SH part at all (fewer registers, no LDS slice)
#ifndef E3_GEOM_DEFER_OCC
#define E3_GEOM_DEFER_OCC 4
#endif
// NK: coefficients per channel the SH loop is unrolled for -- 16 (degrees 0..3, the reference model's; the instantiation
// every training path runs) or 25 (degree 4, utils/sh_utils.py:97-110: a caller of the multi-view entry points with
// (P, 25, 3) coefficients)
#ifndef E3_GEOM_PREFETCH
#define E3_GEOM_PREFETCH 1
#endif
template <bool DEFER, int NK = 16>
__global__ __launch_bounds__(256, DEFER ? E3_GEOM_DEFER_OCC : 4) void geom_bwd_multi_kernel(
    int P, int D, int M, const float* __restrict__ means, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ opac_in,
    MultiViews mv, int flags, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmean3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float* __restrict__ dL_drot, float* __restrict__ gcol /* (n, P, 3) or null: per-view clamp-masked dL/dcolour */) {
    // per-view (unit direction, 1/len, masked colour gradient) of this thread's Gaussian.  The view loop of the
    // geometry part is NOT unrolled (one copy of a ~90-register body); its per-view results go through this
    // LDS slice so that the SH part can hold them in statically indexed registers.
    __shared__ float sV[DEFER ? 1 : 7][DEFER ? 1 : E3_MAX_VIEWS][DEFER ? 1 : 256];
    const int tid = threadIdx.x;
    int i = blockIdx.x * blockDim.x + tid;
    if (i >= P) return;
    const bool preact = (flags & E3_FLAG_PREACT) != 0;
    const bool pl = (flags & E3_FLAG_SH_PLANAR) != 0;
    const float* sh = pl ? shs + i : shs + (size_t)i * M * 3;
    float* dsh = !dL_dsh ? nullptr : (pl ? dL_dsh + i : dL_dsh + (size_t)i * M * 3);    // null: SH gradient rebuilt from
    const size_t st = pl ? (size_t)P : (size_t)1;                                        // gcol (sh_grad_views_kernel)
    uint32_t vis = 0;
    const int nv = mv.vs.n;
#pragma unroll
    for (int v = 0; v < E3_MAX_VIEWS; ++v)
        if (v < nv && mv.radii[(size_t)v * P + i] > 0) vis |= 1u << v;
    // A Gaussian no view saw gets zeros -- but NOT on a path of its own: a wave mixes seen and unseen Gaussians, and two
    // paths storing to the same cache lines at different times reach memory as partial-sector writes (WRITE_SIZE of this
    // kernel: 434 MB for 80 MB of gradients, and it runs at the bandwidth of that traffic).  Unseen lanes therefore walk
    // the common path with zero sums and every store below is executed by the whole wave: each store instruction covers
    // whole, contiguous lines.  Only a wave without any seen Gaussian takes the shortcut (its stores are whole lines too).
    const bool seen = vis != 0u;
    // E3_FLAG_MEAN2D_VIEWS: dL_dmean2D is (nviews, P, 3) and receives the screen-space gradient of EVERY view (the
    // deferred renders of adopt.render: each render() call owns a viewspace_points leaf); otherwise view 0's only
    const bool m2views = dL_dmean2D && (flags & E3_FLAG_MEAN2D_VIEWS) != 0;
    if (__builtin_amdgcn_ballot_w64(seen) == 0ull) {
        if (dL_dmean2D) { dL_dmean2D[3 * (size_t)i] = 0.0f; dL_dmean2D[3 * (size_t)i + 1] = 0.0f; dL_dmean2D[3 * (size_t)i + 2] = 0.0f; }
        if (m2views)
            for (int v = 1; v < nv; ++v) {
                float* mp = dL_dmean2D + ((size_t)v * P + i) * 3;
                mp[0] = 0.0f; mp[1] = 0.0f; mp[2] = 0.0f;
            }
        dL_dopacity[i] = 0.0f;
        dL_dmean3D[3 * (size_t)i] = 0.0f; dL_dmean3D[3 * (size_t)i + 1] = 0.0f; dL_dmean3D[3 * (size_t)i + 2] = 0.0f;
        if (dsh) for (int k = 0; k < 3 * M; ++k) dsh[(size_t)k * st] = 0.0f;
        if (gcol)
            for (int v = 0; v < nv; ++v) {
                float* gp = gcol + ((size_t)v * P + i) * 3;
                gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
            }
        dL_dscale[3 * (size_t)i] = 0.0f; dL_dscale[3 * (size_t)i + 1] = 0.0f; dL_dscale[3 * (size_t)i + 2] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dL_drot[4 * (size_t)i + k] = 0.0f;
        return;
    }
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float sact[3], qn[4], qinv;
    act_load_scale_rot(scales + 3 * (size_t)i, rots + 4 * (size_t)i, preact, sact, qn, qinv);
    Cov3 cv;
    build_cov3(sact, mv.vs.v[0].scale_modifier, qn[0], qn[1], qn[2], qn[3], cv);
    float gcov[6] = {0, 0, 0, 0, 0, 0}, gmean[3] = {0, 0, 0}, gopac = 0.0f;
    float m2x = 0.0f, m2y = 0.0f;
#if E3_GEOM_PREFETCH
    // the NEXT view's sums (and clamp bits) are requested before this view is computed: the view loop is rolled (one copy of
    // a ~90-register body), so without this each view's round trip starts only when the previous view's arithmetic is done
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
    uint32_t ncl = 0u;
    auto fetch = [&](int v) __attribute__((always_inline)) {
        if (v < nv && ((vis >> v) & 1u)) {
            const size_t q = (size_t)i * nv + v;
            n0 = mv.gsum[3 * q]; n1 = mv.gsum[3 * q + 1]; n2 = mv.gsum[3 * q + 2];
            ncl = mv.clamped[q];
        }
    };
    fetch(0);
#endif
#pragma unroll 1
    for (int v = 0; v < nv; ++v) {
        float o_dx = 0.0f, o_dy = 0.0f, o_dz = 0.0f, o_il = 0.0f, o_g0 = 0.0f, o_g1 = 0.0f, o_g2 = 0.0f;
        float o_m2x = 0.0f, o_m2y = 0.0f;
#if E3_GEOM_PREFETCH
        const float4 c0 = n0, c1 = n1, c2 = n2;
        const uint32_t ccl = ncl;
        fetch(v + 1);
#endif
        if ((vis >> v) & 1u) {
            const ViewParams& vp = mv.vs.v[v];
            const size_t q = (size_t)i * nv + v;
            float g12[9], gcv[6], gmv[3];
#if E3_GEOM_PREFETCH
            g12[0] = c0.x; g12[1] = c0.y; g12[2] = c0.z; g12[3] = c0.w;
            g12[4] = c1.x; g12[5] = c1.y; g12[6] = c1.z; g12[7] = c1.w; g12[8] = c2.x;
#else
            load_sums(mv.gsum, q, g12);
#endif
            o_m2x = g12[0]; o_m2y = g12[1];          // NDC-unit screen-space gradient of this view (E3_FLAG_MEAN2D_VIEWS)
            view_geom_backward(vp, mx, my, mz, cv.S, g12, gcv, gmv);
#pragma unroll
            for (int k = 0; k < 6; ++k) gcov[k] += gcv[k];
            gmean[0] += gmv[0]; gmean[1] += gmv[1]; gmean[2] += gmv[2];
            gopac += g12[5];
            if (v == 0) {
#if E3_GEOM_PREFETCH
                if (mv.stats) { m2x = c2.y; m2y = c2.z; }
#else
                if (mv.stats) { const float4 s2 = mv.gsum[3 * q + 2]; m2x = s2.y; m2y = s2.z; }
#endif
                else { m2x = g12[0]; m2y = g12[1]; }
            }
#if E3_GEOM_PREFETCH
            const uint32_t cl = ccl;
            (void)q;
#else
            const uint32_t cl = mv.clamped[q];
#endif
            o_g0 = (cl & 1u) ? 0.0f : g12[6];
            o_g1 = (cl & 2u) ? 0.0f : g12[7];
            o_g2 = (cl & 4u) ? 0.0f : g12[8];
            if (!DEFER) {
                const float ox = mx - vp.campos[0], oy = my - vp.campos[1], oz = mz - vp.campos[2];
                const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
                o_dx = ox / len; o_dy = oy / len; o_dz = oz / len;
                o_il = 1.0f / len;
            }
        }
        if (!DEFER) {
            sV[0][v][tid] = o_dx; sV[1][v][tid] = o_dy; sV[2][v][tid] = o_dz; sV[3][v][tid] = o_il;
            sV[4][v][tid] = o_g0; sV[5][v][tid] = o_g1; sV[6][v][tid] = o_g2;
        }
        if (gcol) {
            float* gp = gcol + ((size_t)v * P + i) * 3;
            gp[0] = o_g0; gp[1] = o_g1; gp[2] = o_g2;
        }
        if (m2views && v > 0) {      // (view 0's block is written below; unseen views: zeros, o_m2 = 0)
            float* mp = dL_dmean2D + ((size_t)v * P + i) * 3;
            mp[0] = o_m2x; mp[1] = o_m2y; mp[2] = 0.0f;
        }
    }
    if (dL_dmean2D) {   // densification statistics use render #1 only (train.py:145)
        dL_dmean2D[3 * (size_t)i] = m2x; dL_dmean2D[3 * (size_t)i + 1] = m2y; dL_dmean2D[3 * (size_t)i + 2] = 0.0f;
    }
    if (preact) { float o = act_sigmoid(opac_in[i]); gopac = gopac * o * (1.0f - o); }
    dL_dopacity[i] = seen ? gopac : 0.0f;
    {
        float ds[3], dq[4];
        cov3_backward(cv, gcov, mv.vs.v[0].scale_modifier, qn[0], qn[1], qn[2], qn[3], preact, sact, qinv, ds, dq);
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dscale[3 * (size_t)i + k] = seen ? ds[k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dL_drot[4 * (size_t)i + k] = seen ? dq[k] : 0.0f;
    }
    // E3_FLAG_DEFER_SH_MEAN (only without dL_dsh): the part of dL/dmean that comes through the view directions of the SH
    // colours needs all 48 coefficients -- a third of this kernel's fetches -- and sh_adam_views_kernel streams over
    // exactly those coefficients right afterwards with the same directions and colour gradients in registers: it adds
    // the term there (same operations in the same order: bit-identical), this kernel stores the geometric part only.
    if (DEFER) {
        dL_dmean3D[3 * (size_t)i] = seen ? gmean[0] : 0.0f; dL_dmean3D[3 * (size_t)i + 1] = seen ? gmean[1] : 0.0f;
        dL_dmean3D[3 * (size_t)i + 2] = seen ? gmean[2] : 0.0f;
        return;
    }
    // ---- SH coefficients: each written once; direction gradients collected per view
    float dx[E3_MAX_VIEWS], dy[E3_MAX_VIEWS], dz[E3_MAX_VIEWS], gc[E3_MAX_VIEWS][3];
    float ddx[E3_MAX_VIEWS], ddy[E3_MAX_VIEWS], ddz[E3_MAX_VIEWS];
#pragma unroll
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        ddx[v] = ddy[v] = ddz[v] = 0.0f;
        dx[v] = dy[v] = dz[v] = gc[v][0] = gc[v][1] = gc[v][2] = 0.0f;
        if (v < nv) {
            dx[v] = sV[0][v][tid]; dy[v] = sV[1][v][tid]; dz[v] = sV[2][v][tid];
            gc[v][0] = sV[4][v][tid]; gc[v][1] = sV[5][v][tid]; gc[v][2] = sV[6][v][tid];
        }
    }
    const int nk = (D + 1) * (D + 1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k < nk) {
            // (unseen Gaussians do not read their coefficients: exec-masked loads)
            const float c0 = seen ? sh[(size_t)(3 * k) * st] : 0.0f, c1 = seen ? sh[(size_t)(3 * k + 1) * st] : 0.0f,
                        c2 = seen ? sh[(size_t)(3 * k + 2) * st] : 0.0f;
            float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
            for (int v = 0; v < E3_MAX_VIEWS; ++v) {
                if (v < nv) {
                    float Y, Yx, Yy, Yz;
                    sh_basis(k, dx[v], dy[v], dz[v], Y, Yx, Yy, Yz);
                    o0 = FMA(Y, gc[v][0], o0); o1 = FMA(Y, gc[v][1], o1); o2 = FMA(Y, gc[v][2], o2);
                    const float sgn = FMA(c0, gc[v][0], FMA(c1, gc[v][1], c2 * gc[v][2]));
                    ddx[v] = FMA(Yx, sgn, ddx[v]); ddy[v] = FMA(Yy, sgn, ddy[v]); ddz[v] = FMA(Yz, sgn, ddz[v]);
                }
            }
            if (dsh) { dsh[(size_t)(3 * k) * st] = o0; dsh[(size_t)(3 * k + 1) * st] = o1; dsh[(size_t)(3 * k + 2) * st] = o2; }
        }
    }
    if (dsh) for (int k = 3 * nk; k < 3 * M; ++k) dsh[(size_t)k * st] = 0.0f;
#pragma unroll
    for (int v = 0; v < E3_MAX_VIEWS; ++v) {
        if (v < nv) {
            const float il = sV[3][v][tid];
            const float dot = dx[v] * ddx[v] + dy[v] * ddy[v] + dz[v] * ddz[v];
            gmean[0] += (ddx[v] - dx[v] * dot) * il;
            gmean[1] += (ddy[v] - dy[v] * dot) * il;
            gmean[2] += (ddz[v] - dz[v] * dot) * il;
        }
    }
    dL_dmean3D[3 * (size_t)i] = seen ? gmean[0] : 0.0f; dL_dmean3D[3 * (size_t)i + 1] = seen ? gmean[1] : 0.0f;
    dL_dmean3D[3 * (size_t)i + 2] = seen ? gmean[2] : 0.0f;
}

// ---- SH gradient from per-view colour gradients -------------------------------------------------------------
// dL/dsh[k][ch] = sum over views of Y_k(dir_view) * dL/dcolour_view[ch]: a rank-<=3 structure per view.  Under view-
// parallel data parallelism the ranks therefore do not have to average the 48 SH-gradient floats per Gaussian
// (81 % of the gradient bytes): they exchange the 3 clamp-masked colour-gradient floats per (Gaussian, view) plus
// the camera centres -- 9 instead of 48 floats per Gaussian and rank for an event iteration -- and every rank
// rebuilds the averaged SH gradient itself.  `packed` holds one block per rank:
// [views_per_rank x P x 3 colour gradients | views_per_rank x 3 camera centres], `rank_stride` floats apart.
__global__ __launch_bounds__(256) void sh_grad_views_kernel(int P, int nranks, int views_per_rank, int D, int M,
                                                            const float* __restrict__ means,
                                                            const float* __restrict__ packed, size_t rank_stride,
                                                            float scale, float* __restrict__ dL_dsh, int planar) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float acc[16][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
    const int nk = (D + 1) * (D + 1);
#pragma unroll 1
    for (int r = 0; r < nranks; ++r) {
        const float* blk = packed + (size_t)r * rank_stride;
        const float* cams = blk + (size_t)views_per_rank * P * 3;
#pragma unroll 1
        for (int v = 0; v < views_per_rank; ++v) {
            const float* gp = blk + ((size_t)v * P + i) * 3;
            const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
            if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;          // culled / clamped in this view
            const float ox = mx - cams[3 * v], oy = my - cams[3 * v + 1], oz = mz - cams[3 * v + 2];
            const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox / len, y = oy / len, z = oz / len;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < nk) {
                    float Y, Yx, Yy, Yz;
                    sh_basis(k, x, y, z, Y, Yx, Yy, Yz);
                    acc[k][0] = FMA(Y, g0, acc[k][0]); acc[k][1] = FMA(Y, g1, acc[k][1]); acc[k][2] = FMA(Y, g2, acc[k][2]);
                }
            }
        }
    }
    float* dsh = planar ? dL_dsh + i : dL_dsh + (size_t)i * M * 3;
    const size_t st = planar ? (size_t)P : (size_t)1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < nk) {
            dsh[(size_t)(3 * k) * st] = acc[k][0] * scale; dsh[(size_t)(3 * k + 1) * st] = acc[k][1] * scale;
            dsh[(size_t)(3 * k + 2) * st] = acc[k][2] * scale;
        }
    }
    for (int k = 3 * nk; k < 3 * M; ++k) dsh[(size_t)k * st] = 0.0f;
}

// The same rebuild fused with the optimizer step of the SH coefficients: the gradient never reaches memory -- it goes
// straight into torch.optim.Adam's update (f_dc / f_rest learning rates, scene/gaussian_model.py:156-157).  One
// streaming pass that reads 3 colour-gradient floats per (Gaussian, view) instead of writing and re-reading 48
// gradients.  blockIdx.y selects E3_SH_SLICE coefficients (2: 6 elements): a workgroup streams 18 planes of the
// coefficient-major arrays (all 144 at once ran at 3.9 TB/s: too many concurrent DRAM streams), the moments and
// parameters are requested before the gradient is computed, and the colour gradients re-read by the y-slices come
// from L2 / Infinity Cache.  Same arithmetic as sh_grad_views_kernel + adam_segments_kernel: bit-identical results.
#ifndef E3_SH_SLICE
#define E3_SH_SLICE 2      // SH coefficients (x 3 channels) per workgroup slice: 18 streams per thread (1 / 2 / 4 / 8: 0.329 / 0.303 / 0.318 / 0.388 ms for the optimizer stage)
#endif
struct ShAdam { float* m; float* v; float ss_dc, ss_rest, bc2s, b1, b2, eps, omb1, omb2 /* 1 - beta as torch's fp32 (common.h) */; };
__global__ __launch_bounds__(256) void sh_adam_views_sliced_kernel(int P, int nranks, int views_per_rank, int D, int M,
                                                            const float* __restrict__ means,
                                                            const float* __restrict__ packed, size_t rank_stride,
                                                            float scale, float* __restrict__ sh, int planar, ShAdam ad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    constexpr int CS = E3_SH_SLICE, NE = 3 * CS;
    const int k0 = CS * (int)blockIdx.y;                 // this slice: coefficients k0 .. k0 + CS - 1
    const size_t st = planar ? (size_t)P : (size_t)1;
    const size_t e0 = (planar ? (size_t)i : (size_t)i * M * 3) + (size_t)(3 * k0) * st;
    float m0[NE], v0[NE], p0[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { m0[j] = ad.m[e0 + j * st]; v0[j] = ad.v[e0 + j * st]; p0[j] = sh[e0 + j * st]; }
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float acc[CS][3];
#pragma unroll
    for (int k = 0; k < CS; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
    const int nk = (D + 1) * (D + 1);
#pragma unroll 1
    for (int r = 0; r < nranks; ++r) {
        const float* blk = packed + (size_t)r * rank_stride;
        const float* cams = blk + (size_t)views_per_rank * P * 3;
#pragma unroll 1
        for (int v = 0; v < views_per_rank; ++v) {
            const float* gp = blk + ((size_t)v * P + i) * 3;
            const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
            if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;          // culled / clamped in this view
            const float ox = mx - cams[3 * v], oy = my - cams[3 * v + 1], oz = mz - cams[3 * v + 2];
            const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox / len, y = oy / len, z = oz / len;
#pragma unroll
            for (int kk = 0; kk < CS; ++kk) {
                if (k0 + kk < nk) {                                         // (uniform per workgroup)
                    float Y, Yx, Yy, Yz;
                    sh_basis(k0 + kk, x, y, z, Y, Yx, Yy, Yz);
                    acc[kk][0] = FMA(Y, g0, acc[kk][0]); acc[kk][1] = FMA(Y, g1, acc[kk][1]); acc[kk][2] = FMA(Y, g2, acc[kk][2]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const float g = (k0 + j / 3 < nk) ? acc[j / 3][j % 3] * scale : 0.0f;     // inactive degrees: zero gradient, moments decay
        const float mi = m0[j] + ad.omb1 * (g - m0[j]);
        const float vi = v0[j] * ad.b2 + ad.omb2 * g * g;
        ad.m[e0 + j * st] = mi; ad.v[e0 + j * st] = vi;
        sh[e0 + j * st] = p0[j] - ((k0 == 0 && j < 3) ? ad.ss_dc : ad.ss_rest) * (mi / (__builtin_sqrtf(vi) / ad.bc2s + ad.eps));
    }
}


// Up to E3_SH_REG_VIEWS views in all (one rank's triplet): the per-view unit directions and colour gradients of the
// thread's Gaussian are computed ONCE and kept in registers, and the thread walks the M / E3_SH_SLICE coefficient slices
// itself -- the sliced kernel above re-reads the colour gradients and means once per slice (8 x 48 MB of its 960 MB of
// fetches at 1 M Gaussians).  Each slice is the same 18 streams per thread; same arithmetic per element: bit-identical.
constexpr int E3_SH_REG_VIEWS = 4;
#ifndef E3_SH_MEAN_OCC
#define E3_SH_MEAN_OCC 2      // workgroups per CU of the MEAN variant: 3 = 168 VGPRs + 80 B of scratch (optimizer stage +28 us), 2 = no scratch (+8 us)
#endif
template <bool MEAN, int NV>
__global__ __launch_bounds__(256, MEAN ? E3_SH_MEAN_OCC : 3) void sh_adam_views_kernel(int P, int nranks, int views_per_rank, int D, int M,
                                                            const float* __restrict__ means,
                                                            const float* __restrict__ packed, size_t rank_stride,
                                                            float scale, float* __restrict__ sh, int planar, ShAdam ad,
                                                            float* __restrict__ dmean /* or null: E3_FLAG_DEFER_SH_MEAN */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    constexpr int CS = E3_SH_SLICE, NE = 3 * CS;
    const size_t st = planar ? (size_t)P : (size_t)1;
    const size_t ebase = planar ? (size_t)i : (size_t)i * M * 3;
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float vx[NV], vy[NV], vz[NV], vg[NV][3];
    float vil[NV], ddx[NV], ddy[NV], ddz[NV];       // dmean: 1 / |mean - camera|, dL/d(unit direction) (geom_bwd_multi_kernel's SH part)
    bool von[NV];
    const int nvt = nranks * views_per_rank;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        von[t] = false; vx[t] = vy[t] = vz[t] = vg[t][0] = vg[t][1] = vg[t][2] = 0.0f;
        vil[t] = ddx[t] = ddy[t] = ddz[t] = 0.0f;
        if (t < nvt) {
            const int r = t / views_per_rank, v = t - r * views_per_rank;
            const float* blk = packed + (size_t)r * rank_stride;
            const float* cams = blk + (size_t)views_per_rank * P * 3;
            const float* gp = blk + ((size_t)v * P + i) * 3;
            vg[t][0] = gp[0]; vg[t][1] = gp[1]; vg[t][2] = gp[2];
            von[t] = !(vg[t][0] == 0.0f && vg[t][1] == 0.0f && vg[t][2] == 0.0f);   // culled / clamped in this view
            const float ox = mx - cams[3 * v], oy = my - cams[3 * v + 1], oz = mz - cams[3 * v + 2];
            const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
            vx[t] = ox / len; vy[t] = oy / len; vz[t] = oz / len;
            vil[t] = 1.0f / len;
        }
    }
    const int nk = (D + 1) * (D + 1);
    const int nslices = M / CS;
    constexpr int SLICE_UNROLL = MEAN ? 1 : 2;
#pragma unroll SLICE_UNROLL
    for (int sl = 0; sl < nslices; ++sl) {
        const int k0 = CS * sl;
        const size_t e0 = ebase + (size_t)(3 * k0) * st;
        float m0[NE], v0[NE], p0[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) { m0[j] = ad.m[e0 + j * st]; v0[j] = ad.v[e0 + j * st]; p0[j] = sh[e0 + j * st]; }
        float acc[CS][3];
#pragma unroll
        for (int k = 0; k < CS; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (von[t]) {
#pragma unroll
                for (int kk = 0; kk < CS; ++kk) {
                    if (k0 + kk < nk) {
                        float Y, Yx, Yy, Yz;
                        sh_basis(k0 + kk, vx[t], vy[t], vz[t], Y, Yx, Yy, Yz);
                        acc[kk][0] = FMA(Y, vg[t][0], acc[kk][0]); acc[kk][1] = FMA(Y, vg[t][1], acc[kk][1]);
                        acc[kk][2] = FMA(Y, vg[t][2], acc[kk][2]);
                        if (MEAN) {              // the coefficients as the backward saw them: p0, not yet stepped
                            const float sgn = FMA(p0[3 * kk], vg[t][0], FMA(p0[3 * kk + 1], vg[t][1], p0[3 * kk + 2] * vg[t][2]));
                            ddx[t] = FMA(Yx, sgn, ddx[t]); ddy[t] = FMA(Yy, sgn, ddy[t]); ddz[t] = FMA(Yz, sgn, ddz[t]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const float g = (k0 + j / 3 < nk) ? acc[j / 3][j % 3] * scale : 0.0f;     // inactive degrees: zero gradient, moments decay
            const float mi = m0[j] + ad.omb1 * (g - m0[j]);
            const float vi = v0[j] * ad.b2 + ad.omb2 * g * g;
            ad.m[e0 + j * st] = mi; ad.v[e0 + j * st] = vi;
            sh[e0 + j * st] = p0[j] - ((k0 == 0 && j < 3) ? ad.ss_dc : ad.ss_rest) * (mi / (__builtin_sqrtf(vi) / ad.bc2s + ad.eps));
        }
    }
    if (MEAN) {
        // direction gradient -> position, view by view on top of the geometric part (the order of geom_bwd_multi_kernel)
        float g0 = dmean[3 * (size_t)i], g1 = dmean[3 * (size_t)i + 1], g2 = dmean[3 * (size_t)i + 2];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (von[t]) {
                const float dot = vx[t] * ddx[t] + vy[t] * ddy[t] + vz[t] * ddz[t];
                g0 += (ddx[t] - vx[t] * dot) * vil[t];
                g1 += (ddy[t] - vy[t] * dot) * vil[t];
                g2 += (ddz[t] - vz[t] * dot) * vil[t];
            }
        }
        dmean[3 * (size_t)i] = g0; dmean[3 * (size_t)i + 1] = g1; dmean[3 * (size_t)i + 2] = g2;
    }
}

int e3_sh_adam_views_impl(int P, int nranks, int views_per_rank, int D, int M, const float* means3D, const float* packed,
                          size_t rank_stride, float scale, float* sh, float* exp_avg, float* exp_avg_sq, float lr_dc,
                          float lr_rest, float b1, float b2, float eps, int step, int flags, hipStream_t s, float* dmean) {
    if (P <= 0) return 0;
    const double bc1 = 1.0 - pow(e3_beta_double(b1), step), bc2 = 1.0 - pow(e3_beta_double(b2), step);
    ShAdam ad;
    ad.omb1 = e3_one_minus_beta(b1); ad.omb2 = e3_one_minus_beta(b2);
    ad.m = exp_avg; ad.v = exp_avg_sq; ad.ss_dc = (float)((double)lr_dc / bc1); ad.ss_rest = (float)((double)lr_rest / bc1);
    ad.bc2s = (float)sqrt(bc2); ad.b1 = b1; ad.b2 = b2; ad.eps = eps;
    static const bool reg_views = !(getenv("E3DGS_SH_ADAM_SLICED") && atoi(getenv("E3DGS_SH_ADAM_SLICED")) != 0);   // (A/B switch)
    if (dmean && !(nranks * views_per_rank <= E3_SH_REG_VIEWS && M % E3_SH_SLICE == 0)) return (int)hipErrorInvalidValue;
    if ((reg_views || dmean) && nranks * views_per_rank <= E3_SH_REG_VIEWS && M % E3_SH_SLICE == 0)
        (dmean ? (nranks * views_per_rank <= 3 ? sh_adam_views_kernel<true, 3> : sh_adam_views_kernel<true, E3_SH_REG_VIEWS>)
               : sh_adam_views_kernel<false, E3_SH_REG_VIEWS>)<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, nranks, views_per_rank, D, M, means3D, packed, rank_stride, scale, sh, (flags & E3_FLAG_SH_PLANAR) != 0, ad, dmean);
    else
        sh_adam_views_sliced_kernel<<<dim3((P + 255) / 256, M / E3_SH_SLICE), dim3(256), 0, s>>>(
            P, nranks, views_per_rank, D, M, means3D, packed, rank_stride, scale, sh, (flags & E3_FLAG_SH_PLANAR) != 0, ad);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int e3_sh_grad_views_impl(int P, int nranks, int views_per_rank, int D, int M, const float* means3D, const float* packed,
                          size_t rank_stride, float scale, float* dL_dsh, int flags, hipStream_t s) {
    if (P <= 0) return 0;
    sh_grad_views_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, nranks, views_per_rank, D, M, means3D, packed,
                                                                     rank_stride, scale, dL_dsh,
                                                                     (flags & E3_FLAG_SH_PLANAR) != 0);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------------------------ host driver
int e3_fail(hipError_t e, const char* what);
#define KERNEL_OK(name)                                       \
    do {                                                      \
        hipError_t _e = hipGetLastError();                    \
        if (_e != hipSuccess) return e3_fail(_e, name);       \
        if (debug) {                                          \
            _e = hipStreamSynchronize(s);                     \
            if (_e != hipSuccess) return e3_fail(_e, name);   \
        }                                                     \
    } while (0)

int e3_backward_impl(const ViewBatch& views, int P, int D, int M, int num_rendered, const float* background, int W,
                     int H, const float* means3D, const float* shs, const float* colors, const float* opacities,
                     const float* scales, float scale_modifier, const float* rots, const float* cov_pre,
                     const int* radii, const char* geom_buffer, const char* binning_buffer,
                     const char* image_buffer, const float* dL_dpix, float* grad_acc, float* dL_dmean2D,
                     float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                     float* dL_dscale, float* dL_drot, int debug, int flags, hipStream_t s, float* dL_dcolour_views,
                     const float* dL_dpix_stats, const Rank1Views* rank1) {
    (void)colors;
    if (P <= 0) return 0;
    const ViewSet vs = make_view_set(views, W, H, scale_modifier);
    const int nv = vs.n;
    const int gx = vs.v[0].gx;
    const int tiles_per_view = gx * vs.v[0].gy;
    const int ntiles = tiles_per_view * nv;
    const size_t Q = (size_t)P * nv;
    char* gp = const_cast<char*>(geom_buffer);
    char* bp = const_cast<char*>(binning_buffer);
    if (const int rc = e3_geom_opts_check(gp, Q, e3_call_opts(flags), "backward")) return rc;
    char* ip = const_cast<char*>(image_buffer);
    GeomState geom = GeomState::from(gp, Q);
    BinningState bin = BinningState::from(bp, (size_t)num_rendered);
    ImageState img = ImageState::from(ip, (size_t)W * H * nv, ntiles);
    if (num_rendered > 0 && !(flags & E3_FLAG_BWD_ONLY_GEOM)) {
        ProfScope ps(PS_RENDER_BWD, s);
        // launch order by the cost the forward measured per tile: per-class lists the forward kernel filled, or the
        // ordering kernel (e3_use_lpt_lists: the same answer as in the forward)
        const bool use_lpt = e3_use_lpt_lists(ntiles, num_rendered, P);
        const int nslots = use_lpt ? ntiles : launch_tile_order(ntiles, tiles_per_view, gx, img.ranges, img.work, img.order_bwd, s);
        const uint32_t* ord = use_lpt ? nullptr : img.order_bwd;
        const uint32_t* lc = img.lpt_cnt + E3_LPT_CLASSES;
        const uint32_t* ll = img.lpt_list + (size_t)E3_LPT_CLASSES * ntiles;
        // (STATS: the second chain's records sit in the slack between the packed 9-float records and the per-splat sums:
        // the caller provides E3_ACC_STRIDE = 12 floats per instance)
        float* part2 = grad_acc + E3_REC_FLOATS * (size_t)num_rendered;
        const bool fast = e3_call_opts(flags).fast_exp != 0;
        Rank1Views r1;
        memset(&r1, 0, sizeof r1);
        if (rank1) r1 = *rank1;
        r1.mask &= (1u << nv) - 1u;
        if (dL_dpix_stats)
            (fast ? render_bwd_stats_fast_kernel : render_bwd_stats_kernel)<<<dim3((nslots + BWD_WAVES - 1) / BWD_WAVES), dim3(BWD_WAVES * WAVE), 0, s>>>(
                g_trace, nslots, tiles_per_view, ord, gx, W, H, img.ranges, bin.emit_gid, geom.rec,
                background, img.final_T, img.n_contrib, bin.perm, bin.strip_mask, dL_dpix, grad_acc, lc, ll, (uint32_t)ntiles,
                r1, dL_dpix_stats, part2);
        else
            (fast ? render_bwd_fast_kernel : render_bwd_kernel)<<<dim3((nslots + BWD_WAVES - 1) / BWD_WAVES), dim3(BWD_WAVES * WAVE), 0, s>>>(
                g_trace, nslots, tiles_per_view, ord, gx, W, H, img.ranges, bin.emit_gid, geom.rec,
                background, img.final_T, img.n_contrib, bin.perm, bin.strip_mask, dL_dpix, grad_acc, lc, ll, (uint32_t)ntiles, r1);
    }
    KERNEL_OK("render_bwd_kernel");
    if (flags & E3_FLAG_BWD_ONLY_RENDER) return 0;
    {
    ProfScope ps(PS_GEOM_BWD, s);
    // per-splat sums live behind the instance records in the caller's scratch: grad_acc is (num_rendered + Q, 12)
    float4* gsum = reinterpret_cast<float4*>(grad_acc + E3_ACC_STRIDE * (size_t)num_rendered);
    // E3_FLAG_COUNT_DEVICE: num_rendered is the capacity of a forward_multi_capacity call; the count itself is the last
    // element of the forward's scan of the per-wave instance counts
    const uint32_t* count_dev = nullptr;
    if (flags & E3_FLAG_COUNT_DEVICE) {
        const int gshift = e3_bin_group_shift(Q, e3_call_opts(flags).small_paths);
        count_dev = geom.offsets + (unsigned)((Q + ((size_t)1 << gshift) - 1) >> gshift);
    }
    if (num_rendered > 0 && Q <= E3_RUN_REDUCE_WAVE_MAX && e3_call_opts(flags).small_paths)
        run_reduce_wave_kernel<<<dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, s>>>((uint32_t)Q, geom.ord0, geom.run,
                                                                                    grad_acc, gsum, bin.touched, geom.nvis,
                                                                                    count_dev, (uint32_t)num_rendered);
    else if (num_rendered > 0)
        run_reduce_kernel<<<dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s>>>((uint32_t)Q, geom.ord0, geom.run,
                                                                                   grad_acc, gsum, geom.nvis, bin.touched,
                                                                                   count_dev, (uint32_t)num_rendered);
    else        // no instance at all (a radius can still be > 0 when every tile of the splat was culled): zero sums
    {
        hipError_t me = hipMemsetAsync(gsum, 0, Q * 3 * sizeof(float4), s);
        if (me != hipSuccess) return e3_fail(me, "hipMemsetAsync(gsum)");
    }
    if (dL_dpix_stats && num_rendered > 0)
        run_reduce_stats_kernel<<<dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s>>>(
            geom.ord0, geom.run, grad_acc + E3_REC_FLOATS * (size_t)num_rendered, gsum, geom.nvis, bin.touched, (uint32_t)nv,
            count_dev, (uint32_t)num_rendered);
    if (nv == 1 && !dL_dcolour_views) {
        // instantiation: accumulate or overwrite x SH degree x (P, M, 3) rows movable as float4 (16-byte aligned rows of the
        // reference layout, overwrite mode)
        const bool acc = (flags & E3_FLAG_ACCUMULATE) != 0;
        const bool vec4 = shs && dL_dsh && !acc && !(flags & E3_FLAG_SH_PLANAR) &&
                          ((reinterpret_cast<uintptr_t>(shs) | reinterpret_cast<uintptr_t>(dL_dsh) |
                            (uintptr_t)((size_t)M * 12)) & 15) == 0;
        using Kern = void (*)(int, int, const float*, const float*, const float*, const float*, const float*, const float*,
                              ViewParams, int, const int*, const uint32_t*, const float4*, float*, float*, float*, float*,
                              float*, float*, float*, float*);
        static const Kern table[3][5] = {
            {geom_bwd_kernel<false, 0, false>, geom_bwd_kernel<false, 1, false>, geom_bwd_kernel<false, 2, false>,
             geom_bwd_kernel<false, 3, false>, geom_bwd_kernel<false, 4, false>},
            {geom_bwd_kernel<false, 0, true>, geom_bwd_kernel<false, 1, true>, geom_bwd_kernel<false, 2, true>,
             geom_bwd_kernel<false, 3, true>, geom_bwd_kernel<false, 4, true>},
            {geom_bwd_kernel<true, 0, false>, geom_bwd_kernel<true, 1, false>, geom_bwd_kernel<true, 2, false>,
             geom_bwd_kernel<true, 3, false>, geom_bwd_kernel<true, 4, false>}};
        const int dsel = !shs ? 0 : (D < 0 ? 0 : (D > 4 ? 4 : D));
        table[acc ? 2 : (vec4 ? 1 : 0)][dsel]<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, M, means3D, shs, scales, rots, opacities, cov_pre, vs.v[0], flags, radii, geom.clamped, gsum,
            dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    }
    else {
        // several views (or one whose colour gradient is handed out instead of the SH gradient): one pass, every
        // gradient element written once (capi.hip checked the argument subset)
        MultiViews mv;
        mv.vs = vs; mv.radii = radii; mv.clamped = geom.clamped; mv.gsum = gsum;
        mv.stats = (dL_dpix_stats && num_rendered > 0) ? 1 : 0;
        ((flags & E3_FLAG_DEFER_SH_MEAN) != 0 && !dL_dsh && dL_dcolour_views ? geom_bwd_multi_kernel<true>
         : (D > 3 ? geom_bwd_multi_kernel<false, 25> : geom_bwd_multi_kernel<false>))<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, D, M, means3D, shs, scales, rots, opacities, mv, flags, dL_dmean2D, dL_dopacity, dL_dmean3D, dL_dsh,
            dL_dscale, dL_drot, dL_dcolour_views);
    }
    }
    KERNEL_OK("geom_bwd_kernel");
    return 0;
}
