/*
 * gs_oracle.c -- CPU restatement of the tile-based differentiable Gaussian
 * rasteriser that the reference imports as `diff_gaussian_rasterization`
 * (gaussian_renderer/__init__.py:15; settings :38-51; call :89-97).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under event_3dgs_amd/,
 * diff_gaussian_rasterization/ or simple_knn/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED for the rasteriser core: the reference's CUDA source is an
 * empty, un-vendored submodule (.gitmodules:4-6, pin unknown), so this file
 * restates the published 3DGS algorithm (Kerbl et al. 2023, sections 4-6) and
 * SURVEY.md Appendix A/B, anchored on the in-tree evidence:
 *   - ndc2Pix in double:        gaussian_renderer/__init__.py:238-241 (and :193)
 *   - row-vector / GLM layout:  scene/cameras.py:54-57, scene/dataset_readers.py:246
 *   - Sigma3 = R S S^T R^T, 6-float packing, (r,x,y,z) quaternion:
 *                               scene/gaussian_model.py:27-31, utils/general_utils.py:64-110
 *   - SH basis and +0.5/clamp:  utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:81
 *   - (3,H,W) output, int radii, NDC-unit (P,3) screen-space gradient:
 *                               gaussian_renderer/__init__.py:89,103, scene/gaussian_model.py:405-407
 * Those anchors ARE pinned by golden vectors generated from the reference's
 * importable Python (tests/golden/make_golden.py) and by the analytic
 * known-answer tests in tests/test_oracle_known_answers.py.
 *
 * Arithmetic contract.  Every fp32 operation below is written as one explicit
 * statement (explicit fmaf where a fused multiply-add is meant) and the file is
 * compiled with -ffp-contract=off, so that the HIP kernels can reproduce the
 * forward pass bit for bit: same culls, same integer radius/rect/tile keys,
 * same stable depth order, same alpha/transmittance decisions.  exp() is the
 * deterministic exp_det() below (|rel err| < 5e-7 on the live range power >= -5.6) for the same reason.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- named constants of the algorithm ---- */
#define TILE_X 16
#define TILE_Y 16
#define NEAR_CULL_Z 0.2f
#define GUARD_BAND 1.3f
#define DILATION 0.3f
#define EIGEN_FLOOR 0.1f
#define ALPHA_CLAMP 0.99f
#define ALPHA_SKIP (1.0f / 255.0f)
#define T_STOP 0.0001f
#define W_EPS 0.0000001f
#define DET2_EPS 0.0000001f

/* SH constants: utils/sh_utils.py:26-54 rounded to fp32 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

#define FMA(a, b, c) fmaf((a), (b), (c))

/* Deterministic exp for x <= 0: 2^(x*log2e) with a degree-5 polynomial on the
 * fractional part.  Bit-reproducible on any IEEE-754 machine with fmaf. */
static inline float exp_det(float x) {
    float t = fmaxf(x * 1.4426950408889634f, -126.0f);
    float n = rintf(t);
    float f = t - n;
    float p = 0.0013218672247603536f;
    p = FMA(p, f, 0.009671698324382305f);
    p = FMA(p, f, 0.05550893023610115f);
    p = FMA(p, f, 0.24022237956523895f);
    p = FMA(p, f, 0.6931468844413757f);
    p = FMA(p, f, 1.0f);
    return ldexpf(p, (int)n);
}
float gso_exp_det(float x) { return exp_det(x); }

/* Activations of the raw parameters (scene/gaussian_model.py:33-41 setup_functions, :95-118 getters):
 *   scaling = exp(_scaling), opacity = sigmoid(_opacity), rotation = F.normalize(_rotation) (eps 1e-12).
 * The rasteriser's fused entry points take the RAW parameters (E3DGS_FLAG_PREACT) and apply these inside the projection
 * kernel; this function is the oracle's statement of them, operation for operation: exp is exp_det (the same
 * deterministic polynomial as in compositing; |rel err| < 2e-6 against exp() on |x| <= 20), division and sqrt are IEEE,
 * the norm is one explicit fmaf chain.  gso_forward() fed these values must equal the PREACT kernels bit for bit.
 * Any of the three inputs may be NULL. */
void gso_activate(int P, const float *log_scales /*P,3*/, const float *raw_rots /*P,4*/, const float *logit_opac /*P*/,
                  float *scales, float *rots, float *opac) {
    for (int i = 0; i < P; ++i) {
        if (log_scales) {
            for (int k = 0; k < 3; ++k) {       /* NaN in -> NaN out (torch.exp); exp_det's clamp would swallow it */
                const float x = log_scales[3 * i + k];
                scales[3 * i + k] = x != x ? x : exp_det(x);
            }
        }
        if (raw_rots) {
            const float *q = raw_rots + 4 * i;
            float nrm = sqrtf(FMA(q[0], q[0], FMA(q[1], q[1], FMA(q[2], q[2], q[3] * q[3]))));
            float qinv = 1.0f / fmaxf(nrm, 1e-12f);
            for (int k = 0; k < 4; ++k) rots[4 * i + k] = q[k] * qinv;
        }
        if (logit_opac) opac[i] = logit_opac[i] != logit_opac[i] ? logit_opac[i] : 1.0f / (1.0f + exp_det(-logit_opac[i]));
    }
}

/* flat[4*c + r]: row r of the column-vector matrix applied to (x,y,z,1) */
#define XFORM(M, r, x, y, z) FMA((M)[(r)], (x), FMA((M)[4 + (r)], (y), FMA((M)[8 + (r)], (z), (M)[12 + (r)])))

typedef struct {
    int P, D, M, W, H, gx, gy, T;
    float tanfovx, tanfovy, focal_x, focal_y, mod;
    float bg[3], view[16], proj[16], campos[3];
    const float *means, *shs, *colors_in, *opac, *scales, *rots, *cov_pre; /* borrowed */
    /* per Gaussian */
    float *depth, *xy, *conic_o, *rgb, *cov3d;
    int *radii, *rect; /* rect: xmin,ymin,xmax,ymax */
    uint32_t *tiles;
    uint8_t *clamped; /* 3 per Gaussian */
    /* per instance */
    int64_t I;
    uint64_t *keys;
    uint32_t *vals;
    /* per tile / pixel */
    uint32_t *ranges; /* 2 per tile */
    float *final_T;
    uint32_t *n_contrib;
    int row0, row1;      /* tile-row window composited (bench cpu_baseline sample); default all */
    double t_pre, t_bin, t_comp;
} GsoCtx;

/* Optional tile-row window for the NEXT gso_forward (bench.py bounded CPU sample). */
static int g_row0 = 0, g_row1 = 1 << 30;
void gso_set_tile_rows(int r0, int r1) { g_row0 = r0; g_row1 = r1; }
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* ---------- SH colour (only when `shs` is given; render() never does) ---------- */
/* degree 4: utils/sh_utils.py:38-48 (constants), :97-110 (terms) */
static const float SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f,
                               0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f,
                               0.6258357354491761f};
static void sh_to_rgb(const GsoCtx *c, int idx, float *rgb, uint8_t *clamped) {
    const float *sh = c->shs + (size_t)idx * c->M * 3;
    float dx = c->means[3 * idx + 0] - c->campos[0];
    float dy = c->means[3 * idx + 1] - c->campos[1];
    float dz = c->means[3 * idx + 2] - c->campos[2];
    float len = sqrtf(FMA(dx, dx, FMA(dy, dy, dz * dz)));
    float x = dx / len, y = dy / len, z = dz / len;
    for (int ch = 0; ch < 3; ++ch) {
        float r = SH_C0 * sh[0 * 3 + ch];
        if (c->D > 0) {
            r = FMA(-(SH_C1 * y), sh[1 * 3 + ch], r);
            r = FMA(SH_C1 * z, sh[2 * 3 + ch], r);
            r = FMA(-(SH_C1 * x), sh[3 * 3 + ch], r);
            if (c->D > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = FMA(SH_C2[0] * xy, sh[4 * 3 + ch], r);
                r = FMA(SH_C2[1] * yz, sh[5 * 3 + ch], r);
                r = FMA(SH_C2[2] * (FMA(2.0f, zz, -xx) - yy), sh[6 * 3 + ch], r);
                r = FMA(SH_C2[3] * xz, sh[7 * 3 + ch], r);
                r = FMA(SH_C2[4] * (xx - yy), sh[8 * 3 + ch], r);
                if (c->D > 2) {
                    r = FMA(SH_C3[0] * y * FMA(3.0f, xx, -yy), sh[9 * 3 + ch], r);
                    r = FMA(SH_C3[1] * xy * z, sh[10 * 3 + ch], r);
                    r = FMA(SH_C3[2] * y * (FMA(4.0f, zz, -xx) - yy), sh[11 * 3 + ch], r);
                    r = FMA(SH_C3[3] * z * (FMA(2.0f, zz, -(3.0f * xx)) - 3.0f * yy), sh[12 * 3 + ch], r);
                    r = FMA(SH_C3[4] * x * (FMA(4.0f, zz, -xx) - yy), sh[13 * 3 + ch], r);
                    r = FMA(SH_C3[5] * z * (xx - yy), sh[14 * 3 + ch], r);
                    r = FMA(SH_C3[6] * x * FMA(-3.0f, yy, xx), sh[15 * 3 + ch], r);
                    if (c->D > 3) {
                        r = FMA(SH_C4[0] * xy * (xx - yy), sh[16 * 3 + ch], r);
                        r = FMA(SH_C4[1] * yz * FMA(3.0f, xx, -yy), sh[17 * 3 + ch], r);
                        r = FMA(SH_C4[2] * xy * FMA(7.0f, zz, -1.0f), sh[18 * 3 + ch], r);
                        r = FMA(SH_C4[3] * yz * FMA(7.0f, zz, -3.0f), sh[19 * 3 + ch], r);
                        r = FMA(SH_C4[4] * FMA(zz, FMA(35.0f, zz, -30.0f), 3.0f), sh[20 * 3 + ch], r);
                        r = FMA(SH_C4[5] * xz * FMA(7.0f, zz, -3.0f), sh[21 * 3 + ch], r);
                        r = FMA(SH_C4[6] * (xx - yy) * FMA(7.0f, zz, -1.0f), sh[22 * 3 + ch], r);
                        r = FMA(SH_C4[7] * xz * FMA(-3.0f, yy, xx), sh[23 * 3 + ch], r);
                        r = FMA(SH_C4[8] * (xx * FMA(-3.0f, yy, xx) - yy * FMA(3.0f, xx, -yy)), sh[24 * 3 + ch], r);
                    }
                }
            }
        }
        r = r + 0.5f;
        clamped[ch] = (r < 0.0f);
        rgb[ch] = fmaxf(r, 0.0f);
    }
}

/* ---------- Sigma3 from scale and quaternion (scene/gaussian_model.py:27-31) ---------- */
static void cov3d_from_scale_rot(const float *s3, float mod, const float *q, float *cov) {
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

/* ---------- EWA projection to the 2-D covariance (before dilation) ---------- */
typedef struct {
    float T00, T01, T02, T10, T11, T12; /* T = J * Wr */
    float u0, u1, u2, w0, w1, w2;       /* Sigma3 * T0^T, Sigma3 * T1^T */
    float tx, ty, tz;                   /* clamped view-space point */
    float xmul, ymul;                   /* guard-band gradient masks */
} Cov2dAux;

static void cov2d(const GsoCtx *c, float vx, float vy, float vz, const float *S, float *a, float *b, float *cc,
                  Cov2dAux *aux) {
    const float *V = c->view;
    float limx = GUARD_BAND * c->tanfovx, limy = GUARD_BAND * c->tanfovy;
    float txtz = vx / vz, tytz = vy / vz;
    float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    float tz = vz;
    float J00 = c->focal_x / tz, J02 = -(c->focal_x * tx) / (tz * tz);
    float J11 = c->focal_y / tz, J12 = -(c->focal_y * ty) / (tz * tz);
    /* Wr[r][k] = V[4k + r] */
    float T00 = FMA(J00, V[0], J02 * V[2]), T01 = FMA(J00, V[4], J02 * V[6]), T02 = FMA(J00, V[8], J02 * V[10]);
    float T10 = FMA(J11, V[1], J12 * V[2]), T11 = FMA(J11, V[5], J12 * V[6]), T12 = FMA(J11, V[9], J12 * V[10]);
    float u0 = FMA(S[0], T00, FMA(S[1], T01, S[2] * T02));
    float u1 = FMA(S[1], T00, FMA(S[3], T01, S[4] * T02));
    float u2 = FMA(S[2], T00, FMA(S[4], T01, S[5] * T02));
    float w0 = FMA(S[0], T10, FMA(S[1], T11, S[2] * T12));
    float w1 = FMA(S[1], T10, FMA(S[3], T11, S[4] * T12));
    float w2 = FMA(S[2], T10, FMA(S[4], T11, S[5] * T12));
    *a = FMA(T00, u0, FMA(T01, u1, T02 * u2));
    *b = FMA(T10, u0, FMA(T11, u1, T12 * u2));
    *cc = FMA(T10, w0, FMA(T11, w1, T12 * w2));
    if (aux) {
        aux->T00 = T00; aux->T01 = T01; aux->T02 = T02; aux->T10 = T10; aux->T11 = T11; aux->T12 = T12;
        aux->u0 = u0; aux->u1 = u1; aux->u2 = u2; aux->w0 = w0; aux->w1 = w1; aux->w2 = w2;
        aux->tx = tx; aux->ty = ty; aux->tz = tz;
        aux->xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
        aux->ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    }
}

/* ndc2Pix evaluated in double (gaussian_renderer/__init__.py:238-241), one final rounding */
static inline float ndc2pix_d(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void preprocess(GsoCtx *c) {
    for (int i = 0; i < c->P; ++i) {
        c->radii[i] = 0;
        c->tiles[i] = 0;
        float mx = c->means[3 * i], my = c->means[3 * i + 1], mz = c->means[3 * i + 2];
        float vx = XFORM(c->view, 0, mx, my, mz), vy = XFORM(c->view, 1, mx, my, mz), vz = XFORM(c->view, 2, mx, my, mz);
        if (vz <= NEAR_CULL_Z) continue;
        float hx = XFORM(c->proj, 0, mx, my, mz), hy = XFORM(c->proj, 1, mx, my, mz), hw = XFORM(c->proj, 3, mx, my, mz);
        float pw = 1.0f / (hw + W_EPS);
        float ndcx = hx * pw, ndcy = hy * pw;
        float *S = c->cov3d + 6 * (size_t)i;
        if (c->cov_pre) memcpy(S, c->cov_pre + 6 * (size_t)i, 6 * sizeof(float));
        else cov3d_from_scale_rot(c->scales + 3 * (size_t)i, c->mod, c->rots + 4 * (size_t)i, S);
        float a, b, cc;
        cov2d(c, vx, vy, vz, S, &a, &b, &cc, NULL);
        a = a + DILATION;
        cc = cc + DILATION;
        float det = FMA(a, cc, -(b * b));
        if (det == 0.0f) continue;
        float det_inv = 1.0f / det;
        float conx = cc * det_inv, cony = -b * det_inv, conz = a * det_inv;
        float mid = 0.5f * (a + cc);
        float disc = sqrtf(fmaxf(EIGEN_FLOOR, FMA(mid, mid, -det)));
        float lam1 = mid + disc, lam2 = mid - disc;
        int radius = (int)ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
        float px = ndc2pix_d(ndcx, c->W), py = ndc2pix_d(ndcy, c->H);
        float fr = (float)radius;
        int xmin = clampi((int)((px - fr) / (float)TILE_X), 0, c->gx);
        int ymin = clampi((int)((py - fr) / (float)TILE_Y), 0, c->gy);
        int xmax = clampi((int)((((px + fr) + (float)TILE_X) - 1.0f) / (float)TILE_X), 0, c->gx);
        int ymax = clampi((int)((((py + fr) + (float)TILE_Y) - 1.0f) / (float)TILE_Y), 0, c->gy);
        if ((xmax - xmin) * (ymax - ymin) == 0) continue;
        if (c->shs) sh_to_rgb(c, i, c->rgb + 3 * (size_t)i, c->clamped + 3 * (size_t)i);
        else memcpy(c->rgb + 3 * (size_t)i, c->colors_in + 3 * (size_t)i, 3 * sizeof(float));
        c->depth[i] = vz;
        c->radii[i] = radius;
        c->xy[2 * i] = px;
        c->xy[2 * i + 1] = py;
        c->conic_o[4 * i] = conx; c->conic_o[4 * i + 1] = cony; c->conic_o[4 * i + 2] = conz;
        c->conic_o[4 * i + 3] = c->opac[i];
        c->rect[4 * i] = xmin; c->rect[4 * i + 1] = ymin; c->rect[4 * i + 2] = xmax; c->rect[4 * i + 3] = ymax;
        c->tiles[i] = (uint32_t)((xmax - xmin) * (ymax - ymin));
    }
}

/* stable merge sort of (key, val) pairs by key */
static void merge_sort_pairs(uint64_t *k, uint32_t *v, uint64_t *tk, uint32_t *tv, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (k[j] < k[i]) { tk[o] = k[j]; tv[o++] = v[j++]; }
                else { tk[o] = k[i]; tv[o++] = v[i++]; }
            }
            while (i < mid) { tk[o] = k[i]; tv[o++] = v[i++]; }
            while (j < hi) { tk[o] = k[j]; tv[o++] = v[j++]; }
        }
        memcpy(k, tk, (size_t)n * sizeof(uint64_t));
        memcpy(v, tv, (size_t)n * sizeof(uint32_t));
    }
}

static void binning(GsoCtx *c) {
    int64_t I = 0;
    for (int i = 0; i < c->P; ++i) I += c->tiles[i];
    c->I = I;
    c->keys = (uint64_t *)malloc((size_t)(I ? I : 1) * sizeof(uint64_t));
    c->vals = (uint32_t *)malloc((size_t)(I ? I : 1) * sizeof(uint32_t));
    int64_t off = 0;
    for (int i = 0; i < c->P; ++i) {
        if (c->radii[i] <= 0) continue;
        uint32_t dbits;
        memcpy(&dbits, &c->depth[i], 4);
        for (int y = c->rect[4 * i + 1]; y < c->rect[4 * i + 3]; ++y)
            for (int x = c->rect[4 * i]; x < c->rect[4 * i + 2]; ++x) {
                uint64_t key = (uint64_t)(y * c->gx + x);
                key = (key << 32) | dbits;
                c->keys[off] = key;
                c->vals[off] = (uint32_t)i;
                ++off;
            }
    }
    uint64_t *tk = (uint64_t *)malloc((size_t)(I ? I : 1) * sizeof(uint64_t));
    uint32_t *tv = (uint32_t *)malloc((size_t)(I ? I : 1) * sizeof(uint32_t));
    merge_sort_pairs(c->keys, c->vals, tk, tv, I);
    free(tk);
    free(tv);
    memset(c->ranges, 0, (size_t)c->T * 2 * sizeof(uint32_t));
    for (int64_t j = 0; j < I; ++j) {
        uint32_t tile = (uint32_t)(c->keys[j] >> 32);
        if (j == 0 || (uint32_t)(c->keys[j - 1] >> 32) != tile) c->ranges[2 * tile] = (uint32_t)j;
        if (j == I - 1 || (uint32_t)(c->keys[j + 1] >> 32) != tile) c->ranges[2 * tile + 1] = (uint32_t)(j + 1);
    }
}

static void composite(GsoCtx *c, float *out) {
    const int W = c->W, H = c->H;
    for (int ty = (c->row0 > 0 ? c->row0 : 0); ty < c->gy && ty < c->row1; ++ty)
        for (int tx = 0; tx < c->gx; ++tx) {
            uint32_t lo = c->ranges[2 * (ty * c->gx + tx)], hi = c->ranges[2 * (ty * c->gx + tx) + 1];
            for (int py = ty * TILE_Y; py < (ty + 1) * TILE_Y && py < H; ++py)
                for (int px = tx * TILE_X; px < (tx + 1) * TILE_X && px < W; ++px) {
                    float pfx = (float)px, pfy = (float)py;
                    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
                    uint32_t contributor = 0, last = 0;
                    for (uint32_t j = lo; j < hi; ++j) {
                        ++contributor;
                        uint32_t g = c->vals[j];
                        float dx = c->xy[2 * g] - pfx, dy = c->xy[2 * g + 1] - pfy;
                        float cx = c->conic_o[4 * g], cy = c->conic_o[4 * g + 1], cz = c->conic_o[4 * g + 2];
                        float o = c->conic_o[4 * g + 3];
                        float q = FMA(cz * dy, dy, (cx * dx) * dx);
                        float power = FMA(-0.5f, q, -((cy * dx) * dy));
                        if (power > 0.0f) continue;
                        float G = exp_det(power);
                        float alpha = fminf(ALPHA_CLAMP, o * G);
                        if (alpha < ALPHA_SKIP) continue;
                        float w = alpha * T;
                        float test_T = T - w;   /* = T(1-alpha), one rounding less */
                        if (test_T < T_STOP) break; /* pixel done; entry not applied */
                        C0 = FMA(c->rgb[3 * g], w, C0);
                        C1 = FMA(c->rgb[3 * g + 1], w, C1);
                        C2 = FMA(c->rgb[3 * g + 2], w, C2);
                        T = test_T;
                        last = contributor;
                    }
                    size_t pix = (size_t)py * W + px;
                    c->final_T[pix] = T;
                    c->n_contrib[pix] = last;
                    out[0 * (size_t)H * W + pix] = FMA(T, c->bg[0], C0);
                    out[1 * (size_t)H * W + pix] = FMA(T, c->bg[1], C1);
                    out[2 * (size_t)H * W + pix] = FMA(T, c->bg[2], C2);
                }
        }
}

GsoCtx *gso_forward(int P, int D, int M, const float *bg, int W, int H, const float *means, const float *shs,
                    const float *colors, const float *opac, const float *scales, float mod, const float *rots,
                    const float *cov_pre, const float *view, const float *proj, const float *campos, float tanfovx,
                    float tanfovy, float *out_color, int *radii) {
    GsoCtx *c = (GsoCtx *)calloc(1, sizeof(GsoCtx));
    c->P = P; c->D = D; c->M = M; c->W = W; c->H = H;
    c->gx = (W + TILE_X - 1) / TILE_X; c->gy = (H + TILE_Y - 1) / TILE_Y; c->T = c->gx * c->gy;
    c->tanfovx = tanfovx; c->tanfovy = tanfovy; c->mod = mod;
    c->focal_x = (float)W / (2.0f * tanfovx);
    c->focal_y = (float)H / (2.0f * tanfovy);
    memcpy(c->bg, bg, 12); memcpy(c->view, view, 64); memcpy(c->proj, proj, 64); memcpy(c->campos, campos, 12);
    c->means = means; c->shs = shs; c->colors_in = colors; c->opac = opac; c->scales = scales; c->rots = rots;
    c->cov_pre = cov_pre;
    size_t n = (size_t)(P ? P : 1);
    c->depth = (float *)calloc(n, 4); c->xy = (float *)calloc(n, 8); c->conic_o = (float *)calloc(n, 16);
    c->rgb = (float *)calloc(n, 12); c->cov3d = (float *)calloc(n, 24); c->radii = (int *)calloc(n, 4);
    c->rect = (int *)calloc(n, 16); c->tiles = (uint32_t *)calloc(n, 4); c->clamped = (uint8_t *)calloc(n, 3);
    c->ranges = (uint32_t *)calloc((size_t)c->T, 8);
    c->final_T = (float *)calloc((size_t)W * H, 4);
    c->n_contrib = (uint32_t *)calloc((size_t)W * H, 4);
    c->row0 = g_row0; c->row1 = g_row1;
    g_row0 = 0; g_row1 = 1 << 30;
    double t0 = now_s();
    preprocess(c);
    double t1 = now_s();
    binning(c);
    double t2 = now_s();
    composite(c, out_color);
    c->t_pre = t1 - t0; c->t_bin = t2 - t1; c->t_comp = now_s() - t2;
    if (radii) memcpy(radii, c->radii, (size_t)P * 4);
    return c;
}

int64_t gso_num_rendered(const GsoCtx *c) { return c->I; }
void gso_timings(const GsoCtx *c, double *out3) { out3[0] = c->t_pre; out3[1] = c->t_bin; out3[2] = c->t_comp; }
const float *gso_depth(const GsoCtx *c) { return c->depth; }
const float *gso_xy(const GsoCtx *c) { return c->xy; }
const float *gso_conic_opacity(const GsoCtx *c) { return c->conic_o; }
const float *gso_rgb(const GsoCtx *c) { return c->rgb; }
const float *gso_cov3d(const GsoCtx *c) { return c->cov3d; }
const int *gso_rect(const GsoCtx *c) { return c->rect; }
const uint32_t *gso_tiles_touched(const GsoCtx *c) { return c->tiles; }
const uint8_t *gso_clamped(const GsoCtx *c) { return c->clamped; }
const uint64_t *gso_keys(const GsoCtx *c) { return c->keys; }
const uint32_t *gso_point_list(const GsoCtx *c) { return c->vals; }
const uint32_t *gso_ranges(const GsoCtx *c) { return c->ranges; }
const float *gso_final_T(const GsoCtx *c) { return c->final_T; }
const uint32_t *gso_n_contrib(const GsoCtx *c) { return c->n_contrib; }

void gso_free(GsoCtx *c) {
    if (!c) return;
    free(c->depth); free(c->xy); free(c->conic_o); free(c->rgb); free(c->cov3d); free(c->radii); free(c->rect);
    free(c->tiles); free(c->clamped); free(c->keys); free(c->vals); free(c->ranges); free(c->final_T);
    free(c->n_contrib); free(c);
}

/* =====================================================================
 * Backward.  Per-pixel terms in fp32 as the reference
 * op does; the sums over pixels are kept in double (the reference uses
 * order-nondeterministic float atomics, so any summation order is valid).
 * ===================================================================== */
typedef struct { double mean2d[2], conic[3], opacity, color[3]; } PixAcc;

static void composite_backward(const GsoCtx *c, const float *dL_dpix, PixAcc *acc) {
    const int W = c->W, H = c->H;
    const size_t HW = (size_t)H * W;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    for (int ty = (c->row0 > 0 ? c->row0 : 0); ty < c->gy && ty < c->row1; ++ty)
        for (int tx = 0; tx < c->gx; ++tx) {
            uint32_t lo = c->ranges[2 * (ty * c->gx + tx)];
            for (int py = ty * TILE_Y; py < (ty + 1) * TILE_Y && py < H; ++py)
                for (int px = tx * TILE_X; px < (tx + 1) * TILE_X && px < W; ++px) {
                    size_t pix = (size_t)py * W + px;
                    float pfx = (float)px, pfy = (float)py;
                    const float T_final = c->final_T[pix];
                    float T = T_final;
                    uint32_t last = c->n_contrib[pix];
                    float dp[3] = {dL_dpix[pix], dL_dpix[HW + pix], dL_dpix[2 * HW + pix]};
                    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.0f;
                    float bg_dot = FMA(c->bg[0], dp[0], FMA(c->bg[1], dp[1], c->bg[2] * dp[2]));
                    for (uint32_t k = last; k-- > 0;) {
                        uint32_t g = c->vals[lo + k];
                        float dx = c->xy[2 * g] - pfx, dy = c->xy[2 * g + 1] - pfy;
                        float cx = c->conic_o[4 * g], cy = c->conic_o[4 * g + 1], cz = c->conic_o[4 * g + 2];
                        float o = c->conic_o[4 * g + 3];
                        float q = FMA(cz * dy, dy, (cx * dx) * dx);
                        float power = FMA(-0.5f, q, -((cy * dx) * dy));
                        if (power > 0.0f) continue;
                        float G = exp_det(power);
                        float alpha = fminf(ALPHA_CLAMP, o * G);
                        if (alpha < ALPHA_SKIP) continue;
                        T = T / (1.0f - alpha);
                        float dchannel_dcolor = alpha * T;
                        float dL_dalpha = 0.0f;
                        for (int ch = 0; ch < 3; ++ch) {
                            float col = c->rgb[3 * g + ch];
                            accum_rec[ch] = FMA(last_alpha, last_color[ch], (1.0f - last_alpha) * accum_rec[ch]);
                            last_color[ch] = col;
                            dL_dalpha = FMA(col - accum_rec[ch], dp[ch], dL_dalpha);
                            acc[g].color[ch] += (double)(dchannel_dcolor * dp[ch]);
                        }
                        dL_dalpha = dL_dalpha * T;
                        last_alpha = alpha;
                        dL_dalpha = FMA(-T_final / (1.0f - alpha), bg_dot, dL_dalpha);
                        /* straight-through min(0.99, .): gradient as if alpha = o*G */
                        float dL_dG = o * dL_dalpha;
                        float gdx = G * dx, gdy = G * dy;
                        float dG_ddelx = FMA(-gdx, cx, -(gdy * cy));
                        float dG_ddely = FMA(-gdy, cz, -(gdx * cy));
                        acc[g].mean2d[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        acc[g].mean2d[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        acc[g].conic[0] += (double)(-0.5f * gdx * dx * dL_dG);
                        acc[g].conic[1] += (double)(-(gdx * dy) * dL_dG); /* true d/d(conic.y) */
                        acc[g].conic[2] += (double)(-0.5f * gdy * dy * dL_dG);
                        acc[g].opacity += (double)(G * dL_dalpha);
                    }
                }
        }
}

/* SH backward for one Gaussian: dL/dsh and dL/dmean through the view direction */
static void sh_backward(const GsoCtx *c, int idx, const float *dL_drgb_in, float *dL_dsh, float *dL_dmean) {
    const float *sh = c->shs + (size_t)idx * c->M * 3;
    float *dsh = dL_dsh + (size_t)idx * c->M * 3;
    float ox = c->means[3 * idx] - c->campos[0], oy = c->means[3 * idx + 1] - c->campos[1],
          oz = c->means[3 * idx + 2] - c->campos[2];
    float len = sqrtf(ox * ox + oy * oy + oz * oz);
    float x = ox / len, y = oy / len, z = oz / len;
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    float g[3];
    for (int ch = 0; ch < 3; ++ch) g[ch] = c->clamped[3 * idx + ch] ? 0.0f : dL_drgb_in[ch];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    /* basis values and their gradients w.r.t. (x,y,z) */
    float Y[25], Yx[25], Yy[25], Yz[25];
    memset(Yx, 0, sizeof Yx); memset(Yy, 0, sizeof Yy); memset(Yz, 0, sizeof Yz); memset(Y, 0, sizeof Y);
    Y[0] = SH_C0;
    Y[1] = -SH_C1 * y; Yy[1] = -SH_C1;
    Y[2] = SH_C1 * z; Yz[2] = SH_C1;
    Y[3] = -SH_C1 * x; Yx[3] = -SH_C1;
    Y[4] = SH_C2[0] * xy; Yx[4] = SH_C2[0] * y; Yy[4] = SH_C2[0] * x;
    Y[5] = SH_C2[1] * yz; Yy[5] = SH_C2[1] * z; Yz[5] = SH_C2[1] * y;
    Y[6] = SH_C2[2] * (2.0f * zz - xx - yy); Yx[6] = SH_C2[2] * -2.0f * x; Yy[6] = SH_C2[2] * -2.0f * y; Yz[6] = SH_C2[2] * 4.0f * z;
    Y[7] = SH_C2[3] * xz; Yx[7] = SH_C2[3] * z; Yz[7] = SH_C2[3] * x;
    Y[8] = SH_C2[4] * (xx - yy); Yx[8] = SH_C2[4] * 2.0f * x; Yy[8] = SH_C2[4] * -2.0f * y;
    Y[9] = SH_C3[0] * y * (3.0f * xx - yy); Yx[9] = SH_C3[0] * 6.0f * xy; Yy[9] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
    Y[10] = SH_C3[1] * xy * z; Yx[10] = SH_C3[1] * yz; Yy[10] = SH_C3[1] * xz; Yz[10] = SH_C3[1] * xy;
    Y[11] = SH_C3[2] * y * (4.0f * zz - xx - yy); Yx[11] = SH_C3[2] * -2.0f * xy; Yy[11] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); Yz[11] = SH_C3[2] * 8.0f * yz;
    Y[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); Yx[12] = SH_C3[3] * -6.0f * xz; Yy[12] = SH_C3[3] * -6.0f * yz; Yz[12] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    Y[13] = SH_C3[4] * x * (4.0f * zz - xx - yy); Yx[13] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); Yy[13] = SH_C3[4] * -2.0f * xy; Yz[13] = SH_C3[4] * 8.0f * xz;
    Y[14] = SH_C3[5] * z * (xx - yy); Yx[14] = SH_C3[5] * 2.0f * xz; Yy[14] = SH_C3[5] * -2.0f * yz; Yz[14] = SH_C3[5] * (xx - yy);
    Y[15] = SH_C3[6] * x * (xx - 3.0f * yy); Yx[15] = SH_C3[6] * (3.0f * xx - 3.0f * yy); Yy[15] = SH_C3[6] * -6.0f * xy;
    /* degree 4: partial derivatives of the polynomials AS WRITTEN in utils/sh_utils.py:101-109 (x, y, z independent) */
    Y[16] = SH_C4[0] * xy * (xx - yy); Yx[16] = SH_C4[0] * y * (3.0f * xx - yy); Yy[16] = SH_C4[0] * x * (xx - 3.0f * yy);
    Y[17] = SH_C4[1] * yz * (3.0f * xx - yy); Yx[17] = SH_C4[1] * 6.0f * xy * z; Yy[17] = SH_C4[1] * z * (3.0f * xx - 3.0f * yy); Yz[17] = SH_C4[1] * y * (3.0f * xx - yy);
    Y[18] = SH_C4[2] * xy * (7.0f * zz - 1.0f); Yx[18] = SH_C4[2] * y * (7.0f * zz - 1.0f); Yy[18] = SH_C4[2] * x * (7.0f * zz - 1.0f); Yz[18] = SH_C4[2] * 14.0f * xy * z;
    Y[19] = SH_C4[3] * yz * (7.0f * zz - 3.0f); Yy[19] = SH_C4[3] * z * (7.0f * zz - 3.0f); Yz[19] = SH_C4[3] * y * (21.0f * zz - 3.0f);
    Y[20] = SH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f); Yz[20] = SH_C4[4] * z * (140.0f * zz - 60.0f);
    Y[21] = SH_C4[5] * xz * (7.0f * zz - 3.0f); Yx[21] = SH_C4[5] * z * (7.0f * zz - 3.0f); Yz[21] = SH_C4[5] * x * (21.0f * zz - 3.0f);
    Y[22] = SH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f); Yx[22] = SH_C4[6] * 2.0f * x * (7.0f * zz - 1.0f); Yy[22] = SH_C4[6] * -2.0f * y * (7.0f * zz - 1.0f); Yz[22] = SH_C4[6] * 14.0f * z * (xx - yy);
    Y[23] = SH_C4[7] * xz * (xx - 3.0f * yy); Yx[23] = SH_C4[7] * z * (3.0f * xx - 3.0f * yy); Yy[23] = SH_C4[7] * -6.0f * xy * z; Yz[23] = SH_C4[7] * x * (xx - 3.0f * yy);
    Y[24] = SH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)); Yx[24] = SH_C4[8] * x * (4.0f * xx - 12.0f * yy); Yy[24] = SH_C4[8] * y * (4.0f * yy - 12.0f * xx);
    int ncoef = (c->D + 1) * (c->D + 1);
    for (int k = 0; k < ncoef; ++k)
        for (int ch = 0; ch < 3; ++ch) {
            dsh[k * 3 + ch] = Y[k] * g[ch];
            dRGBdx[ch] += Yx[k] * sh[k * 3 + ch];
            dRGBdy[ch] += Yy[k] * sh[k * 3 + ch];
            dRGBdz[ch] += Yz[k] * sh[k * 3 + ch];
        }
    float ddx = dRGBdx[0] * g[0] + dRGBdx[1] * g[1] + dRGBdx[2] * g[2];
    float ddy = dRGBdy[0] * g[0] + dRGBdy[1] * g[1] + dRGBdy[2] * g[2];
    float ddz = dRGBdz[0] * g[0] + dRGBdz[1] * g[1] + dRGBdz[2] * g[2];
    /* d(v/|v|)/dv applied to (ddx,ddy,ddz) */
    float dot = x * ddx + y * ddy + z * ddz;
    dL_dmean[0] += (ddx - x * dot) / len;
    dL_dmean[1] += (ddy - y * dot) / len;
    dL_dmean[2] += (ddz - z * dot) / len;
}

/* All outputs must be zero-initialised by the caller; any may be NULL. */
void gso_backward(const GsoCtx *c, const float *dL_dpix, float *dL_dmean2D /*P,3*/, float *dL_dconic /*P,3 true grads*/,
                  float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                  float *dL_dscale, float *dL_drot) {
    const int P = c->P;
    PixAcc *acc = (PixAcc *)calloc((size_t)(P ? P : 1), sizeof(PixAcc));
    composite_backward(c, dL_dpix, acc);
    const float *V = c->view, *Pm = c->proj;
    for (int i = 0; i < P; ++i) {
        float gm2x = (float)acc[i].mean2d[0], gm2y = (float)acc[i].mean2d[1];
        float gA = (float)acc[i].conic[0], gB = (float)acc[i].conic[1], gC = (float)acc[i].conic[2];
        float gcol[3] = {(float)acc[i].color[0], (float)acc[i].color[1], (float)acc[i].color[2]};
        if (dL_dmean2D) { dL_dmean2D[3 * i] = gm2x; dL_dmean2D[3 * i + 1] = gm2y; }
        if (dL_dconic) { dL_dconic[3 * i] = gA; dL_dconic[3 * i + 1] = gB; dL_dconic[3 * i + 2] = gC; }
        if (dL_dopacity) dL_dopacity[i] = (float)acc[i].opacity;
        if (dL_dcolor) { dL_dcolor[3 * i] = gcol[0]; dL_dcolor[3 * i + 1] = gcol[1]; dL_dcolor[3 * i + 2] = gcol[2]; }
        if (c->radii[i] <= 0) continue;
        float mx = c->means[3 * i], my = c->means[3 * i + 1], mz = c->means[3 * i + 2];
        float gmean[3] = {0, 0, 0};
        /* ---- B.2: conic -> Sigma2 -> Sigma3, and -> view-space point ---- */
        const float *S = c->cov3d + 6 * (size_t)i;
        float vx = XFORM(V, 0, mx, my, mz), vy = XFORM(V, 1, mx, my, mz), vz = XFORM(V, 2, mx, my, mz);
        float a, b, cc;
        Cov2dAux x;
        cov2d(c, vx, vy, vz, S, &a, &b, &cc, &x);
        a += DILATION; cc += DILATION;
        float det = a * cc - b * b;
        float d2inv = 1.0f / (det * det + DET2_EPS);
        float g_a = 0, g_b = 0, g_c = 0;
        if (det != 0.0f) {
            g_a = d2inv * (-cc * cc * gA + b * cc * gB - b * b * gC);
            g_c = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
            g_b = d2inv * (2.0f * b * cc * gA - (a * cc + b * b) * gB + 2.0f * a * b * gC);
        }
        float gcov[6];
        gcov[0] = x.T00 * x.T00 * g_a + x.T00 * x.T10 * g_b + x.T10 * x.T10 * g_c;
        gcov[3] = x.T01 * x.T01 * g_a + x.T01 * x.T11 * g_b + x.T11 * x.T11 * g_c;
        gcov[5] = x.T02 * x.T02 * g_a + x.T02 * x.T12 * g_b + x.T12 * x.T12 * g_c;
        gcov[1] = 2.0f * x.T00 * x.T01 * g_a + (x.T00 * x.T11 + x.T01 * x.T10) * g_b + 2.0f * x.T10 * x.T11 * g_c;
        gcov[2] = 2.0f * x.T00 * x.T02 * g_a + (x.T00 * x.T12 + x.T02 * x.T10) * g_b + 2.0f * x.T10 * x.T12 * g_c;
        gcov[4] = 2.0f * x.T02 * x.T01 * g_a + (x.T01 * x.T12 + x.T02 * x.T11) * g_b + 2.0f * x.T11 * x.T12 * g_c;
        if (dL_dcov3D) memcpy(dL_dcov3D + 6 * (size_t)i, gcov, 24);
        float gT00 = 2.0f * g_a * x.u0 + g_b * x.w0, gT01 = 2.0f * g_a * x.u1 + g_b * x.w1, gT02 = 2.0f * g_a * x.u2 + g_b * x.w2;
        float gT10 = 2.0f * g_c * x.w0 + g_b * x.u0, gT11 = 2.0f * g_c * x.w1 + g_b * x.u1, gT12 = 2.0f * g_c * x.w2 + g_b * x.u2;
        float gJ00 = V[0] * gT00 + V[4] * gT01 + V[8] * gT02;
        float gJ02 = V[2] * gT00 + V[6] * gT01 + V[10] * gT02;
        float gJ11 = V[1] * gT10 + V[5] * gT11 + V[9] * gT12;
        float gJ12 = V[2] * gT10 + V[6] * gT11 + V[10] * gT12;
        float tz = 1.0f / x.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        float gtx = x.xmul * (-c->focal_x * tz2 * gJ02);
        float gty = x.ymul * (-c->focal_y * tz2 * gJ12);
        float gtz = -c->focal_x * tz2 * gJ00 - c->focal_y * tz2 * gJ11 + (2.0f * c->focal_x * x.tx) * tz3 * gJ02 +
                    (2.0f * c->focal_y * x.ty) * tz3 * gJ12;
        gmean[0] += V[0] * gtx + V[1] * gty + V[2] * gtz;
        gmean[1] += V[4] * gtx + V[5] * gty + V[6] * gtz;
        gmean[2] += V[8] * gtx + V[9] * gty + V[10] * gtz;
        /* ---- B.3: NDC mean gradient through the projection ---- */
        float hx = XFORM(Pm, 0, mx, my, mz), hy = XFORM(Pm, 1, mx, my, mz), hw = XFORM(Pm, 3, mx, my, mz);
        float pw = 1.0f / (hw + W_EPS);
        float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
        gmean[0] += (Pm[0] * pw - Pm[3] * mul1) * gm2x + (Pm[1] * pw - Pm[3] * mul2) * gm2y;
        gmean[1] += (Pm[4] * pw - Pm[7] * mul1) * gm2x + (Pm[5] * pw - Pm[7] * mul2) * gm2y;
        gmean[2] += (Pm[8] * pw - Pm[11] * mul1) * gm2x + (Pm[9] * pw - Pm[11] * mul2) * gm2y;
        if (c->shs && dL_dsh) sh_backward(c, i, gcol, dL_dsh, gmean);
        if (dL_dmean3D) { dL_dmean3D[3 * i] = gmean[0]; dL_dmean3D[3 * i + 1] = gmean[1]; dL_dmean3D[3 * i + 2] = gmean[2]; }
        /* ---- Sigma3 = (R diag s)(R diag s)^T backward ---- */
        if (!c->cov_pre && dL_dscale && dL_drot) {
            const float *q = c->rots + 4 * (size_t)i, *s3 = c->scales + 3 * (size_t)i;
            float s[3] = {c->mod * s3[0], c->mod * s3[1], c->mod * s3[2]};
            float r = q[0], qx = q[1], qy = q[2], qz = q[3];
            float R[3][3] = {{1.0f - 2.0f * (qy * qy + qz * qz), 2.0f * (qx * qy - r * qz), 2.0f * (qx * qz + r * qy)},
                             {2.0f * (qx * qy + r * qz), 1.0f - 2.0f * (qx * qx + qz * qz), 2.0f * (qy * qz - r * qx)},
                             {2.0f * (qx * qz - r * qy), 2.0f * (qy * qz + r * qx), 1.0f - 2.0f * (qx * qx + qy * qy)}};
            float Gs[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                              {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                              {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
            float dLm[3][3]; /* dL/dL, L = R diag(s) */
            for (int a2 = 0; a2 < 3; ++a2)
                for (int j = 0; j < 3; ++j) {
                    float t = 0;
                    for (int k = 0; k < 3; ++k) t += Gs[a2][k] * (R[k][j] * s[j]);
                    dLm[a2][j] = 2.0f * t;
                }
            float dR[3][3];
            for (int j = 0; j < 3; ++j) {
                float t = 0;
                for (int a2 = 0; a2 < 3; ++a2) { t += R[a2][j] * dLm[a2][j]; dR[a2][j] = s[j] * dLm[a2][j]; }
                dL_dscale[3 * i + j] = c->mod * t;
            }
            dL_drot[4 * i + 0] = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
            dL_drot[4 * i + 1] = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.0f * qx * dR[1][1] - r * dR[1][2] + qz * dR[2][0] + r * dR[2][1] - 2.0f * qx * dR[2][2]);
            dL_drot[4 * i + 2] = 2.0f * (-2.0f * qy * dR[0][0] + qx * dR[0][1] + r * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - r * dR[2][0] + qz * dR[2][1] - 2.0f * qy * dR[2][2]);
            dL_drot[4 * i + 3] = 2.0f * (-2.0f * qz * dR[0][0] - r * dR[0][1] + qx * dR[0][2] + r * dR[1][0] - 2.0f * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        }
    }
    free(acc);
}
