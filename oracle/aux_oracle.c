/*
 * aux_oracle.c -- CPU restatements of the small kernels around the rasteriser.
 * TEST INFRASTRUCTURE ONLY (see gs_oracle.c header for the import rule).
 *
 *   gso_knn3        simple_knn._C.distCUDA2 (scene/gaussian_model.py:134): exact
 *                   mean squared distance to the 3 nearest other points.  The
 *                   CUDA source is an un-vendored submodule (.gitmodules:1-3) ->
 *                   parity unpinned; exactness makes brute force (here) and
 *                   scipy cKDTree (tests) valid independent checks.
 *   gso_event_loss  train.py:165-203 with utils/loss_utils.py:24-28,234-249,270-271;
 *                   pinned by tests/golden/event_loss.npz (generated from the
 *                   reference's own functions).
 *   gso_adam        torch.optim.Adam single-tensor update (train.py:330-332,
 *                   scene/gaussian_model.py:154-163), pinned against torch in tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

void gso_knn3(int P, const float *pts, float *out) {
    for (int i = 0; i < P; ++i) {
        float best[3] = {INFINITY, INFINITY, INFINITY};
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < P; ++j) {
            if (j == i) continue;
            float dx = pts[3 * j] - x, dy = pts[3 * j + 1] - y, dz = pts[3 * j + 2] - z;
            float d = dx * dx + dy * dy + dz * dz;
            if (d < best[2]) {
                if (d < best[1]) {
                    best[2] = best[1];
                    if (d < best[0]) { best[1] = best[0]; best[0] = d; }
                    else best[1] = d;
                } else best[2] = d;
            }
        }
        float s = 0.0f; int n = 0;
        for (int k = 0; k < 3; ++k) if (best[k] != INFINITY) { s += best[k]; ++n; }
        out[i] = n ? s / 3.0f : 0.0f;
        if (n && n < 3) out[i] = s / (float)n;
    }
}

static inline float lum(const float *img, size_t HW, size_t p) {
    return 0.4124f * img[p] + 0.35758f * img[HW + p] + 0.1804f * img[2 * HW + p];
}

/* scalars: [0] loss [1] dL/dc [2] rho [3] L1 event [4] L1 intensity [5] L1 blur */
void gso_event_loss(int W, int H, const float *image, const float *now, const float *next, const float *gt_int,
                    const float *gt_now, const float *gt_next, const float *gt_blur, float c, float gt_c,
                    float *d_image, float *d_now, float *d_next, double *scalars) {
    const size_t HW = (size_t)W * H;
    const float eps = 1e-8f;
    double sumE = 0, sumI = 0, sumB = 0, cnt = 0, sumSD = 0;
    float *D = (float *)malloc(HW * 4), *Dg = (float *)malloc(HW * 4);
    for (size_t p = 0; p < HW; ++p) {
        D[p] = (logf(lum(next, HW, p) + eps) - logf(lum(now, HW, p) + eps)) / c;
        Dg[p] = (logf(lum(gt_next, HW, p) + eps) - logf(lum(gt_now, HW, p) + eps)) / gt_c;
        float e = D[p] - Dg[p];
        sumE += fabsf(e);
        if (Dg[p] != 0.0f) cnt += 1.0;
        float sg = (e > 0) - (e < 0);
        sumSD += (double)sg * D[p];
    }
    for (size_t p = 0; p < 3 * HW; ++p) {
        sumI += fabsf(image[p] - gt_int[p]);
        if (gt_blur) sumB += fabsf(image[p] - gt_blur[p]);
    }
    double L1E = sumE / HW, L1I = sumI / (3.0 * HW), L1B = sumB / (3.0 * HW), rho = cnt / HW;
    double loss = 0.9 * L1E * rho + 0.1 * L1I * (1.0 - rho);
    double outer = 1.0;
    if (gt_blur) { loss = 0.5 * loss + 0.5 * L1B; outer = 0.5; }
    scalars[0] = loss; scalars[2] = rho; scalars[3] = L1E; scalars[4] = L1I; scalars[5] = L1B;
    scalars[1] = -outer * 0.9 * rho * (sumSD / HW) / c;
    const float wch[3] = {0.4124f, 0.35758f, 0.1804f};
    for (size_t p = 0; p < HW; ++p) {
        float e = D[p] - Dg[p];
        float sg = (e > 0) - (e < 0);
        float k = (float)(outer * 0.9 * rho / HW) * sg / c;
        float yn = lum(next, HW, p) + eps, yo = lum(now, HW, p) + eps;
        for (int ch = 0; ch < 3; ++ch) {
            d_next[ch * HW + p] = k * wch[ch] / yn;
            d_now[ch * HW + p] = -k * wch[ch] / yo;
        }
    }
    for (size_t p = 0; p < 3 * HW; ++p) {
        float e = image[p] - gt_int[p];
        float g = (float)(outer * 0.1 * (1.0 - rho) / (3.0 * HW)) * (float)((e > 0) - (e < 0));
        if (gt_blur) {
            float eb = image[p] - gt_blur[p];
            g += (float)(0.5 / (3.0 * HW)) * (float)((eb > 0) - (eb < 0));
        }
        d_image[p] = g;
    }
    free(D); free(Dg);
}

/* torch.optim.Adam evaluates 1 - beta, beta^step on the Python double the caller wrote (0.9, 0.999); the betas arrive here
 * as fp32, so the decimal is recovered first (7 digits) -- (float)0.999 would put 1 - beta2 1.3e-5 away from torch's. */
static double beta_double(float b) {
    const double d = nearbyint((double)b * 1e7) / 1e7;       /* only when it IS the caller's decimal and < 1 */
    return ((float)d == b && d >= 0.0 && d < 1.0) ? d : (double)b;
}
void gso_adam(size_t n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps, int step) {
    double bc1 = 1.0 - pow(beta_double(b1), step), bc2 = 1.0 - pow(beta_double(b2), step);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - beta_double(b1)), omb2 = (float)(1.0 - beta_double(b2));
    for (size_t i = 0; i < n; ++i) {
        m[i] = m[i] + omb1 * (g[i] - m[i]);                  /* lerp form used by torch */
        v[i] = v[i] * b2 + omb2 * g[i] * g[i];
        float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}
