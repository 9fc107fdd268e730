// densify.hip -- adaptive density control on the device.
//
// scene/gaussian_model.py:389-403 (densify_and_prune) = clone (:374-387) + split (:349-372) + prune (:273-305,396-402),
// each of which the reference runs as boolean-mask gathers, torch.cat and optimizer-state surgery over six tensors.
// Here the whole step is ONE plan pass (per-Gaussian decisions -> four exclusive scans) and ONE apply pass that
// compacts parameters AND both Adam moments straight from the old flat training buffers into the new ones
// (xyz | SH coefficient-major (48, N) | opacity | scaling | rotation | c), appends the clones and the split children
// in the reference's row order, and never materialises the intermediate (post-clone, post-split) models:
//
//   rows of the result = [kept originals, in order] [kept clones] [kept first children] [kept second children]
//
// which is exactly what clone -> split -> prune produces, because every prune criterion is a row-wise function of the
// row's own (possibly new) parameters.  Faithfully kept quirks: the split sees the PRE-clone gradient statistics
// (clones have none), and max_radii2D was zeroed by the postfix before the size test reads it (:347 vs :398), so only
// the world-space size test can fire.
#include "common.h"

int e3_fail(hipError_t e, const char* what);

struct FlatLayout {      // offsets (in floats) of the groups inside one flat training buffer of N Gaussians
    size_t xyz, feat, opac, scal, rot, c;
    __host__ __device__ static FlatLayout of(size_t N) {
        FlatLayout L;
        L.xyz = 0; L.feat = 3 * N; L.opac = 51 * N; L.scal = 52 * N; L.rot = 55 * N; L.c = 59 * N;
        return L;
    }
};

// flags[i]: bit 0 keep original, bit 1 keep clone, bit 2 selected for split, bit 3 keep the two children
__global__ __launch_bounds__(256) void densify_flags_kernel(int N, const float* __restrict__ param,
                                                            const float* __restrict__ grad_accum,
                                                            const float* __restrict__ denom, float max_grad,
                                                            float min_opacity, float thr_dense, float thr_big, int size_prune,
                                                            uint32_t* __restrict__ f_orig, uint32_t* __restrict__ f_clone,
                                                            uint32_t* __restrict__ f_split, uint32_t* __restrict__ f_child) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const FlatLayout L = FlatLayout::of((size_t)N);
    float g = grad_accum[i] / denom[i];                    // xyz_gradient_accum / denom, NaN (0/0) -> 0   (:390-391)
    if (g != g) g = 0.0f;
    const float* s = param + L.scal + 3 * (size_t)i;
    const float e0 = expf(s[0]), e1 = expf(s[1]), e2 = expf(s[2]);
    const float smax = fmaxf(fmaxf(e0, e1), e2);
    const bool clone = (fabsf(g) >= max_grad) && (smax <= thr_dense);       // :376-378
    const bool split = (g >= max_grad) && (smax > thr_dense);              // :354-356
    const bool low = sigmoid_libm(param[L.opac + i]) < min_opacity;          // :396
    const bool big = size_prune && (smax > thr_big);                        // :399 (world-space half of the size test)
    // the children carry scaling = log(exp(s) / (0.8 * 2)) (:362): their size test sees exp() of THAT
    const float c0 = expf(logf(e0 / 1.6f)), c1 = expf(logf(e1 / 1.6f)), c2 = expf(logf(e2 / 1.6f));
    const bool big_child = size_prune && (fmaxf(fmaxf(c0, c1), c2) > thr_big);
    f_orig[i] = (!split && !low && !big) ? 1u : 0u;                         // split parents are pruned (:371-372)
    f_clone[i] = (clone && !low && !big) ? 1u : 0u;
    f_split[i] = split ? 1u : 0u;
    f_child[i] = (split && !low && !big_child) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void densify_split_rows_kernel(int N, const uint32_t* __restrict__ f_split,
                                                                 const uint32_t* __restrict__ s_split,
                                                                 int* __restrict__ rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && f_split[i]) rows[s_split[i]] = i;
}

struct DensifyPlan {
    const uint32_t *f_orig, *f_clone, *f_split, *f_child;      // flags
    const uint32_t *s_orig, *s_clone, *s_split, *s_child;      // exclusive scans of the flags
    uint32_t n_orig, n_clone, n_split, n_child;                // totals
};

__device__ __forceinline__ void copy_row(const float* __restrict__ src, float* __restrict__ dst, const FlatLayout& A,
                                         const FlatLayout& B, size_t N, size_t M, size_t i, size_t d, bool zero) {
    // one Gaussian's 59 floats from row i of an N-Gaussian flat buffer to row d of an M-Gaussian one
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[B.xyz + 3 * d + k] = zero ? 0.0f : src[A.xyz + 3 * i + k];
    for (int k = 0; k < 48; ++k) dst[B.feat + (size_t)k * M + d] = zero ? 0.0f : src[A.feat + (size_t)k * N + i];
    dst[B.opac + d] = zero ? 0.0f : src[A.opac + i];
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[B.scal + 3 * d + k] = zero ? 0.0f : src[A.scal + 3 * i + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[B.rot + 4 * d + k] = zero ? 0.0f : src[A.rot + 4 * i + k];
}

__global__ __launch_bounds__(256) void densify_apply_kernel(int N, int M, DensifyPlan pl, const float* __restrict__ p,
                                                            const float* __restrict__ m, const float* __restrict__ v,
                                                            const float* __restrict__ samples /* (2 n_split, 3) */,
                                                            float* __restrict__ pn, float* __restrict__ mn,
                                                            float* __restrict__ vn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const FlatLayout A = FlatLayout::of((size_t)N), B = FlatLayout::of((size_t)M);
    if (i == 0) { pn[B.c] = p[A.c]; mn[B.c] = m[A.c]; vn[B.c] = v[A.c]; }          // the contrast threshold rides along
    if (i >= N) return;
    if (pl.f_orig[i]) {
        const size_t d = pl.s_orig[i];
        copy_row(p, pn, A, B, N, M, i, d, false);
        copy_row(m, mn, A, B, N, M, i, d, false);
        copy_row(v, vn, A, B, N, M, i, d, false);
    }
    if (pl.f_clone[i]) {                                     // new rows start with zero Adam moments (:307-327)
        const size_t d = (size_t)pl.n_orig + pl.s_clone[i];
        copy_row(p, pn, A, B, N, M, i, d, false);
        copy_row(m, mn, A, B, N, M, i, d, true);
        copy_row(v, vn, A, B, N, M, i, d, true);
    }
    if (pl.f_child[i]) {
        const size_t j = pl.s_split[i];                      // rank among ALL selected rows: indexes the noise
        const float* q = p + A.rot + 4 * (size_t)i;
        const float inv = 1.0f / __builtin_sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);      // build_rotation
        const float r = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
        const float R[3][3] = {{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)},
                               {2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)},
                               {2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)}};
        const float* s = p + A.scal + 3 * (size_t)i;
        const float ns[3] = {logf(expf(s[0]) / 1.6f), logf(expf(s[1]) / 1.6f), logf(expf(s[2]) / 1.6f)};
#pragma unroll
        for (int copy = 0; copy < 2; ++copy) {
            const size_t d = (size_t)pl.n_orig + pl.n_clone + (size_t)copy * pl.n_child + pl.s_child[i];
            copy_row(p, pn, A, B, N, M, i, d, false);
            copy_row(m, mn, A, B, N, M, i, d, true);
            copy_row(v, vn, A, B, N, M, i, d, true);
            const float* sm = samples + 3 * ((size_t)copy * pl.n_split + j);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pn[B.xyz + 3 * d + a] = (R[a][0] * sm[0] + R[a][1] * sm[1] + R[a][2] * sm[2]) + p[A.xyz + 3 * (size_t)i + a];
                pn[B.scal + 3 * d + a] = ns[a];
            }
        }
    }
}

// scratch: 8 uint32 arrays of N (+64), the split-row list, scan scratch, 8 words for the totals
static void carve_plan(char* scratch, size_t N, uint32_t* f[4], uint32_t* s[4], int** rows, uint32_t** scan_scratch) {
    char* p = scratch;
    for (int k = 0; k < 4; ++k) f[k] = carve<uint32_t>(p, N + 64);
    for (int k = 0; k < 4; ++k) s[k] = carve<uint32_t>(p, N + 64);
    *rows = carve<int>(p, N + 64);
    *scan_scratch = carve<uint32_t>(p, scan_blocks(N) + 64);
}
size_t e3_densify_scratch_bytes(int N) {
    size_t n = N > 0 ? (size_t)N : 1;
    return 9 * align_up((n + 64) * 4, 256) + align_up((scan_blocks(n) + 64) * 4, 256) + 256;
}

int e3_densify_plan_impl(int N, const float* param, const float* grad_accum, const float* denom, float max_grad,
                         float min_opacity, float extent, float percent_dense, int size_prune, char* scratch,
                         int* counts_host, hipStream_t st) {
    counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
    if (N <= 0) return 0;
    uint32_t *f[4], *s[4], *scan_scratch;
    int* rows;
    carve_plan(scratch, (size_t)N, f, s, &rows, &scan_scratch);
    const unsigned nb = (unsigned)((N + 255) / 256);
    // thresholds are python floats in the reference, cast to float32 by the tensor comparison
    densify_flags_kernel<<<dim3(nb), dim3(256), 0, st>>>(N, param, grad_accum, denom, max_grad, min_opacity,
                                                         (float)((double)percent_dense * (double)extent),
                                                         (float)(0.1 * (double)extent), size_prune, f[0], f[1], f[2], f[3]);
    for (int k = 0; k < 4; ++k)
        if (int rc = launch_exclusive_scan_u32(f[k], s[k], (size_t)N, scan_scratch, false, st)) return rc;
    densify_split_rows_kernel<<<dim3(nb), dim3(256), 0, st>>>(N, f[2], s[2], rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e3_fail(e, "densify plan kernels");
    uint32_t last_f[4], last_s[4];
    for (int k = 0; k < 4; ++k) {
        e = hipMemcpyAsync(&last_f[k], f[k] + (N - 1), 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&last_s[k], s[k] + (N - 1), 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e3_fail(e, "densify plan read-back");
    }
    e = hipStreamSynchronize(st);                  // the step's one host wait: the new size sizes the new buffers
    if (e != hipSuccess) return e3_fail(e, "densify plan sync");
    for (int k = 0; k < 4; ++k) counts_host[k] = (int)(last_f[k] + last_s[k]);
    return 0;
}

int e3_densify_apply_impl(int N, int M, const int* counts, const float* p, const float* m, const float* v,
                          const float* samples, float* pn, float* mn, float* vn, char* scratch, hipStream_t st) {
    if (N <= 0) return 0;
    uint32_t *f[4], *s[4], *scan_scratch;
    int* rows;
    carve_plan(scratch, (size_t)N, f, s, &rows, &scan_scratch);
    DensifyPlan pl;
    pl.f_orig = f[0]; pl.f_clone = f[1]; pl.f_split = f[2]; pl.f_child = f[3];
    pl.s_orig = s[0]; pl.s_clone = s[1]; pl.s_split = s[2]; pl.s_child = s[3];
    pl.n_orig = (uint32_t)counts[0]; pl.n_clone = (uint32_t)counts[1]; pl.n_split = (uint32_t)counts[2];
    pl.n_child = (uint32_t)counts[3];
    if ((long long)counts[0] + counts[1] + 2ll * counts[3] != (long long)M)
        return e3_fail(hipErrorInvalidValue, "new size must be kept originals + kept clones + 2 x kept children");
    densify_apply_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st>>>(N, M, pl, p, m, v, samples, pn, mn, vn);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "densify_apply_kernel");
}

const int* e3_densify_split_rows(int N, char* scratch) {
    uint32_t *f[4], *s[4], *scan_scratch;
    int* rows;
    carve_plan(scratch, (size_t)(N > 0 ? N : 1), f, s, &rows, &scan_scratch);
    return rows;
}
