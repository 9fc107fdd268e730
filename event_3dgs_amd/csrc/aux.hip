// aux.hip -- small kernels around the rasteriser: distCUDA2, event loss, Adam.
#include "common.h"

int e3_fail(hipError_t e, const char* what);

// ------------------------------------------------------------------------------------ distCUDA2
// Exact mean squared distance to the 3 nearest other points (scene/gaussian_model.py:134,
// SURVEY Appendix C).  Uniform-grid search: points are binned into cubic cells (radix sort by
// cell id with the same sort kernels as the rasteriser), then each point scans cell shells of
// growing Chebyshev radius until the shell's lower distance bound exceeds its current 3rd best.
struct KnnGrid {
    float minx, miny, minz, inv_h, h;
    int nx, ny, nz;
};

__global__ __launch_bounds__(256) void knn_bbox_kernel(int P, const float* __restrict__ pts, float* __restrict__ bbox) {
    // bbox: [minx,miny,minz,maxx,maxy,maxz] as ordered-int atomics
    __shared__ float smin[3][4], smax[3][4];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { smin[a][wave] = mn[a]; smax[a][wave] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        int a = threadIdx.x;
        float lo = fminf(fminf(smin[a][0], smin[a][1]), fminf(smin[a][2], smin[a][3]));
        float hi = fmaxf(fmaxf(smax[a][0], smax[a][1]), fmaxf(smax[a][2], smax[a][3]));
        // monotone float->int mapping for atomicMin/Max
        auto enc = [](float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; };
        atomicMin(reinterpret_cast<int*>(bbox) + a, enc(lo));
        atomicMax(reinterpret_cast<int*>(bbox) + 3 + a, enc(hi));
    }
}

__device__ __forceinline__ float knn_dec(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__device__ __forceinline__ KnnGrid knn_grid(const float* bbox, int P) {
    const int* b = reinterpret_cast<const int*>(bbox);
    KnnGrid g;
    g.minx = knn_dec(b[0]); g.miny = knn_dec(b[1]); g.minz = knn_dec(b[2]);
    float ex = knn_dec(b[3]) - g.minx, ey = knn_dec(b[4]) - g.miny, ez = knn_dec(b[5]) - g.minz;
    float ext = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-20f));
    // aim at ~4 points per cell, at most 128 cells per axis (21 bits of cell id)
    float cells = fminf(128.0f, fmaxf(1.0f, cbrtf((float)P / 4.0f)));
    g.h = ext / cells * 1.0001f;
    g.inv_h = 1.0f / g.h;
    g.nx = min(128, (int)(ex * g.inv_h) + 1);
    g.ny = min(128, (int)(ey * g.inv_h) + 1);
    g.nz = min(128, (int)(ez * g.inv_h) + 1);
    return g;
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = min(g.nx - 1, max(0, (int)((x - g.minx) * g.inv_h)));
    cy = min(g.ny - 1, max(0, (int)((y - g.miny) * g.inv_h)));
    cz = min(g.nz - 1, max(0, (int)((z - g.minz) * g.inv_h)));
}

__global__ __launch_bounds__(256) void knn_cellid_kernel(int P, const float* __restrict__ pts,
                                                         const float* __restrict__ bbox, uint32_t* __restrict__ cell,
                                                         uint32_t* __restrict__ idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    KnnGrid g = knn_grid(bbox, P);
    int cx, cy, cz;
    knn_cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
    cell[i] = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void knn_cell_ranges_kernel(int P, const uint32_t* __restrict__ cell_sorted,
                                                              const uint32_t* __restrict__ idx_sorted,
                                                              const float* __restrict__ pts,
                                                              uint32_t* __restrict__ cstart, uint32_t* __restrict__ cend,
                                                              float4* __restrict__ pts_sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t c = cell_sorted[i];
    if (i == 0 || cell_sorted[i - 1] != c) cstart[c] = (uint32_t)i;
    if (i == P - 1 || cell_sorted[i + 1] != c) cend[c] = (uint32_t)i + 1;
    uint32_t j = idx_sorted[i];
    pts_sorted[i] = make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], __uint_as_float(j));
}

__global__ __launch_bounds__(256) void knn_search_kernel(int P, const float4* __restrict__ pts_sorted,
                                                         const float* __restrict__ bbox,
                                                         const uint32_t* __restrict__ cstart,
                                                         const uint32_t* __restrict__ cend, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    KnnGrid g = knn_grid(bbox, P);
    float4 p = pts_sorted[i];
    int cx, cy, cz;
    knn_cell_of(g, p.x, p.y, p.z, cx, cy, cz);
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    const int maxr = max(g.nx, max(g.ny, g.nz));
    // distance from p to the faces of its own cell -> lower bound for shell r is (r-1)*h + that margin
    float fx = (p.x - g.minx) - cx * g.h, fy = (p.y - g.miny) - cy * g.h, fz = (p.z - g.minz) - cz * g.h;
    float margin = fminf(fminf(fminf(fx, g.h - fx), fminf(fy, g.h - fy)), fminf(fz, g.h - fz));
    margin = fmaxf(margin, 0.0f);
    for (int r = 0; r <= maxr; ++r) {
        if (r > 0) {
            float lb = (float)(r - 1) * g.h + margin;
            if (lb * lb > b2) break;
        }
        for (int dz = -r; dz <= r; ++dz) {
            int z = cz + dz;
            if (z < 0 || z >= g.nz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                int y = cy + dy;
                if (y < 0 || y >= g.ny) continue;
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                const int step = face ? 1 : 2 * r;   // interior rows: only the two x extremes belong to the shell
                for (int dx = -r; dx <= r; dx += (step > 0 ? step : 1)) {
                    int x = cx + dx;
                    if (x < 0 || x >= g.nx) continue;
                    uint32_t c = (uint32_t)((z * g.ny + y) * g.nx + x);
                    uint32_t s = cstart[c], e = cend[c];
                    for (uint32_t k = s; k < e; ++k) {
                        if ((int)k == i) continue;
                        float4 q = pts_sorted[k];
                        float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
                        float d = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (d < b2) {
                            if (d < b1) {
                                b2 = b1;
                                if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
                            } else b2 = d;
                        }
                    }
                }
            }
        }
    }
    float s = 0.0f; int n = 0;
    if (b0 != INFINITY) { s += b0; ++n; }
    if (b1 != INFINITY) { s += b1; ++n; }
    if (b2 != INFINITY) { s += b2; ++n; }
    float r = n == 3 ? s / 3.0f : (n ? s / (float)n : 0.0f);
    out[__float_as_uint(p.w)] = r;
}

constexpr size_t KNN_MAX_CELLS = 128ull * 128ull * 128ull;

size_t e3_knn_scratch_bytes(int P) {
    size_t n = P > 0 ? (size_t)P : 1;
    char* p = nullptr;
    carve<float>(p, 64);
    carve<uint32_t>(p, n); carve<uint32_t>(p, n); carve<uint32_t>(p, n); carve<uint32_t>(p, n);
    carve<uint32_t>(p, sort_scratch_words(n));
    carve<uint32_t>(p, KNN_MAX_CELLS); carve<uint32_t>(p, KNN_MAX_CELLS);
    carve<float4>(p, n);
    return (size_t)p + 256;
}

int e3_knn_impl(int P, const float* pts, float* out, char* scratch, hipStream_t s) {
    if (P <= 0) return 0;
    size_t n = (size_t)P;
    char* p = scratch;
    float* bbox = carve<float>(p, 64);
    uint32_t* c0 = carve<uint32_t>(p, n); uint32_t* c1 = carve<uint32_t>(p, n);
    uint32_t* i0 = carve<uint32_t>(p, n); uint32_t* i1 = carve<uint32_t>(p, n);
    uint32_t* sscr = carve<uint32_t>(p, sort_scratch_words(n));
    uint32_t* cstart = carve<uint32_t>(p, KNN_MAX_CELLS); uint32_t* cend = carve<uint32_t>(p, KNN_MAX_CELLS);
    float4* ps = carve<float4>(p, n);
    // init bbox: min slots = +inf encoding, max slots = -inf encoding
    int init[6] = {0x7F800000, 0x7F800000, 0x7F800000, (int)0xFF800000 ^ 0x7FFFFFFF, (int)0xFF800000 ^ 0x7FFFFFFF,
                   (int)0xFF800000 ^ 0x7FFFFFFF};
    hipError_t e = hipMemcpyAsync(bbox, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e3_fail(e, "knn bbox init");
    e = hipStreamSynchronize(s);   // `init` is a stack buffer
    if (e != hipSuccess) return e3_fail(e, "knn bbox init sync");
    unsigned pb = (unsigned)((P + 255) / 256);
    knn_bbox_kernel<<<dim3(pb < 1024 ? pb : 1024), dim3(256), 0, s>>>(P, pts, bbox);
    knn_cellid_kernel<<<dim3(pb), dim3(256), 0, s>>>(P, pts, bbox, c0, i0);
    uint32_t *cs, *is;
    if (int rc = launch_radix_sort_pairs(c0, c1, i0, i1, n, 21, sscr, &cs, &is, s)) return rc;
    e = hipMemsetAsync(cstart, 0, KNN_MAX_CELLS * 4, s);
    if (e != hipSuccess) return e3_fail(e, "knn memset");
    e = hipMemsetAsync(cend, 0, KNN_MAX_CELLS * 4, s);
    if (e != hipSuccess) return e3_fail(e, "knn memset");
    knn_cell_ranges_kernel<<<dim3(pb), dim3(256), 0, s>>>(P, cs, is, pts, cstart, cend, ps);
    knn_search_kernel<<<dim3(pb), dim3(256), 0, s>>>(P, ps, bbox, cstart, cend, out);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "knn kernels");
}

// ------------------------------------------------------------------------------------ event loss
// train.py:165-203 fused into a reduction pass and a gradient pass.
constexpr int EV_THREADS = 256;
constexpr int EV_NSUM = 5;   // sum|D-D*|, count(D*!=0), sum sign*D, sum|image-gt_int|, sum|image-gt_blur|

__device__ __forceinline__ float lum3(const float* __restrict__ img, size_t HW, size_t p) {
    return FMA(0.1804f, img[2 * HW + p], FMA(0.35758f, img[HW + p], 0.4124f * img[p]));
}

__global__ __launch_bounds__(EV_THREADS) void event_reduce_kernel(
    size_t HW, const float* __restrict__ image, const float* __restrict__ now, const float* __restrict__ next,
    const float* __restrict__ gt_int, const float* __restrict__ gt_now, const float* __restrict__ gt_next,
    const float* __restrict__ gt_blur, const float* __restrict__ c_ptr, float gt_c, double* __restrict__ partials) {
    __shared__ double sred[EV_NSUM][EV_THREADS / WAVE];
    const float c = c_ptr[0];
    double acc[EV_NSUM] = {0, 0, 0, 0, 0};
    // one pixel: the per-pixel terms, accumulated in the order of the pixel index
    auto pixel = [&](const float nx[3], const float nw[3], const float gx[3], const float gw[3], const float im[3],
                     const float gi[3], const float gb[3]) {
        // logf, not the fast __logf: the mask rho = count(D* != 0) and sign(D - D*) hinge on exact zeros / ties, and the
        // reference's torch.log is the 1-ulp device logf (equal luminances give exactly D* = 0 with any deterministic log)
        auto lum = [](const float v[3]) { return FMA(0.1804f, v[2], FMA(0.35758f, v[1], 0.4124f * v[0])); };
        float D = (logf(lum(nx) + 1e-8f) - logf(lum(nw) + 1e-8f)) / c;
        float Dg = (logf(lum(gx) + 1e-8f) - logf(lum(gw) + 1e-8f)) / gt_c;
        float e = D - Dg;
        acc[0] += fabsf(e);
        acc[1] += (Dg != 0.0f) ? 1.0 : 0.0;
        acc[2] += (double)((e > 0.0f) - (e < 0.0f)) * (double)D;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            acc[3] += fabsf(im[ch] - gi[ch]);
            if (gt_blur) acc[4] += fabsf(im[ch] - gb[ch]);
        }
    };
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(now) |
                                        reinterpret_cast<uintptr_t>(next) | reinterpret_cast<uintptr_t>(gt_int) |
                                        reinterpret_cast<uintptr_t>(gt_now) | reinterpret_cast<uintptr_t>(gt_next) |
                                        reinterpret_cast<uintptr_t>(gt_blur)) & 15) == 0;
    if (vec) {
        // four pixels per thread and trip, every plane read with 16-byte loads (a quarter of the load instructions)
        const size_t HW4 = HW >> 2;
        for (size_t q = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; q < HW4; q += (size_t)gridDim.x * EV_THREADS) {
            float4 vn[3], vw[3], vgx[3], vgw[3], vim[3], vgi[3], vgb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                vn[ch] = reinterpret_cast<const float4*>(next + ch * HW)[q];
                vw[ch] = reinterpret_cast<const float4*>(now + ch * HW)[q];
                vgx[ch] = reinterpret_cast<const float4*>(gt_next + ch * HW)[q];
                vgw[ch] = reinterpret_cast<const float4*>(gt_now + ch * HW)[q];
                vim[ch] = reinterpret_cast<const float4*>(image + ch * HW)[q];
                vgi[ch] = reinterpret_cast<const float4*>(gt_int + ch * HW)[q];
                vgb[ch] = gt_blur ? reinterpret_cast<const float4*>(gt_blur + ch * HW)[q] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                auto comp = [u](const float4& v) { return u == 0 ? v.x : (u == 1 ? v.y : (u == 2 ? v.z : v.w)); };
                const float nx[3] = {comp(vn[0]), comp(vn[1]), comp(vn[2])}, nw[3] = {comp(vw[0]), comp(vw[1]), comp(vw[2])};
                const float gx[3] = {comp(vgx[0]), comp(vgx[1]), comp(vgx[2])}, gw[3] = {comp(vgw[0]), comp(vgw[1]), comp(vgw[2])};
                const float im[3] = {comp(vim[0]), comp(vim[1]), comp(vim[2])}, gi[3] = {comp(vgi[0]), comp(vgi[1]), comp(vgi[2])};
                const float gb[3] = {comp(vgb[0]), comp(vgb[1]), comp(vgb[2])};
                pixel(nx, nw, gx, gw, im, gi, gb);
            }
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; p < HW; p += (size_t)gridDim.x * EV_THREADS) {
            float nx[3], nw[3], gx[3], gw[3], im[3], gi[3], gb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                nx[ch] = next[ch * HW + p]; nw[ch] = now[ch * HW + p]; gx[ch] = gt_next[ch * HW + p];
                gw[ch] = gt_now[ch * HW + p]; im[ch] = image[ch * HW + p]; gi[ch] = gt_int[ch * HW + p];
                gb[ch] = gt_blur ? gt_blur[ch * HW + p] : 0.0f;
            }
            pixel(nx, nw, gx, gw, im, gi, gb);
        }
    }
#pragma unroll
    for (int k = 0; k < EV_NSUM; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        if ((threadIdx.x & 63) == 0) sred[k][threadIdx.x >> 6] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < EV_NSUM) {
        double s = 0;
        for (int w = 0; w < EV_THREADS / WAVE; ++w) s += sred[threadIdx.x][w];
        partials[(size_t)blockIdx.x * EV_NSUM + threadIdx.x] = s;
    }
}

// scalars (float[8]): 0 loss, 1 dL/dc, 2 rho, 3 L1 event, 4 L1 intensity, 5 L1 blur, 6 kE, 7 kI
__global__ __launch_bounds__(EV_THREADS) void event_finalize_kernel(int nblocks, size_t HW, const double* __restrict__ partials,
                                                                    const float* __restrict__ c_ptr, int has_blur,
                                                                    float* __restrict__ scalars, float* __restrict__ dc_out,
                                                                    double* __restrict__ nz_out) {
    // one workgroup (a single wave walked the ~2 000 partial rows in 12 us of dependent loads); fixed order: deterministic
    __shared__ double sfin[EV_NSUM][EV_THREADS / WAVE];
    double acc[EV_NSUM] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += EV_THREADS)
#pragma unroll
        for (int k = 0; k < EV_NSUM; ++k) acc[k] += partials[(size_t)b * EV_NSUM + k];
#pragma unroll
    for (int k = 0; k < EV_NSUM; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        if ((threadIdx.x & 63) == 0) sfin[k][threadIdx.x >> 6] = acc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EV_NSUM; ++k) {
        acc[k] = 0;
        for (int w = 0; w < EV_THREADS / WAVE; ++w) acc[k] += sfin[k][w];
    }
    if (threadIdx.x == 0) {
        const double n = (double)HW, c = (double)c_ptr[0];
        double L1E = acc[0] / n, rho = acc[1] / n, L1I = acc[3] / (3.0 * n), L1B = acc[4] / (3.0 * n);
        double loss = 0.9 * L1E * rho + 0.1 * L1I * (1.0 - rho);
        double outer = 1.0;
        if (has_blur) { loss = 0.5 * loss + 0.5 * L1B; outer = 0.5; }
        scalars[0] = (float)loss;
        scalars[1] = (float)(-outer * 0.9 * rho * (acc[2] / n) / c);
        if (dc_out) *dc_out = scalars[1];      // (e.g. the threshold's slot of a flat gradient buffer: no copy kernel)
        scalars[2] = (float)rho; scalars[3] = (float)L1E; scalars[4] = (float)L1I; scalars[5] = (float)L1B;
        scalars[6] = (float)(outer * 0.9 * rho / n);
        scalars[7] = (float)(outer * 0.1 * (1.0 - rho) / (3.0 * n));
        if (nz_out) *nz_out = acc[1];          // count(D* != 0): a property of the ground-truth pair (event_fused_kernel)
    }
}

__global__ __launch_bounds__(EV_THREADS) void event_grad_kernel(
    size_t HW, const float* __restrict__ image, const float* __restrict__ now, const float* __restrict__ next,
    const float* __restrict__ gt_int, const float* __restrict__ gt_now, const float* __restrict__ gt_next,
    const float* __restrict__ gt_blur, const float* __restrict__ c_ptr, float gt_c,
    const float* __restrict__ scalars, float* d_image, float* d_now /* may alias d_image: the SUM is stored */,
    float* __restrict__ d_next, int rank1 /* d_next (and d_now when it is a render of its own) as scalar fields */) {
    // d_now == d_image: render #1 and render #2 are the same render (the reference's event camera `index` carries the
    // pose of its training camera `index`: scene/dataset_readers.py:157 reads both with the same extrinsics), so the
    // caller rendered it once and wants dL/d(that image) = intensity part + contrast part
    const bool shared = d_now == d_image;
    // image == now with separate outputs: d_now receives the TOTAL gradient of that render (what its backward needs),
    // d_image the intensity part alone (what the densification statistics need: e3dgs_rasterize_backward_multi_stats)
    const bool total_in_now = !shared && image == now;
    const float c = c_ptr[0];
    const float kE = scalars[6], kI = scalars[7];
    const float kB = 0.5f / (3.0f * (float)HW);
    const float wch[3] = {0.4124f, 0.35758f, 0.1804f};
    // one pixel: inputs per channel -> the three gradient triples
    auto pixel = [&](const float nx[3], const float nw[3], const float gx[3], const float gw[3], const float im[3],
                     const float gi[3], const float gb[3], float dn[3], float dw[3], float di[3], float& sn, float& sw) {
        auto lum = [](const float v[3]) { return FMA(0.1804f, v[2], FMA(0.35758f, v[1], 0.4124f * v[0])); };
        float yn = lum(nx) + 1e-8f, yo = lum(nw) + 1e-8f;
        float D = (logf(yn) - logf(yo)) / c;
        float Dg = (logf(lum(gx) + 1e-8f) - logf(lum(gw) + 1e-8f)) / gt_c;
        float e = D - Dg;
        float k = kE * (float)((e > 0.0f) - (e < 0.0f)) / c;
        sn = k / yn; sw = -(k / yo);         // rank-1 form: dL/dC = s * (0.4124, 0.35758, 0.1804)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            dn[ch] = k * wch[ch] / yn;
            dw[ch] = -(k * wch[ch]) / yo;
            float ei = im[ch] - gi[ch];
            float g = kI * (float)((ei > 0.0f) - (ei < 0.0f));
            if (gt_blur) {
                float eb = im[ch] - gb[ch];
                g += kB * (float)((eb > 0.0f) - (eb < 0.0f));
            }
            di[ch] = g;
        }
    };
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(now) |
                                        reinterpret_cast<uintptr_t>(next) | reinterpret_cast<uintptr_t>(gt_int) |
                                        reinterpret_cast<uintptr_t>(gt_now) | reinterpret_cast<uintptr_t>(gt_next) |
                                        reinterpret_cast<uintptr_t>(gt_blur)) & 15) == 0;
    // (the pixel -> thread mapping, i.e. the accumulation order of the partial sums, follows from the INPUTS alone, as in
    // event_reduce_kernel: the loss scalars of this path stay bit-identical to the three-launch path whatever the alignment
    // of the outputs; misaligned outputs only turn the 16-byte stores into dword stores)
    const bool vec_out = ((reinterpret_cast<uintptr_t>(d_image) | reinterpret_cast<uintptr_t>(d_now) |
                           reinterpret_cast<uintptr_t>(d_next)) & 15) == 0;
    auto store4 = [vec_out](float* plane, size_t q, float a, float b, float c4, float d) {
        if (vec_out) reinterpret_cast<float4*>(plane)[q] = make_float4(a, b, c4, d);
        else { plane[4 * q] = a; plane[4 * q + 1] = b; plane[4 * q + 2] = c4; plane[4 * q + 3] = d; }
    };
    if (vec) {
        const size_t HW4 = HW >> 2;                          // four pixels per thread and trip, 16-byte loads and stores
        for (size_t q = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; q < HW4; q += (size_t)gridDim.x * EV_THREADS) {
            float4 vn[3], vw[3], vgx[3], vgw[3], vim[3], vgi[3], vgb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                vn[ch] = reinterpret_cast<const float4*>(next + ch * HW)[q];
                vw[ch] = reinterpret_cast<const float4*>(now + ch * HW)[q];
                vgx[ch] = reinterpret_cast<const float4*>(gt_next + ch * HW)[q];
                vgw[ch] = reinterpret_cast<const float4*>(gt_now + ch * HW)[q];
                vim[ch] = reinterpret_cast<const float4*>(image + ch * HW)[q];
                vgi[ch] = reinterpret_cast<const float4*>(gt_int + ch * HW)[q];
                vgb[ch] = gt_blur ? reinterpret_cast<const float4*>(gt_blur + ch * HW)[q] : make_float4(0, 0, 0, 0);
            }
            float on[3][4], ow[3][4], oi[3][4], osn[4], osw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                auto comp = [u](const float4& v) { return u == 0 ? v.x : (u == 1 ? v.y : (u == 2 ? v.z : v.w)); };
                const float nx[3] = {comp(vn[0]), comp(vn[1]), comp(vn[2])}, nw[3] = {comp(vw[0]), comp(vw[1]), comp(vw[2])};
                const float gx[3] = {comp(vgx[0]), comp(vgx[1]), comp(vgx[2])}, gw[3] = {comp(vgw[0]), comp(vgw[1]), comp(vgw[2])};
                const float im[3] = {comp(vim[0]), comp(vim[1]), comp(vim[2])}, gi[3] = {comp(vgi[0]), comp(vgi[1]), comp(vgi[2])};
                const float gb[3] = {comp(vgb[0]), comp(vgb[1]), comp(vgb[2])};
                float dn[3], dw[3], di[3];
                pixel(nx, nw, gx, gw, im, gi, gb, dn, dw, di, osn[u], osw[u]);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { on[ch][u] = dn[ch]; ow[ch][u] = dw[ch]; oi[ch][u] = di[ch]; }
            }
            if (rank1) {                     // the contrast renders' gradients as scalar fields (plane 0 only)
                store4(d_next, q, osn[0], osn[1], osn[2], osn[3]);
                if (!shared && !total_in_now) store4(d_now, q, osw[0], osw[1], osw[2], osw[3]);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (!rank1) store4(d_next + ch * HW, q, on[ch][0], on[ch][1], on[ch][2], on[ch][3]);
                if (shared) {
                    store4(d_image + ch * HW, q, oi[ch][0] + ow[ch][0], oi[ch][1] + ow[ch][1], oi[ch][2] + ow[ch][2],
                           oi[ch][3] + ow[ch][3]);
                } else {
                    if (total_in_now) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) ow[ch][u] = oi[ch][u] + ow[ch][u];
                    }
                    if (!rank1 || total_in_now) store4(d_now + ch * HW, q, ow[ch][0], ow[ch][1], ow[ch][2], ow[ch][3]);
                    store4(d_image + ch * HW, q, oi[ch][0], oi[ch][1], oi[ch][2], oi[ch][3]);
                }
            }
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; p < HW; p += (size_t)gridDim.x * EV_THREADS) {
            float nx[3], nw[3], gx[3], gw[3], im[3], gi[3], gb[3], dn[3], dw[3], di[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                nx[ch] = next[ch * HW + p]; nw[ch] = now[ch * HW + p]; gx[ch] = gt_next[ch * HW + p];
                gw[ch] = gt_now[ch * HW + p]; im[ch] = image[ch * HW + p]; gi[ch] = gt_int[ch * HW + p];
                gb[ch] = gt_blur ? gt_blur[ch * HW + p] : 0.0f;
            }
            float sn, sw;
            pixel(nx, nw, gx, gw, im, gi, gb, dn, dw, di, sn, sw);
            if (rank1) {
                d_next[p] = sn;
                if (!shared && !total_in_now) d_now[p] = sw;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (!rank1) d_next[ch * HW + p] = dn[ch];
                if (shared) d_image[ch * HW + p] = di[ch] + dw[ch];
                else {
                    if (!rank1 || total_in_now) d_now[ch * HW + p] = total_in_now ? di[ch] + dw[ch] : dw[ch];
                    d_image[ch * HW + p] = di[ch];
                }
            }
        }
    }
}


// One pass instead of two.  The gradient pass needs two numbers of the reduction, kE and kI -- and both depend on rho =
// count(D* != 0) / n alone, which is a property of the GROUND-TRUTH pair (D* is the contrast of gt_now / gt_next): a caller
// that has seen the pair before hands the count back (`nz_count`, written by event_finalize_kernel the first time), and the
// partial sums of the loss and the three pixel gradients come out of ONE sweep over the seven images -- the planes are
// read once, the four logarithms evaluated once.  Same per-pixel arithmetic, same accumulation order, same block / thread
// -> pixel mapping as event_reduce_kernel + event_grad_kernel: bit-identical results.
__global__ __launch_bounds__(EV_THREADS) void event_fused_kernel(
    size_t HW, const float* __restrict__ image, const float* __restrict__ now, const float* __restrict__ next,
    const float* __restrict__ gt_int, const float* __restrict__ gt_now, const float* __restrict__ gt_next,
    const float* __restrict__ gt_blur, const float* __restrict__ c_ptr, float gt_c, const double* __restrict__ nz_count,
    double* __restrict__ partials, float* d_image, float* d_now /* may alias d_image: the SUM is stored */,
    float* __restrict__ d_next, int rank1) {
    __shared__ double sred[EV_NSUM][EV_THREADS / WAVE];
    const bool shared = d_now == d_image;
    const bool total_in_now = !shared && image == now;
    const float c = c_ptr[0];
    float kE, kI;
    {   // event_finalize_kernel's expressions on the handed-back count
        const double n = (double)HW, rho = nz_count[0] / n, outer = gt_blur ? 0.5 : 1.0;
        kE = (float)(outer * 0.9 * rho / n);
        kI = (float)(outer * 0.1 * (1.0 - rho) / (3.0 * n));
    }
    const float kB = 0.5f / (3.0f * (float)HW);
    const float wch[3] = {0.4124f, 0.35758f, 0.1804f};
    double acc[EV_NSUM] = {0, 0, 0, 0, 0};
    auto pixel = [&](const float nx[3], const float nw[3], const float gx[3], const float gw[3], const float im[3],
                     const float gi[3], const float gb[3], float dn[3], float dw[3], float di[3], float& sn, float& sw) {
        auto lum = [](const float v[3]) { return FMA(0.1804f, v[2], FMA(0.35758f, v[1], 0.4124f * v[0])); };
        const float yn = lum(nx) + 1e-8f, yo = lum(nw) + 1e-8f;
        const float D = (logf(yn) - logf(yo)) / c;
        const float Dg = (logf(lum(gx) + 1e-8f) - logf(lum(gw) + 1e-8f)) / gt_c;
        const float e = D - Dg;
        const float sg = (float)((e > 0.0f) - (e < 0.0f));
        acc[0] += fabsf(e);
        acc[1] += (Dg != 0.0f) ? 1.0 : 0.0;
        acc[2] += (double)((e > 0.0f) - (e < 0.0f)) * (double)D;
        const float k = kE * sg / c;
        sn = k / yn; sw = -(k / yo);         // rank-1 form: dL/dC = s * (0.4124, 0.35758, 0.1804)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            acc[3] += fabsf(im[ch] - gi[ch]);
            if (gt_blur) acc[4] += fabsf(im[ch] - gb[ch]);
            dn[ch] = k * wch[ch] / yn;
            dw[ch] = -(k * wch[ch]) / yo;
            const float ei = im[ch] - gi[ch];
            float g = kI * (float)((ei > 0.0f) - (ei < 0.0f));
            if (gt_blur) {
                const float eb = im[ch] - gb[ch];
                g += kB * (float)((eb > 0.0f) - (eb < 0.0f));
            }
            di[ch] = g;
        }
    };
    const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(now) |
                                        reinterpret_cast<uintptr_t>(next) | reinterpret_cast<uintptr_t>(gt_int) |
                                        reinterpret_cast<uintptr_t>(gt_now) | reinterpret_cast<uintptr_t>(gt_next) |
                                        reinterpret_cast<uintptr_t>(gt_blur)) & 15) == 0;
    // (the pixel -> thread mapping, i.e. the accumulation order of the partial sums, follows from the INPUTS alone, as in
    // event_reduce_kernel: the loss scalars of this path stay bit-identical to the three-launch path whatever the alignment
    // of the outputs; misaligned outputs only turn the 16-byte stores into dword stores)
    const bool vec_out = ((reinterpret_cast<uintptr_t>(d_image) | reinterpret_cast<uintptr_t>(d_now) |
                           reinterpret_cast<uintptr_t>(d_next)) & 15) == 0;
    auto store4 = [vec_out](float* plane, size_t q, float a, float b, float c4, float d) {
        if (vec_out) reinterpret_cast<float4*>(plane)[q] = make_float4(a, b, c4, d);
        else { plane[4 * q] = a; plane[4 * q + 1] = b; plane[4 * q + 2] = c4; plane[4 * q + 3] = d; }
    };
    if (vec) {
        const size_t HW4 = HW >> 2;
        for (size_t q = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; q < HW4; q += (size_t)gridDim.x * EV_THREADS) {
            float4 vn[3], vw[3], vgx[3], vgw[3], vim[3], vgi[3], vgb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                vn[ch] = reinterpret_cast<const float4*>(next + ch * HW)[q];
                vw[ch] = reinterpret_cast<const float4*>(now + ch * HW)[q];
                vgx[ch] = reinterpret_cast<const float4*>(gt_next + ch * HW)[q];
                vgw[ch] = reinterpret_cast<const float4*>(gt_now + ch * HW)[q];
                vim[ch] = reinterpret_cast<const float4*>(image + ch * HW)[q];
                vgi[ch] = reinterpret_cast<const float4*>(gt_int + ch * HW)[q];
                vgb[ch] = gt_blur ? reinterpret_cast<const float4*>(gt_blur + ch * HW)[q] : make_float4(0, 0, 0, 0);
            }
            float on[3][4], ow[3][4], oi[3][4], osn[4], osw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                auto comp = [u](const float4& v) { return u == 0 ? v.x : (u == 1 ? v.y : (u == 2 ? v.z : v.w)); };
                const float nx[3] = {comp(vn[0]), comp(vn[1]), comp(vn[2])}, nw[3] = {comp(vw[0]), comp(vw[1]), comp(vw[2])};
                const float gx[3] = {comp(vgx[0]), comp(vgx[1]), comp(vgx[2])}, gw[3] = {comp(vgw[0]), comp(vgw[1]), comp(vgw[2])};
                const float im[3] = {comp(vim[0]), comp(vim[1]), comp(vim[2])}, gi[3] = {comp(vgi[0]), comp(vgi[1]), comp(vgi[2])};
                const float gb[3] = {comp(vgb[0]), comp(vgb[1]), comp(vgb[2])};
                float dn[3], dw[3], di[3];
                pixel(nx, nw, gx, gw, im, gi, gb, dn, dw, di, osn[u], osw[u]);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { on[ch][u] = dn[ch]; ow[ch][u] = dw[ch]; oi[ch][u] = di[ch]; }
            }
            if (rank1) {                     // the contrast renders' gradients as scalar fields (plane 0 only)
                store4(d_next, q, osn[0], osn[1], osn[2], osn[3]);
                if (!shared && !total_in_now) store4(d_now, q, osw[0], osw[1], osw[2], osw[3]);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (!rank1) store4(d_next + ch * HW, q, on[ch][0], on[ch][1], on[ch][2], on[ch][3]);
                if (shared) {
                    store4(d_image + ch * HW, q, oi[ch][0] + ow[ch][0], oi[ch][1] + ow[ch][1], oi[ch][2] + ow[ch][2],
                           oi[ch][3] + ow[ch][3]);
                } else {
                    if (total_in_now) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) ow[ch][u] = oi[ch][u] + ow[ch][u];
                    }
                    if (!rank1 || total_in_now) store4(d_now + ch * HW, q, ow[ch][0], ow[ch][1], ow[ch][2], ow[ch][3]);
                    store4(d_image + ch * HW, q, oi[ch][0], oi[ch][1], oi[ch][2], oi[ch][3]);
                }
            }
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * EV_THREADS + threadIdx.x; p < HW; p += (size_t)gridDim.x * EV_THREADS) {
            float nx[3], nw[3], gx[3], gw[3], im[3], gi[3], gb[3], dn[3], dw[3], di[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                nx[ch] = next[ch * HW + p]; nw[ch] = now[ch * HW + p]; gx[ch] = gt_next[ch * HW + p];
                gw[ch] = gt_now[ch * HW + p]; im[ch] = image[ch * HW + p]; gi[ch] = gt_int[ch * HW + p];
                gb[ch] = gt_blur ? gt_blur[ch * HW + p] : 0.0f;
            }
            float sn, sw;
            pixel(nx, nw, gx, gw, im, gi, gb, dn, dw, di, sn, sw);
            if (rank1) {
                d_next[p] = sn;
                if (!shared && !total_in_now) d_now[p] = sw;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (!rank1) d_next[ch * HW + p] = dn[ch];
                if (shared) d_image[ch * HW + p] = di[ch] + dw[ch];
                else {
                    if (!rank1 || total_in_now) d_now[ch * HW + p] = total_in_now ? di[ch] + dw[ch] : dw[ch];
                    d_image[ch * HW + p] = di[ch];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < EV_NSUM; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        if ((threadIdx.x & 63) == 0) sred[k][threadIdx.x >> 6] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < EV_NSUM) {
        double s = 0;
        for (int w = 0; w < EV_THREADS / WAVE; ++w) s += sred[threadIdx.x][w];
        partials[(size_t)blockIdx.x * EV_NSUM + threadIdx.x] = s;
    }
}

static inline int ev_blocks(size_t HW) {
    size_t b = (HW + EV_THREADS - 1) / EV_THREADS;
    return (int)(b < 2048 ? b : 2048);
}
size_t e3_event_scratch_bytes(int W, int H) { return (size_t)ev_blocks((size_t)W * H) * EV_NSUM * sizeof(double) + 256; }

int e3_event_loss_impl(int W, int H, const float* image, const float* now, const float* next, const float* gt_int,
                       const float* gt_now, const float* gt_next, const float* gt_blur, const float* c, float gt_c,
                       float* d_image, float* d_now, float* d_next, float* scalars, char* scratch, hipStream_t s,
                       float* dc_out, double* nz_count, int nz_valid, int rank1) {
    size_t HW = (size_t)W * H;
    if (HW == 0) return 0;
    int nb = ev_blocks(HW);
    double* partials = reinterpret_cast<double*>(scratch);
    if (nz_count && nz_valid) {
        // the ground-truth pair's count is known: one sweep (partial sums + the three pixel gradients), then the scalars
        event_fused_kernel<<<dim3(nb), dim3(EV_THREADS), 0, s>>>(HW, image, now, next, gt_int, gt_now, gt_next, gt_blur, c,
                                                                 gt_c, nz_count, partials, d_image, d_now, d_next, rank1);
        event_finalize_kernel<<<dim3(1), dim3(EV_THREADS), 0, s>>>(nb, HW, partials, c, gt_blur != nullptr, scalars, dc_out,
                                                                   nullptr);
    } else {
        event_reduce_kernel<<<dim3(nb), dim3(EV_THREADS), 0, s>>>(HW, image, now, next, gt_int, gt_now, gt_next, gt_blur, c,
                                                                  gt_c, partials);
        event_finalize_kernel<<<dim3(1), dim3(EV_THREADS), 0, s>>>(nb, HW, partials, c, gt_blur != nullptr, scalars, dc_out,
                                                                   nz_count);
        event_grad_kernel<<<dim3(nb), dim3(EV_THREADS), 0, s>>>(HW, image, now, next, gt_int, gt_now, gt_next, gt_blur, c,
                                                                gt_c, scalars, d_image, d_now, d_next, rank1);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "event loss kernels");
}

// ------------------------------------------------------------------------------------ densification statistics
// train.py:317-320 + scene/gaussian_model.py:405-407 for the Gaussians the render saw (radii > 0):
// max_radii2D = max(., radii); xyz_gradient_accum += |viewspace gradient (x, y)|; denom += 1.
__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const float* __restrict__ viewspace_grad,
                                                            const int* __restrict__ radii, float* __restrict__ max_radii2D,
                                                            float* __restrict__ grad_accum, float* __restrict__ denom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = viewspace_grad[3 * (size_t)i], gy = viewspace_grad[3 * (size_t)i + 1];
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    grad_accum[i] += __builtin_sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}
int e3_densify_stats_impl(int P, const float* viewspace_grad, const int* radii, float* max_radii2D, float* grad_accum,
                          float* denom, hipStream_t s) {
    if (P <= 0) return 0;
    densify_stats_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, viewspace_grad, radii, max_radii2D, grad_accum, denom);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "densify_stats_kernel");
}

// ------------------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float step_size,
                                                   float b1, float b2, float bc2_sqrt, float eps, float step_size_b,
                                                   int period, int split, float omb1, float omb2 /* 1 - beta, as torch's fp32 */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const float step_a = step_size;
    for (; i < n; i += stride) {
        if (period > 0) step_size = ((int)(i % (size_t)period) < split) ? step_a : step_size_b;
        float gi = g[i];
        float mi = m[i] + omb1 * (gi - m[i]);
        float vi = v[i] * b2 + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = __builtin_sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

int e3_adam_impl(size_t n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                 int step, float lr_b, int period, int split, hipStream_t s) {
    if (n == 0) return 0;
    double bc1 = 1.0 - pow(e3_beta_double(b1), step), bc2 = 1.0 - pow(e3_beta_double(b2), step);
    size_t nb = (n + 255) / 256;
    if (nb > 65536) nb = 65536;      // measured at 59 M floats: 8192 -> 4.8 TB/s, 65536 -> 5.85 TB/s, no cap 5.7
    adam_kernel<<<dim3((unsigned)nb), dim3(256), 0, s>>>(n, p, g, m, v, (float)(lr / bc1), b1, b2, (float)sqrt(bc2), eps,
                                                         (float)(lr_b / bc1), period, split, e3_one_minus_beta(b1),
                                                         e3_one_minus_beta(b2));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "adam_kernel");
}

// All parameter groups of one flat buffer in ONE launch: consecutive segments [end[k-1], end[k]) with their own
// learning rate and eps (xyz | f_dc | f_rest | opacity | scaling | rotation | c).  Same arithmetic as adam_kernel.
// The launch is a list of PIECES -- the segments that take part, minus the gap -- cut into 256-element chunks: a workgroup
// finds the piece of its chunk with scalar compares, so the group's constants are scalar loads from the argument block
// and no element is visited that has nothing to do.  (Round 3's form looked the segment up PER ELEMENT -- seven 64-bit
// compares and three vector loads from the argument block for each float -- and ran at 4.6 TB/s against adam_kernel's 5.85.)
constexpr int ADAM_MAX_SEG = 8;
constexpr int ADAM_MAX_PIECES = 2 * ADAM_MAX_SEG;
struct AdamPieces { size_t begin[ADAM_MAX_PIECES]; size_t end[ADAM_MAX_PIECES]; unsigned chunk_first[ADAM_MAX_PIECES + 1];
                    float step_size[ADAM_MAX_PIECES]; float eps[ADAM_MAX_PIECES]; float bc2_sqrt[ADAM_MAX_PIECES]; int n; };
__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, AdamPieces pc,
                                                            float b1, float b2, float omb1, float omb2) {
    const unsigned nchunks = pc.chunk_first[pc.n];
    for (unsigned c = blockIdx.x; c < nchunks; c += gridDim.x) {
        int k = 0;                               // (scalar: the chunk index is the workgroup's)
#pragma unroll
        for (int j = 1; j < ADAM_MAX_PIECES; ++j) k += (j < pc.n && c >= pc.chunk_first[j]) ? 1 : 0;
        const size_t i = pc.begin[k] + (size_t)(c - pc.chunk_first[k]) * 256u + threadIdx.x;
        if (i >= pc.end[k]) continue;
        const float bc2 = pc.bc2_sqrt[k], ep = pc.eps[k], ss = pc.step_size[k];
        float gi = g[i];
        float mi = m[i] + omb1 * (gi - m[i]);
        float vi = v[i] * b2 + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = __builtin_sqrtf(vi) / bc2 + ep;
        p[i] = p[i] - ss * (mi / denom);
    }
}
// steps[k]: the 1-based Adam step of segment k (torch keeps one `step` per parameter; groups that were skipped on some
// iterations lag behind), <= 0: skip the segment (parameter and moments untouched).  steps == NULL: `step` for all.
// gap_len > 0: the elements [gap_begin, gap_begin + gap_len) are left alone.  The trainer's buffer is xyz | SH | opacity |
// scaling | rotation | c with the SH coefficients updated by their own kernel: one launch for everything around them.
int e3_adam_segments_impl(size_t n, float* p, const float* g, float* m, float* v, int nseg, const size_t* seg_end,
                          const float* lr, const float* eps, float b1, float b2, int step, const int* steps,
                          hipStream_t s, size_t gap_begin, size_t gap_len) {
    if (n == 0) return 0;
    if (gap_len > n || gap_begin > n - gap_len) return e3_fail(hipErrorInvalidValue, "the gap must lie inside the buffer");
    if (nseg < 1 || nseg > ADAM_MAX_SEG) return e3_fail(hipErrorInvalidValue, "1..8 segments");
    AdamPieces pc;
    for (int q = 0; q < ADAM_MAX_PIECES; ++q) {
        pc.begin[q] = pc.end[q] = 0; pc.chunk_first[q] = 0u; pc.step_size[q] = 0.0f; pc.eps[q] = 1.0f; pc.bc2_sqrt[q] = 1.0f;
    }
    pc.n = 0;
    size_t prev = 0, chunks = 0;
    const size_t gap_end = gap_begin + gap_len;
    for (int k = 0; k < nseg; ++k) {
        if (seg_end[k] < prev || seg_end[k] > n) return e3_fail(hipErrorInvalidValue, "segment ends must ascend within n");
        const int st = steps ? steps[k] : step;
        const size_t b = prev, e = seg_end[k];
        prev = e;
        if (st <= 0 || b == e) continue;         // a group torch.optim.Adam would skip (its .grad is None): not in the launch
        const double bc1 = 1.0 - pow(e3_beta_double(b1), st), bc2 = 1.0 - pow(e3_beta_double(b2), st);
        // the segment minus the gap: up to two pieces
        const size_t lo[2] = {b, b > gap_end ? b : gap_end}, hi[2] = {e < gap_begin ? e : gap_begin, e};
        for (int h = 0; h < 2; ++h) {
            if (gap_len == 0 && h == 1) break;
            const size_t pb = gap_len ? lo[h] : b, pe = gap_len ? hi[h] : e;
            if (pb >= pe) continue;
            const int q = pc.n++;
            pc.begin[q] = pb; pc.end[q] = pe; pc.chunk_first[q] = (unsigned)chunks;
            pc.step_size[q] = (float)((double)lr[k] / bc1); pc.eps[q] = eps[k]; pc.bc2_sqrt[q] = (float)sqrt(bc2);
            chunks += (pe - pb + 255) / 256;
            if (chunks > 0xFFFFFFFFull) return e3_fail(hipErrorInvalidValue, "too many elements for one launch");
        }
    }
    if (seg_end[nseg - 1] != n) return e3_fail(hipErrorInvalidValue, "last segment must end at n");
    for (int q = pc.n; q <= ADAM_MAX_PIECES; ++q) pc.chunk_first[q] = (unsigned)chunks;
    if (chunks == 0) return 0;
    size_t nb = chunks;
    if (nb > 65536) nb = 65536;      // measured at 59 M floats: 8192 -> 4.8 TB/s, 65536 -> 5.85 TB/s, no cap 5.7
    adam_segments_kernel<<<dim3((unsigned)nb), dim3(256), 0, s>>>(p, g, m, v, pc, b1, b2, e3_one_minus_beta(b1),
                                                                  e3_one_minus_beta(b2));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "adam_segments_kernel");
}

// ------------------------------------------------------------------------------------ SSIM
// utils/loss_utils.py:359-418: 11x11 Gaussian window (sigma 1.5, normalised), zero padding 5, depthwise,
// C1 = 0.01^2, C2 = 0.03^2, mean over the map.  `to_gray` applies rgb_to_grayscale (:18-23) to both inputs
// first (ssim_gray :368-385).  Forward writes the three partials dm/dmu1, dm/ds11, dm/ds12 per pixel; the
// backward pass blurs them (the window is symmetric) and applies the chain rule of mu1, E[x^2], E[xy].
// Work decomposition: a workgroup produces a 32x32 output tile from a 42x42 input patch (1.7x read amplification, served
// by L2).  Both separable passes are register-tiled: a thread of the horizontal pass slides the window over 18 inputs
// for a run of 8 outputs (36 LDS reads and 54 products instead of 176 and 264), a thread of the vertical pass over 14 rows
// for 4 outputs.  Every output still accumulates its 11 taps in ascending order (same rounding as the untiled form).
// Row pitches 43 / 33 keep both passes free of LDS bank conflicts for the lane -> (row, run) maps used below.
constexpr int SS_R = 5, SS_TX = 32, SS_TY = 32, SS_INX = SS_TX + 2 * SS_R, SS_INY = SS_TY + 2 * SS_R;
constexpr int SS_SP = SS_INX + 1, SS_HP = SS_TX + 1, SS_RUN_H = 8, SS_RUN_V = 4;
static_assert(SS_INY * (SS_TX / SS_RUN_H) <= 256 && SS_TX * (SS_TY / SS_RUN_V) == 256, "one pass per thread");
struct SsimWin { float w[11]; };      // the 11 taps travel as a kernel argument: no per-device / per-process state

__device__ __forceinline__ float ss_load(const float* __restrict__ img, int C, int to_gray, size_t HW, int ch, int x, int y,
                                         int W, int H) {
    if (x < 0 || y < 0 || x >= W || y >= H) return 0.0f;     // zero padding
    size_t p = (size_t)y * W + x;
    if (to_gray) return FMA(0.114f, img[2 * HW + p], FMA(0.587f, img[HW + p], 0.299f * img[p]));
    return img[ch * HW + p];
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimWin win, int C, int H, int W, int to_gray, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ partial3,
                                                       double* __restrict__ block_sums) {
    // the input patches are dead once the horizontal pass has read them: its results go into the same LDS (one more
    // barrier, 28 instead of 42 KB per workgroup)
    __shared__ float lds[5 * SS_INY * SS_HP];
    static_assert(2 * SS_INY * SS_SP <= 5 * SS_INY * SS_HP, "patches fit under the horizontal results");
    float (*s1)[SS_SP] = reinterpret_cast<float (*)[SS_SP]>(lds);
    float (*s2)[SS_SP] = reinterpret_cast<float (*)[SS_SP]>(lds + SS_INY * SS_SP);
    float (*h)[SS_INY][SS_HP] = reinterpret_cast<float (*)[SS_INY][SS_HP]>(lds);
    __shared__ double wred[4];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
    const size_t HW = (size_t)H * W;
    // patch loads: unconditional (clamped address, value selected afterwards) and fully unrolled, so that all of a
    // thread's requests are in flight together -- a load under `if (inside)` waits for its own round trip
    constexpr int NLOAD = (SS_INY * SS_INX + 255) / 256;
    float v1[NLOAD], v2[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = min((int)threadIdx.x + it * 256, SS_INY * SS_INX - 1);
        const int ly = i / SS_INX, lx = i - ly * SS_INX;
        const int x = x0 + lx - SS_R, y = y0 + ly - SS_R;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const size_t p = (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
        float a, b;
        if (to_gray) {
            a = FMA(0.114f, img1[2 * HW + p], FMA(0.587f, img1[HW + p], 0.299f * img1[p]));
            b = FMA(0.114f, img2[2 * HW + p], FMA(0.587f, img2[HW + p], 0.299f * img2[p]));
        } else {
            a = img1[ch * HW + p]; b = img2[ch * HW + p];
        }
        v1[it] = in ? a : 0.0f; v2[it] = in ? b : 0.0f;                 // zero padding
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < SS_INY * SS_INX) { const int ly = i / SS_INX, lx = i - ly * SS_INX; s1[ly][lx] = v1[it]; s2[ly][lx] = v2[it]; }
    }
    __syncthreads();
    const bool hz = threadIdx.x < SS_INY * (SS_TX / SS_RUN_H);       // horizontal pass: (row, run of 8 columns)
    const int row = hz ? threadIdx.x >> 2 : 0, c0 = (threadIdx.x & 3) * SS_RUN_H;
    float acc[SS_RUN_H][5];
#pragma unroll
    for (int o = 0; o < SS_RUN_H; ++o) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = acc[o][4] = 0.0f;
    if (hz) {
#pragma unroll
        for (int j = 0; j < SS_RUN_H + 10; ++j) {
            const float u = s1[row][c0 + j], v = s2[row][c0 + j];
            const float uu = u * u, vv = v * v, uv = u * v;
#pragma unroll
            for (int o = 0; o < SS_RUN_H; ++o) {
                const int k = j - o;
                if (k >= 0 && k < 11) {
                    const float w = win.w[k];
                    acc[o][0] = FMA(w, u, acc[o][0]); acc[o][1] = FMA(w, v, acc[o][1]); acc[o][2] = FMA(w, uu, acc[o][2]);
                    acc[o][3] = FMA(w, vv, acc[o][3]); acc[o][4] = FMA(w, uv, acc[o][4]);
                }
            }
        }
    }
    __syncthreads();                                                  // every patch value has been read
    if (hz) {
#pragma unroll
        for (int o = 0; o < SS_RUN_H; ++o)
#pragma unroll
            for (int q = 0; q < 5; ++q) h[q][row][c0 + o] = acc[o][q];
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * SS_RUN_V;     // vertical pass: (column, run of 4 rows)
    float out[SS_RUN_V][5];
#pragma unroll
    for (int o = 0; o < SS_RUN_V; ++o) out[o][0] = out[o][1] = out[o][2] = out[o][3] = out[o][4] = 0.0f;
#pragma unroll
    for (int j = 0; j < SS_RUN_V + 10; ++j) {
        float hv[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) hv[q] = h[q][r0 + j][lx];
#pragma unroll
        for (int o = 0; o < SS_RUN_V; ++o) {
            const int k = j - o;
            if (k >= 0 && k < 11) {
                const float w = win.w[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) out[o][q] = FMA(w, hv[q], out[o][q]);
            }
        }
    }
    const int x = x0 + lx;
    double v = 0.0;
#pragma unroll
    for (int o = 0; o < SS_RUN_V; ++o) {
        const int y = y0 + r0 + o;
        if (x < W && y < H) {
            const float mu1 = out[o][0], mu2 = out[o][1], e11 = out[o][2], e22 = out[o][3], e12 = out[o][4];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            float A1 = 2.0f * mu1 * mu2 + C1, A2 = 2.0f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
            float inv = 1.0f / (B1 * B2);
            float m = A1 * A2 * inv;
            v += (double)m;
            if (partial3) {
                size_t p = (size_t)ch * HW + (size_t)y * W + x;
                size_t CHW = (size_t)gridDim.z * HW;
                float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -m / B1, dB2 = -m / B2;
                partial3[p] = dA1 * 2.0f * mu2 - dA2 * 2.0f * mu2 + dB1 * 2.0f * mu1 - dB2 * 2.0f * mu1;   // dm/dmu1
                partial3[CHW + p] = dB2;                                                                    // dm/dE[x^2]
                partial3[2 * CHW + p] = 2.0f * dA2;                                                          // dm/dE[xy]
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        block_sums[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (wred[0] + wred[1]) + (wred[2] + wred[3]);
}

__global__ __launch_bounds__(WAVE) void ssim_finalize_kernel(int nblocks, double count, const double* __restrict__ sums,
                                                             float* __restrict__ out_mean) {
    double a = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += WAVE) a += sums[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) out_mean[0] = (float)(a / count);
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimWin win, int C, int H, int W, int to_gray, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, const float* __restrict__ partial3,
                                                       float scale, float* __restrict__ d_img1, float l1_scale,
                                                       double* __restrict__ l1_sums, int rank1) {
    // l1_sums != NULL (e3dgs_image_loss): d_img1 = scale * dSSIM/dimg + l1_scale * sign(img1 - img2) (per channel, or
    // on the gray values with the channel weights), and sum |img1 - img2| of the block goes to l1_sums[block]
    __shared__ float lds[3 * SS_INY * SS_SP];                    // patches, then (same LDS) the horizontal results
    float (*sp)[SS_INY][SS_SP] = reinterpret_cast<float (*)[SS_INY][SS_SP]>(lds);
    float (*h)[SS_INY][SS_HP] = reinterpret_cast<float (*)[SS_INY][SS_HP]>(lds);
    __shared__ double wred[4];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * SS_TX, y0 = blockIdx.y * SS_TY;
    const size_t HW = (size_t)H * W, CHW = (size_t)gridDim.z * HW;
    constexpr int NLOAD = (SS_INY * SS_INX + 255) / 256;      // (unconditional, unrolled loads: see ssim_fwd_kernel)
    float pv[NLOAD][3];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = min((int)threadIdx.x + it * 256, SS_INY * SS_INX - 1);
        const int ly = i / SS_INX, lx = i - ly * SS_INX;
        const int x = x0 + lx - SS_R, y = y0 + ly - SS_R;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const size_t p = (size_t)ch * HW + (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float t = partial3[q * CHW + p]; pv[it][q] = in ? t : 0.0f; }
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < SS_INY * SS_INX) {
            const int ly = i / SS_INX, lx = i - ly * SS_INX;
            sp[0][ly][lx] = pv[it][0]; sp[1][ly][lx] = pv[it][1]; sp[2][ly][lx] = pv[it][2];
        }
    }
    __syncthreads();
    const bool hz = threadIdx.x < SS_INY * (SS_TX / SS_RUN_H);       // horizontal pass: (row, run of 8 columns)
    const int row = hz ? threadIdx.x >> 2 : 0, c0 = (threadIdx.x & 3) * SS_RUN_H;
    float acc[SS_RUN_H][3];
#pragma unroll
    for (int o = 0; o < SS_RUN_H; ++o) acc[o][0] = acc[o][1] = acc[o][2] = 0.0f;
    if (hz) {
#pragma unroll
        for (int j = 0; j < SS_RUN_H + 10; ++j) {
            const float a = sp[0][row][c0 + j], b = sp[1][row][c0 + j], c = sp[2][row][c0 + j];
#pragma unroll
            for (int o = 0; o < SS_RUN_H; ++o) {
                const int k = j - o;
                if (k >= 0 && k < 11) {
                    const float w = win.w[k];
                    acc[o][0] = FMA(w, a, acc[o][0]); acc[o][1] = FMA(w, b, acc[o][1]); acc[o][2] = FMA(w, c, acc[o][2]);
                }
            }
        }
    }
    __syncthreads();                                                  // every patch value has been read
    if (hz) {
#pragma unroll
        for (int o = 0; o < SS_RUN_H; ++o) { h[0][row][c0 + o] = acc[o][0]; h[1][row][c0 + o] = acc[o][1]; h[2][row][c0 + o] = acc[o][2]; }
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * SS_RUN_V;     // vertical pass: (column, run of 4 rows)
    float out[SS_RUN_V][3];
#pragma unroll
    for (int o = 0; o < SS_RUN_V; ++o) out[o][0] = out[o][1] = out[o][2] = 0.0f;
#pragma unroll
    for (int j = 0; j < SS_RUN_V + 10; ++j) {
        const float h0 = h[0][r0 + j][lx], h1 = h[1][r0 + j][lx], h2 = h[2][r0 + j][lx];
#pragma unroll
        for (int o = 0; o < SS_RUN_V; ++o) {
            const int k = j - o;
            if (k >= 0 && k < 11) {
                const float w = win.w[k];
                out[o][0] = FMA(w, h0, out[o][0]); out[o][1] = FMA(w, h1, out[o][1]); out[o][2] = FMA(w, h2, out[o][2]);
            }
        }
    }
    const int x = x0 + lx;
    double absdiff = 0.0;
#pragma unroll
    for (int o = 0; o < SS_RUN_V; ++o) {
        const int y = y0 + r0 + o;
        if (x < W && y < H) {
            float u = ss_load(img1, C, to_gray, HW, ch, x, y, W, H), v = ss_load(img2, C, to_gray, HW, ch, x, y, W, H);
            float g = scale * (out[o][0] + 2.0f * u * out[o][1] + v * out[o][2]);
            if (l1_sums) {
                const float e = u - v;
                absdiff += (double)fabsf(e);
                g += l1_scale * (float)((e > 0.0f) - (e < 0.0f));
            }
            size_t p = (size_t)y * W + x;
            if (to_gray && rank1) d_img1[p] = g;         // dL/dC = g * (0.299, 0.587, 0.114): the scalar field alone
            else if (to_gray) { d_img1[p] = 0.299f * g; d_img1[HW + p] = 0.587f * g; d_img1[2 * HW + p] = 0.114f * g; }
            else d_img1[ch * HW + p] = g;
        }
    }
    if (l1_sums) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) absdiff += __shfl_xor(absdiff, o, 64);
        if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = absdiff;
        __syncthreads();
        if (threadIdx.x == 0)
            l1_sums[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (wred[0] + wred[1]) + (wred[2] + wred[3]);
    }
}

// scalars[0] = (1 - lambda) L1 + lambda (1 - SSIM), [1] = L1, [2] = SSIM   (train.py:213-223 / 292-296)
__global__ __launch_bounds__(256) void image_loss_finalize_kernel(int nblocks, double count, float lambda_dssim,
                                                                  const double* __restrict__ ssim_sums,
                                                                  const double* __restrict__ l1_sums,
                                                                  float* __restrict__ scalars) {
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += ssim_sums[i]; b += l1_sums[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ssim = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / count;
        const double l1 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / count;
        scalars[0] = (float)((1.0 - (double)lambda_dssim) * l1 + (double)lambda_dssim * (1.0 - ssim));
        scalars[1] = (float)l1; scalars[2] = (float)ssim; scalars[3] = 0.0f;
    }
}

static SsimWin ssim_window() {
    // the reference builds the window in fp32 (torch.Tensor of python floats, then / sum): utils/loss_utils.py:359-367
    SsimWin win;
    float fs = 0.0f;
    for (int i = 0; i < 11; ++i) { win.w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); fs += win.w[i]; }
    for (int i = 0; i < 11; ++i) win.w[i] = win.w[i] / fs;
    return win;
}

size_t e3_ssim_scratch_bytes(int C, int H, int W) {
    size_t nb = (size_t)((W + SS_TX - 1) / SS_TX) * ((H + SS_TY - 1) / SS_TY) * C;
    return nb * sizeof(double) + 3 * (size_t)C * H * W * sizeof(float) + 512;
}

int e3_ssim_impl(int C, int H, int W, int to_gray, const float* img1, const float* img2, float* out_mean, float* d_img1,
                 char* scratch, hipStream_t s) {
    if (to_gray && C != 3) return e3_fail(hipErrorInvalidValue, "to_gray needs 3-channel inputs");
    const SsimWin win = ssim_window();
    const int Ceff = to_gray ? 1 : C;
    dim3 grid((W + SS_TX - 1) / SS_TX, (H + SS_TY - 1) / SS_TY, Ceff);
    size_t nb = (size_t)grid.x * grid.y * grid.z;
    double* sums = reinterpret_cast<double*>(scratch);
    float* partial3 = reinterpret_cast<float*>(scratch + align_up(nb * sizeof(double), 256));
    ssim_fwd_kernel<<<grid, dim3(256), 0, s>>>(win, C, H, W, to_gray, img1, img2, d_img1 ? partial3 : nullptr, sums);
    ssim_finalize_kernel<<<dim3(1), dim3(WAVE), 0, s>>>((int)nb, (double)Ceff * H * W, sums, out_mean);
    if (d_img1)
        ssim_bwd_kernel<<<grid, dim3(256), 0, s>>>(win, C, H, W, to_gray, img1, img2, partial3, 1.0f / ((float)Ceff * H * W), d_img1,
                                                   0.0f, nullptr, 0);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "ssim kernels");
}

// (1 - lambda) L1 + lambda (1 - SSIM) and its image gradient in three launches (the --gray and RGB iterations).
size_t e3_image_loss_scratch_bytes(int C, int H, int W) {
    size_t nb = (size_t)((W + SS_TX - 1) / SS_TX) * ((H + SS_TY - 1) / SS_TY) * C;
    return e3_ssim_scratch_bytes(C, H, W) + nb * sizeof(double) + 256;
}
int e3_image_loss_impl(int C, int H, int W, int to_gray, float lambda_dssim, const float* img, const float* gt,
                       float* scalars, float* d_img, char* scratch, hipStream_t s, int rank1) {
    if (to_gray && C != 3) return e3_fail(hipErrorInvalidValue, "to_gray needs 3-channel inputs");
    if (rank1 && !to_gray) return e3_fail(hipErrorInvalidValue, "a rank-1 pixel gradient needs the gray loss");
    if (!d_img || !scalars) return e3_fail(hipErrorInvalidValue, "scalars and d_img are required");
    const SsimWin win = ssim_window();
    const int Ceff = to_gray ? 1 : C;
    dim3 grid((W + SS_TX - 1) / SS_TX, (H + SS_TY - 1) / SS_TY, Ceff);
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    double* sums = reinterpret_cast<double*>(scratch);
    float* partial3 = reinterpret_cast<float*>(scratch + align_up(nb * sizeof(double), 256));
    double* l1_sums = reinterpret_cast<double*>(scratch + align_up(e3_ssim_scratch_bytes(C, H, W), 256));
    const float n = (float)Ceff * H * W;
    ssim_fwd_kernel<<<grid, dim3(256), 0, s>>>(win, C, H, W, to_gray, img, gt, partial3, sums);
    ssim_bwd_kernel<<<grid, dim3(256), 0, s>>>(win, C, H, W, to_gray, img, gt, partial3, -lambda_dssim / n, d_img,
                                               (1.0f - lambda_dssim) / n, l1_sums, rank1);
    image_loss_finalize_kernel<<<dim3(1), dim3(256), 0, s>>>((int)nb, (double)n, lambda_dssim, sums, l1_sums, scalars);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : e3_fail(e, "image loss kernels");
}
