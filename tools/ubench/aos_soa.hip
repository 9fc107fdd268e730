// Do the 12- / 16-byte AoS rows of xyz / scale / rotation (one thread per Gaussian: global_load_dwordx3 / x4 at a 12- / 16-
// byte lane stride) cost anything against planar arrays?  Reads xyz (3), scale (3), rotation (4), writes three gradients of the
// same shapes -- the per-Gaussian traffic of geom_bwd_multi_kernel / preprocess_kernel without their arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -o aos_soa aos_soa.hip && ./aos_soa
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <bool SOA>
__global__ __launch_bounds__(256) void k(int N, const float* __restrict__ xyz, const float* __restrict__ sc,
                                         const float* __restrict__ rot, float* __restrict__ gx, float* __restrict__ gs,
                                         float* __restrict__ gr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float a[3], b[3], c[4];
    if (SOA) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = xyz[(size_t)k * N + i]; b[k] = sc[(size_t)k * N + i]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = rot[(size_t)k * N + i];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = xyz[3 * (size_t)i + k]; b[k] = sc[3 * (size_t)i + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = rot[4 * (size_t)i + k];
    }
    const float s = a[0] * b[1] + a[1] * b[2] + a[2] * b[0] + c[0] * c[1] + c[2] * c[3];
    if (SOA) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { gx[(size_t)k * N + i] = a[k] + s; gs[(size_t)k * N + i] = b[k] - s; }
#pragma unroll
        for (int k = 0; k < 4; ++k) gr[(size_t)k * N + i] = c[k] * s;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { gx[3 * (size_t)i + k] = a[k] + s; gs[3 * (size_t)i + k] = b[k] - s; }
#pragma unroll
        for (int k = 0; k < 4; ++k) gr[4 * (size_t)i + k] = c[k] * s;
    }
}
template <bool SOA>
static float run(int N, float** p) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k<SOA><<<(N + 255) / 256, 256>>>(N, p[0], p[1], p[2], p[3], p[4], p[5]);
    CK(hipEventRecord(e0));
    const int R = 50;
    for (int r = 0; r < R; ++r) k<SOA><<<(N + 255) / 256, 256>>>(N, p[0], p[1], p[2], p[3], p[4], p[5]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
int main() {
    for (int N : {1000000, 8000000}) {
        float* p[6];
        const size_t w[6] = {3, 3, 4, 3, 3, 4};
        for (int j = 0; j < 6; ++j) { CK(hipMalloc(&p[j], w[j] * N * 4)); CK(hipMemset(p[j], 0, w[j] * N * 4)); }
        const double gb = 20.0 * N * 4 / 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            float t = run<false>(N, p); printf("N=%d AoS (dwordx3 / x4 rows): %.1f us  %.2f TB/s\n", N, t * 1e3, gb / t);
            t = run<true>(N, p);        printf("N=%d planar               : %.1f us  %.2f TB/s\n", N, t * 1e3, gb / t);
        }
        for (int j = 0; j < 6; ++j) CK(hipFree(p[j]));
    }
    return 0;
}
