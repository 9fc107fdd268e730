// Does the SH optimizer's coefficient-major layout (48 planes of N floats per array; a thread walks 24 slices of 2
// coefficients x 3 channels = 6 planes x 3 arrays = 18 concurrent streams) cost DRAM efficiency against a layout in which the
// 48 x 64 floats of a WAVE are contiguous (blocked: [N / 64][48][64])?  Both read p, m, v and write p, m, v once:
//   hipcc --offload-arch=gfx950 -O3 -o sh_layout sh_layout.hip && ./sh_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int M3 = 48;
template <bool BLOCKED, int OCC>
__global__ __launch_bounds__(256, OCC) void k(int N, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t base = BLOCKED ? (size_t)(i >> 6) * (M3 * 64) + (i & 63) : (size_t)i;
    const size_t st = BLOCKED ? 64 : (size_t)N;
#pragma unroll 1
    for (int sl = 0; sl < M3 / 6; ++sl) {
        float a[6], b[6], c[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const size_t e = base + (size_t)(6 * sl + j) * st;
            a[j] = p[e]; b[j] = m[e]; c[j] = v[e];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const size_t e = base + (size_t)(6 * sl + j) * st;
            const float mi = b[j] * 0.9f + 0.1f * a[j];
            const float vi = c[j] * 0.999f + 0.001f * a[j] * a[j];
            m[e] = mi; v[e] = vi; p[e] = a[j] - 1e-3f * mi / (sqrtf(vi) + 1e-15f);
        }
    }
}
template <bool BLOCKED, int OCC>
static float run(int N, float* p, float* m, float* v) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k<BLOCKED, OCC><<<(N + 255) / 256, 256>>>(N, p, m, v);
    CK(hipEventRecord(e0));
    const int R = 20;
    for (int r = 0; r < R; ++r) k<BLOCKED, OCC><<<(N + 255) / 256, 256>>>(N, p, m, v);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / R;
}
int main() {
    const int N = 1000000 / 64 * 64;
    float *p, *m, *v;
    const size_t bytes = (size_t)N * M3 * 4;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&m, bytes)); CK(hipMalloc(&v, bytes));
    CK(hipMemset(p, 0, bytes)); CK(hipMemset(m, 0, bytes)); CK(hipMemset(v, 0, bytes));
    const double gb = 6.0 * bytes / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        float t;
        t = run<false, 2>(N, p, m, v); printf("planar  2 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<true, 2>(N, p, m, v);  printf("blocked 2 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<false, 8>(N, p, m, v); printf("planar  8 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<true, 8>(N, p, m, v);  printf("blocked 8 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<false, 1>(N, p, m, v); printf("planar  1 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<false, 3>(N, p, m, v); printf("planar  3 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
        t = run<false, 4>(N, p, m, v); printf("planar  4 wg/cu: %.1f us  %.2f TB/s\n", t * 1e3, gb / t);
    }
    return 0;
}
