// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the compositing kernels
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   stream_read    coalesced float4 loads of a 1 GiB array                         (known: 1 GiB)
//   gather48       3 x float4 loads of a 48-B record at a random index, 8 M times  (known: 384 MB touched; the
//                  records straddle 64-B sectors, so the sector-granular figure is ~1.5 x 64 B x 8 M = 768 MB)
//   stream_write   coalesced float4 stores of 1 GiB
//   scatter48      3 x float4 stores of a 48-B record at a random index, 8 M times
// Run:  rocprofv3 --pmc FETCH_SIZE -- ./pmc_calib ;  rocprofv3 --pmc WRITE_SIZE -- ./pmc_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void stream_read(const float4* __restrict__ a, size_t n, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) *out = s;
}
__global__ void stream_write(float4* __restrict__ a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1, 2, 3, 4);
}
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void gather48(const float4* __restrict__ rec, uint32_t nrec, uint32_t count, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t id = hash(i) % nrec;
    float4 a = rec[3 * (size_t)id], b = rec[3 * (size_t)id + 1], c = rec[3 * (size_t)id + 2];
    float s = a.x + b.y + c.z;
    if (s == 12345.678f) *out = s;
}
__global__ void scatter48(float4* __restrict__ rec, uint32_t nrec, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t id = hash(i) % nrec;
    rec[3 * (size_t)id] = make_float4(1, 2, 3, 4); rec[3 * (size_t)id + 1] = make_float4(5, 6, 7, 8);
    rec[3 * (size_t)id + 2] = make_float4(9, 0, 0, 0);
}
int main() {
    const size_t GiB = 1ull << 30;
    float4* a; float* out;
    hipMalloc(&a, 2 * GiB); hipMalloc(&out, 4);
    hipMemset(a, 0, 2 * GiB);
    const size_t n4 = GiB / 16;
    const uint32_t nrec = (uint32_t)(2 * GiB / 48), count = 8u << 20;
    for (int rep = 0; rep < 2; ++rep) {
        stream_read<<<4096, 256>>>(a, n4, out);
        gather48<<<(count + 255) / 256, 256>>>(a, nrec, count, out);
        stream_write<<<4096, 256>>>(a, n4);
        scatter48<<<(count + 255) / 256, 256>>>(a, nrec, count);
        hipDeviceSynchronize();
    }
    printf("known bytes: stream 1073741824; gather/scatter touched %u x 48 = %llu\n", count, 48ull * count);
    return 0;
}
