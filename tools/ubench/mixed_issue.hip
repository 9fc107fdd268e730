// Micro-benchmark: do SALU / LDS / branch instructions of OTHER waves issue beside a SIMD's VALU stream, or does every
// instruction take the SIMD's issue slot?  Instruction mix of render_bwd_kernel (profiles/r02_final/sq_instruction_mix.txt):
// VALU : SALU : LDS : branch = 541 : 212 : 61 : 82, i.e. per 27 VALU about 10.6 SALU, 3 LDS, 4 branches; 6 waves per SIMD.
// Every variant runs the SAME 27 independent-chain v_fma_f32 per iteration and adds the other types one by one;
// the time per iteration per SIMD says what they cost on top.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mixed_issue tools/ubench/mixed_issue.hip && tools/ubench/mixed_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

#define FMA9(a)                                                                                                       \
    asm volatile("v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n"                  \
                 "v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n"                  \
                 "v_fma_f32 %6, %6, %9, %10\n v_fma_f32 %7, %7, %9, %10\n v_fma_f32 %8, %8, %9, %10\n"                  \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),      \
                   "+v"(x[8])                                                                                          \
                 : "v"(a), "v"(b))
#define SALU4() asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, %0\n s_add_u32 %2, %2, 3\n s_and_b32 %3, %3, %2\n" \
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc")
#define SALU3() asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, %0\n s_add_u32 %2, %2, 3\n" : "+s"(s0), "+s"(s1), "+s"(s2) : : "scc")
#define LDS1() asm volatile("ds_read_b32 %0, %1\n" : "=v"(l) : "v"(laddr))
#define BR1() asm volatile("s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n 1:\n" ::"s"(s0) : "scc")

template <int MODE>   // 0: VALU only, 1: + SALU, 2: + SALU + LDS, 3: + SALU + LDS + branch, 4: VALU + branch, 5: VALU + LDS
__global__ __launch_bounds__(256) void k_mixed(float* out, float a, float b, int n) {
    __shared__ float sm[256];
    sm[threadIdx.x] = a;
    __syncthreads();
    float x[9];
    for (int i = 0; i < 9; ++i) x[i] = a + i + threadIdx.x;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    float l = 0.0f;
    const unsigned laddr = (threadIdx.x & 63) * 4;
    for (int it = 0; it < n; ++it) {
        FMA9(a);
        if (MODE == 1 || MODE == 2 || MODE == 3) SALU4();
        if (MODE == 2 || MODE == 3 || MODE == 5) LDS1();
        if (MODE == 3 || MODE == 4) { BR1(); BR1(); }
        FMA9(a);
        if (MODE == 1 || MODE == 2 || MODE == 3) SALU4();
        if (MODE == 2 || MODE == 3 || MODE == 5) LDS1();
        if (MODE == 3 || MODE == 4) { BR1(); }
        FMA9(a);
        if (MODE == 1 || MODE == 2 || MODE == 3) SALU3();
        if (MODE == 2 || MODE == 3 || MODE == 5) { LDS1(); asm volatile("s_waitcnt lgkmcnt(0)"); x[0] += l; }
        if (MODE == 3 || MODE == 4) { BR1(); }
    }
    float s = 0;
    for (int i = 0; i < 9; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(s0 + s1 + s2 + s3);
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd, n = 2048;
    kern<<<blocks, 256>>>(out, 1.0001f, 0.5f, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 1.0001f, 0.5f, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // iterations per SIMD = waves per SIMD x n
    const double ns_per_iter = ms * 1e6 / ((double)waves_per_simd * n);
    printf("%-34s %d waves/SIMD: %7.3f ms  %.2f ns per iteration per SIMD = %.1f cycles @2.4 GHz (27 VALU: %.2f cyc each)\n",
           name, waves_per_simd, ms, ns_per_iter, ns_per_iter * 2.4, ns_per_iter * 2.4 / 27.0);
    hipFree(out);
}
int main() {
    for (int w : {6, 8, 2, 1}) {
        run("27 VALU", k_mixed<0>, w);
        run("27 VALU + 11 SALU", k_mixed<1>, w);
        run("27 VALU + 11 SALU + 3 LDS", k_mixed<2>, w);
        run("27 VALU + 11 SALU + 3 LDS + 4 br", k_mixed<3>, w);
        run("27 VALU + 4 branch", k_mixed<4>, w);
        run("27 VALU + 3 LDS", k_mixed<5>, w);
    }
    return 0;
}
