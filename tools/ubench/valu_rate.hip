// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the compositing kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define N_ITERS 4096
#define CHAINS 8
#define OPS(NAME, BODY)                                                                       \
    __global__ void k_##NAME(float* out, float a, float b, int n) {                            \
        float x[CHAINS];                                                                     \
        int ix[CHAINS];                                                                       \
        for (int i = 0; i < CHAINS; ++i) { x[i] = a + i + threadIdx.x; ix[i] = (int)x[i]; }  \
        for (int it = 0; it < n; ++it) {                                                     \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { BODY; }                      \
        }                                                                                     \
        float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i] + ix[i];                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
    }
OPS(fma, x[i] = __builtin_fmaf(x[i], a, b))
OPS(mul, x[i] = x[i] * a)
OPS(add, x[i] = x[i] + a)
OPS(minf, x[i] = fminf(x[i], a))
OPS(cndmask, x[i] = (x[i] > b) ? a : x[i] + 1.0f)   // cmp + add + cndmask (3 ops)
OPS(ldexp, x[i] = __builtin_ldexpf(x[i], ix[i] & 1))   // and + ldexp
OPS(rndne, x[i] = __builtin_rintf(x[i]) + a)           // rndne + add
OPS(cvt, ix[i] = (int)(x[i]); x[i] = x[i] + (float)ix[i])   // cvt_i32, cvt_f32, add
OPS(rcp, x[i] = __builtin_amdgcn_rcpf(x[i]) + a)       // rcp + add
OPS(exp2, x[i] = __builtin_amdgcn_exp2f(x[i]) * a)     // exp + mul
OPS(dpp, x[i] = x[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0xB1, 0xf, 0xf, false)))
OPS(cmp_sel, x[i] = (x[i] >= a) ? x[i] : b)           // cmp + cndmask
OPS(sub, x[i] = a - x[i])
OPS(swz, x[i] = x[i] + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x[i]), 0x041F)))
OPS(bperm, x[i] = x[i] + __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x ^ 16) << 2, __float_as_int(x[i]))))
OPS(dpp_ror, x[i] = x[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x124, 0xf, 0xf, false)))
OPS(dpp_mov, x[i] = a + __int_as_float(__builtin_amdgcn_update_dpp(0, ix[i], 0xB1, 0xf, 0xf, false)); ix[i] += 1)
OPS(sel3, x[i] = (ix[i] & 1) ? x[i] * a : b)

// packed fp32: one v_pk_* instruction carries two independent fp32 operations per lane
typedef float f2 __attribute__((ext_vector_type(2)));
#define PKOPS(NAME, BODY)                                                                     \
    __global__ void k_##NAME(float* out, float a, float b, int n) {                            \
        f2 x[CHAINS];                                                                        \
        const f2 av = {a, a + 0.25f}, bv = {b, b - 0.125f};                                   \
        for (int i = 0; i < CHAINS; ++i) { x[i].x = a + i + threadIdx.x; x[i].y = b + i - threadIdx.x; } \
        for (int it = 0; it < n; ++it) {                                                     \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { BODY; }                      \
        }                                                                                     \
        float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i].x + x[i].y;                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
    }
PKOPS(pk_fma, x[i] = __builtin_elementwise_fma(x[i], av, bv))
PKOPS(pk_mul, x[i] = x[i] * av)
PKOPS(pk_add, x[i] = x[i] + av)

template <typename K>
void run(const char* name, K kern, int ops_per_body) {
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 8;   // 8 blocks of 256 per CU -> 8 waves per SIMD
    kern<<<blocks, 256>>>(out, 1.0001f, 0.5f, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 1.0001f, 0.5f, N_ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITERS * CHAINS * ops_per_body;     // wave-instructions
    double per_simd = wave_instr / 1024.0;
    printf("%-8s %7.3f ms  -> %.2f ns per wave-instr per SIMD (x clock GHz = cycles; @2.4: %.2f cyc)\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}
int main() {
    run("fma", k_fma, 1); run("mul", k_mul, 1); run("add", k_add, 1); run("sub", k_sub, 1); run("min", k_minf, 1);
    run("cmp+add+sel", k_cndmask, 3); run("cmp+sel", k_cmp_sel, 2); run("and+ldexp", k_ldexp, 2);
    run("rndne+add", k_rndne, 2); run("cvt,cvt,add", k_cvt, 3); run("rcp+add", k_rcp, 2); run("exp2+mul", k_exp2, 2);
    run("dpp+add", k_dpp, 1); run("dpp_ror+add", k_dpp_ror, 1); run("swz+add", k_swz, 1); run("bperm+add", k_bperm, 1);
    run("dppmov,add,iadd", k_dpp_mov, 3); run("and,mul,sel", k_sel3, 3);
    run("pk_fma", k_pk_fma, 1); run("pk_mul", k_pk_mul, 1); run("pk_add", k_pk_add, 1);
    return 0;
}
