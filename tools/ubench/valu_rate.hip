// valu_rate.hip -- issue cost of single instructions on MI355X (gfx950), measured so that the numbers can be used as a
// roofline for the compositing kernels (bench.py roofline.issue).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip && tools/ubench/valu_rate
//
// Method (each point answers one objection to the round-3 version of this file):
//   * every timed launch runs >= 20 ms, after a >= 50 ms warm-up of the same kernel (DVFS settles within a few ms; a 0.3 ms
//     launch straight after a short warm-up measured the ramp, not the rate);
//   * the EFFECTIVE CLOCK of every run is printed: one wave per workgroup reads s_memtime (shader cycles) at its first and
//     last instruction, and the delta is divided by the launch's wall time from HIP events -- so a row reads both as ns
//     and as cycles per wave-instruction per SIMD;
//   * the instruction under test is emitted with inline asm (the mnemonic in the table IS the instruction; the loop body
//     is NI copies on NI independent registers, 8 dependent steps apart); tools/ubench/check_isa.sh greps the device
//     assembly of this file for every mnemonic of the table;
//   * 8 waves per SIMD (2048 threads per CU on all 256 CUs): the rate is the chip's, not a single wave's latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NI = 8;          // independent chains per thread
constexpr int UNROLL = 16;     // copies of the NI-instruction group per loop trip

struct Stamp { unsigned long long c0, c1; };

// One kernel per instruction: BODY(i) is the asm for chain i.
#define KERNEL(name, DECL, BODY, SINK)                                                                  \
    __global__ __launch_bounds__(256) void name(int trips, float seed, float* out, Stamp* st) {        \
        DECL                                                                                            \
        const unsigned long long c0 = __builtin_readcyclecounter();                                    \
        for (int t = 0; t < trips; ++t) {                                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                        \
                BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7)                         \
            }                                                                                           \
        }                                                                                               \
        const unsigned long long c1 = __builtin_readcyclecounter();                                    \
        SINK                                                                                            \
        if ((threadIdx.x & 63) == 0) { st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1}; }       \
    }

#define DECL_F float a[NI], b = seed, c = seed * 0.5f; _Pragma("unroll") for (int i = 0; i < NI; ++i) a[i] = seed + (float)(i + threadIdx.x);
#define SINK_F float s = 0.f; _Pragma("unroll") for (int i = 0; i < NI; ++i) s += a[i]; if (s == 12345.678f) out[0] = s;

#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
#define B_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1));
#define B_CMP(i) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define B_CMPS(i) asm volatile("v_cmp_ge_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define B_CNDMASK_S(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
#define B_DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define B_DPPADD(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, 48, %1" : "+v"(a[i]) : "v"(b));
#define B_READLANE(i) asm volatile("v_readlane_b32 s22, %0, 5" : : "v"(a[i]) : "s22");
#define B_SAND(i) asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]" : : : "s20", "s21", "scc");
#define B_SCSEL(i) asm volatile("s_cselect_b64 s[20:21], s[22:23], 0" : : : "s20", "s21");
#define B_SNOP(i) asm volatile("s_nop 0");

KERNEL(k_fma, DECL_F, B_FMA, SINK_F)
KERNEL(k_add, DECL_F, B_ADD, SINK_F)
KERNEL(k_sub, DECL_F, B_SUB, SINK_F)
KERNEL(k_mul, DECL_F, B_MUL, SINK_F)
KERNEL(k_min, DECL_F, B_MIN, SINK_F)
KERNEL(k_mov, DECL_F, B_MOV, SINK_F)
KERNEL(k_exp, DECL_F, B_EXP, SINK_F)
KERNEL(k_rcp, DECL_F, B_RCP, SINK_F)
KERNEL(k_rndne, DECL_F, B_RNDNE, SINK_F)
KERNEL(k_cvt, DECL_F, B_CVT, SINK_F)
KERNEL(k_ldexp, DECL_F, B_LDEXP, SINK_F)
KERNEL(k_cmp, DECL_F, B_CMP, SINK_F)
KERNEL(k_cmps, DECL_F, B_CMPS, SINK_F)
KERNEL(k_cndmask, DECL_F, B_CNDMASK, SINK_F)
KERNEL(k_cndmask_s, DECL_F, B_CNDMASK_S, SINK_F)
KERNEL(k_dppmov, DECL_F, B_DPPMOV, SINK_F)
KERNEL(k_dppadd, DECL_F, B_DPPADD, SINK_F)
KERNEL(k_med3, DECL_F, B_MED3, SINK_F)
KERNEL(k_madu24, DECL_F, B_MADU24, SINK_F)
KERNEL(k_readlane, DECL_F, B_READLANE, SINK_F)
KERNEL(k_sand, DECL_F, B_SAND, SINK_F)
KERNEL(k_scsel, DECL_F, B_SCSEL, SINK_F)
KERNEL(k_snop, DECL_F, B_SNOP, SINK_F)

// The live-strip bodies of the two compositing kernels, instruction for instruction as the compiler emits them
// (forward.hip render_fwd_body / backward.hip render_bwd_body, objdump of the library): what ONE evaluated (entry, strip)
// pair costs when 7 / 6 waves per SIMD run nothing else.
#define B_FWD_STRIP(i)                                                                                                   \
    asm volatile(                                                                                                         \
        "v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n"                 \
        "v_fmac_f32 %0, -0.5, %2\n v_mul_f32 %0, 0x3fb8aa3b, %0\n v_rndne_f32 %0, %0\n v_sub_f32 %0, %0, %1\n"            \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_cvt_i32_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"         \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, 1.0\n v_ldexp_f32 %0, %0, 1\n v_mul_f32 %0, %1, %0\n"           \
        "v_min_f32 %0, 0x3f7d70a4, %0\n v_cmp_nlt_f32 vcc, 0, %0\n v_cmp_ngt_f32 s[20:21], %1, %0\n v_mul_f32 %0, %0, %1\n"\
        "s_and_b64 s[20:21], vcc, s[20:21]\n v_sub_f32 %0, %1, %0\n s_and_b64 s[22:23], s[20:21], s[24:25]\n"             \
        "v_cmp_gt_f32 vcc, %2, %0\n s_and_b64 s[26:27], vcc, s[22:23]\n s_andn2_b64 s[20:21], s[22:23], s[26:27]\n"       \
        "s_cselect_b64 s[22:23], s[28:29], 0\n s_or_b64 s[30:31], s[22:23], s[30:31]\n v_cndmask_b32 %0, 0, %0, s[20:21]\n"\
        "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"     \
        "v_sub_f32 %0, %0, %1\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n s_cselect_b64 s[28:29], s[28:29], 0\n"          \
        "s_and_b64 s[22:23], s[28:29], s[30:31]\n s_cmp_eq_u64 s[22:23], 0\n"                                              \
        : "+v"(a[i]) : "v"(b), "v"(c)                                                                                      \
        : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
KERNEL(k_fwd_strip, DECL_F, B_FWD_STRIP, SINK_F)


// Variants of the live-strip body: how VALU and SALU time combine (round 4: removing half-rate VALU instructions from
// the real kernel bought nothing, adding SALU instructions cost time).
#define B_FWD_STRIP_VALU(i)   /* the 28 VALU instructions alone */                                                      \
    asm volatile(                                                                                                         \
        "v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n"                 \
        "v_fmac_f32 %0, -0.5, %2\n v_mul_f32 %0, 0x3fb8aa3b, %0\n v_rndne_f32 %0, %0\n v_sub_f32 %0, %0, %1\n"            \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_cvt_i32_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"         \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, 1.0\n v_ldexp_f32 %0, %0, 1\n v_mul_f32 %0, %1, %0\n"           \
        "v_min_f32 %0, 0x3f7d70a4, %0\n v_cmp_nlt_f32 vcc, 0, %0\n v_cmp_ngt_f32 s[20:21], %1, %0\n v_mul_f32 %0, %0, %1\n"\
        "v_sub_f32 %0, %1, %0\n"             \
        "v_cmp_gt_f32 vcc, %2, %0\n"       \
        "v_cndmask_b32 %0, 0, %0, s[20:21]\n"\
        "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"     \
        "v_sub_f32 %0, %0, %1\n"          \
        : "+v"(a[i]) : "v"(b), "v"(c)                                                                                      \
        : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
KERNEL(k_fwd_strip_valu, DECL_F, B_FWD_STRIP_VALU, SINK_F)
#define B_FWD_STRIP_SALU(i)   /* its 10 SALU instructions alone */                                                      \
    asm volatile(                                                                                                         \
        "s_and_b64 s[20:21], vcc, s[20:21]\n s_and_b64 s[22:23], s[20:21], s[24:25]\n"             \
        "s_and_b64 s[26:27], vcc, s[22:23]\n s_andn2_b64 s[20:21], s[22:23], s[26:27]\n"       \
        "s_cselect_b64 s[22:23], s[28:29], 0\n s_or_b64 s[30:31], s[22:23], s[30:31]\n"\
        "s_andn2_b64 s[24:25], s[24:25], s[26:27]\n s_cselect_b64 s[28:29], s[28:29], 0\n"          \
        "s_and_b64 s[22:23], s[28:29], s[30:31]\n s_cmp_eq_u64 s[22:23], 0\n"                                              \
        : "+v"(a[i]) : "v"(b), "v"(c)                                                                                      \
        : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
KERNEL(k_fwd_strip_salu, DECL_F, B_FWD_STRIP_SALU, SINK_F)
#define B_FWD_STRIP_NOGUARD(i)   /* without v_min, the `power > 0` compare and its s_and: 26 VALU + 9 SALU */             \
    asm volatile(                                                                                                         \
        "v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n"                 \
        "v_fmac_f32 %0, -0.5, %2\n v_mul_f32 %0, 0x3fb8aa3b, %0\n v_rndne_f32 %0, %0\n v_sub_f32 %0, %0, %1\n"            \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_cvt_i32_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"         \
        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, 1.0\n v_ldexp_f32 %0, %0, 1\n v_mul_f32 %0, %1, %0\n"           \
        "v_cmp_ngt_f32 s[20:21], %1, %0\n v_mul_f32 %0, %0, %1\n"\
        "v_sub_f32 %0, %1, %0\n s_and_b64 s[22:23], s[20:21], s[24:25]\n"             \
        "v_cmp_gt_f32 vcc, %2, %0\n s_and_b64 s[26:27], vcc, s[22:23]\n s_andn2_b64 s[20:21], s[22:23], s[26:27]\n"       \
        "s_cselect_b64 s[22:23], s[28:29], 0\n s_or_b64 s[30:31], s[22:23], s[30:31]\n v_cndmask_b32 %0, 0, %0, s[20:21]\n"\
        "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"     \
        "v_sub_f32 %0, %0, %1\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n s_cselect_b64 s[28:29], s[28:29], 0\n"          \
        "s_and_b64 s[22:23], s[28:29], s[30:31]\n s_cmp_eq_u64 s[22:23], 0\n"                                              \
        : "+v"(a[i]) : "v"(b), "v"(c)                                                                                      \
        : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
KERNEL(k_fwd_strip_noguard, DECL_F, B_FWD_STRIP_NOGUARD, SINK_F)
#define B_FWD_STRIP_FAST(i)   /* v_exp_f32 instead of the polynomial: 20 VALU + 10 SALU */             \
    asm volatile(                                                                                                         \
        "v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n"                 \
        "v_fmac_f32 %0, -0.5, %2\n v_mul_f32 %0, 0x3fb8aa3b, %0\n v_exp_f32 %0, %0\n v_mul_f32 %0, %1, %0\n"           \
        "v_min_f32 %0, 0x3f7d70a4, %0\n v_cmp_nlt_f32 vcc, 0, %0\n v_cmp_ngt_f32 s[20:21], %1, %0\n v_mul_f32 %0, %0, %1\n"\
        "s_and_b64 s[20:21], vcc, s[20:21]\n v_sub_f32 %0, %1, %0\n s_and_b64 s[22:23], s[20:21], s[24:25]\n"             \
        "v_cmp_gt_f32 vcc, %2, %0\n s_and_b64 s[26:27], vcc, s[22:23]\n s_andn2_b64 s[20:21], s[22:23], s[26:27]\n"       \
        "s_cselect_b64 s[22:23], s[28:29], 0\n s_or_b64 s[30:31], s[22:23], s[30:31]\n v_cndmask_b32 %0, 0, %0, s[20:21]\n"\
        "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"     \
        "v_sub_f32 %0, %0, %1\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n s_cselect_b64 s[28:29], s[28:29], 0\n"          \
        "s_and_b64 s[22:23], s[28:29], s[30:31]\n s_cmp_eq_u64 s[22:23], 0\n"                                              \
        : "+v"(a[i]) : "v"(b), "v"(c)                                                                                      \
        : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
KERNEL(k_fwd_strip_fast, DECL_F, B_FWD_STRIP_FAST, SINK_F)
#define B_VAND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_VLSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define B_VADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_VMAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
// Does a wave64 VALU instruction whose EXEC mask has an all-zero half skip that half's pass on the SIMD-32 datapath?
#define DECL_F_LO32 DECL_F asm volatile("s_mov_b64 exec, 0xffffffff");
#define DECL_F_LO16 DECL_F asm volatile("s_mov_b64 exec, 0xffff");
#define DECL_F_EVEN DECL_F asm volatile("s_mov_b32 exec_lo, 0x55555555\n s_mov_b32 exec_hi, 0x55555555");
KERNEL(k_fma_lo32, DECL_F_LO32, B_FMA, SINK_F)
KERNEL(k_fma_lo16, DECL_F_LO16, B_FMA, SINK_F)
KERNEL(k_fma_even, DECL_F_EVEN, B_FMA, SINK_F)
KERNEL(k_cmp_lo32, DECL_F_LO32, B_CMPS, SINK_F)
KERNEL(k_exp_lo32, DECL_F_LO32, B_EXP, SINK_F)
KERNEL(k_vand, DECL_F, B_VAND, SINK_F)
KERNEL(k_vlshladd, DECL_F, B_VLSHLADD, SINK_F)
KERNEL(k_vaddu, DECL_F, B_VADDU, SINK_F)
KERNEL(k_vmax, DECL_F, B_VMAX, SINK_F)

typedef void (*kern_t)(int, float, float*, Stamp*);
struct Row { const char* name; kern_t k; int valu_per_body; int salu_per_body; const char* isa_token; };

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;                 // 8 workgroups of 256 threads per CU = 8 waves per SIMD
    float* out; Stamp* st;
    CK(hipMalloc(&out, 256)); CK(hipMalloc(&st, sizeof(Stamp) * blocks * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Row> rows = {
        {"v_fma_f32", k_fma, 1, 0, "v_fma_f32"}, {"v_add_f32", k_add, 1, 0, "v_add_f32"}, {"v_sub_f32", k_sub, 1, 0, "v_sub_f32"},
        {"v_mul_f32", k_mul, 1, 0, "v_mul_f32"}, {"v_min_f32", k_min, 1, 0, "v_min_f32"}, {"v_mov_b32", k_mov, 1, 0, "v_mov_b32"},
        {"v_med3_f32", k_med3, 1, 0, "v_med3_f32"}, {"v_mad_u32_u24", k_madu24, 1, 0, "v_mad_u32_u24"},
        {"v_rndne_f32", k_rndne, 1, 0, "v_rndne_f32"}, {"v_cvt_i32_f32", k_cvt, 1, 0, "v_cvt_i32_f32"},
        {"v_ldexp_f32", k_ldexp, 1, 0, "v_ldexp_f32"}, {"v_cmp_ge_f32 -> vcc", k_cmp, 1, 0, "v_cmp_ge_f32"},
        {"v_cmp_ge_f32 -> sgpr pair", k_cmps, 1, 0, "v_cmp_ge_f32"},         {"v_cndmask_b32 (sgpr pair)", k_cndmask_s, 1, 0, "v_cndmask_b32"},
        {"v_exp_f32", k_exp, 1, 0, "v_exp_f32"}, {"v_rcp_f32", k_rcp, 1, 0, "v_rcp_f32"},
        {"v_mov_b32_dpp quad_perm", k_dppmov, 1, 0, "v_mov_b32_dpp"}, {"v_add_f32_dpp quad_perm", k_dppadd, 1, 0, "v_add_f32_dpp"},
        {"v_readlane_b32", k_readlane, 1, 0, "v_readlane_b32"},
        {"s_and_b64", k_sand, 0, 1, "s_and_b64"}, {"s_cselect_b64", k_scsel, 0, 1, "s_cselect_b64"}, {"s_nop 0", k_snop, 0, 1, "s_nop"},
        {"v_fma_f32, EXEC = low 32 lanes", k_fma_lo32, 1, 0, "v_fma_f32"}, {"v_fma_f32, EXEC = low 16 lanes", k_fma_lo16, 1, 0, "v_fma_f32"},
        {"v_fma_f32, EXEC = even lanes", k_fma_even, 1, 0, "v_fma_f32"}, {"v_cmp_ge_f32 -> sgpr, EXEC = low 32", k_cmp_lo32, 1, 0, "v_cmp_ge_f32"},
        {"v_exp_f32, EXEC = low 32 lanes", k_exp_lo32, 1, 0, "v_exp_f32"},
        {"v_and_b32", k_vand, 1, 0, "v_and_b32"}, {"v_lshl_add_u32", k_vlshladd, 1, 0, "v_lshl_add_u32"},
        {"v_add_u32", k_vaddu, 1, 0, "v_add_u32"}, {"v_max_f32", k_vmax, 1, 0, "v_max_f32"},
        {"render_fwd live strip (28 VALU + 10 SALU)", k_fwd_strip, 28, 10, "v_ldexp_f32"},
        {"  its 28 VALU alone", k_fwd_strip_valu, 28, 0, "v_ldexp_f32"},
        {"  its 10 SALU alone", k_fwd_strip_salu, 0, 10, "s_cselect_b64"},
        {"  without min / power>0 guard (26 + 9)", k_fwd_strip_noguard, 26, 9, "v_ldexp_f32"},
        {"  with v_exp_f32 (20 + 10)", k_fwd_strip_fast, 20, 10, "v_exp_f32"},
    };
    printf("# MI355X instruction issue cost: %d CUs x 4 SIMDs, 8 waves per SIMD, %d x %d independent instructions per trip\n", cus, UNROLL, NI);
    printf("# every row: >= 50 ms warm-up + one launch of >= 20 ms; clock = s_memtime delta of the waves / HIP-event wall time\n");
    printf("%-44s %9s %9s %10s %12s %12s\n", "instruction", "ms", "eff_GHz", "ns/instr", "cyc/instr", "cyc/body");
    for (const Row& r : rows) {
        // calibrate: a short launch, then scale the trip count to ~25 ms
        int trips = 64;
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); r.k<<<blocks, 256>>>(trips, 1.001f, out, st); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms > 1.0f) break;
            trips *= 8;
        }
        trips = (int)(trips * 25.0f / ms) + 1;
        for (int w = 0; w < 3; ++w) r.k<<<blocks, 256>>>(trips, 1.001f, out, st);          // >= 50 ms warm-up (3 x 25 ms)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); r.k<<<blocks, 256>>>(trips, 1.001f, out, st); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> h(blocks * 4);
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        // every wave runs the whole loop and all of them are resident together: a wave's own s_memtime delta spans the
        // launch (the counters of different XCDs are not synchronised, so stamps of different waves are never compared)
        unsigned long long dmax = 0;
        for (const Stamp& s : h) { if (s.c1 - s.c0 > dmax) dmax = s.c1 - s.c0; }
        const double cycles = (double)dmax;
        const double ghz = cycles / (ms * 1e6);
        const double bodies_per_simd = (double)trips * UNROLL * NI * 8.0;      // 8 waves per SIMD
        const double per_body_ns = ms * 1e6 / bodies_per_simd;
        const int v = r.valu_per_body, s = r.salu_per_body;
        const double instr = (double)(v + s);
        printf("%-44s %9.2f %9.3f %10.3f %12.3f %12.3f\n", r.name, ms, ghz, per_body_ns / instr, per_body_ns * ghz / instr,
               per_body_ns * ghz);
    }
    return 0;
}
