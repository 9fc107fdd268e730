#!/bin/bash
# Every instruction tools/ubench/valu_rate.hip claims to time is in its device code (counts per mnemonic in the loop bodies).
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only valu_rate.hip -o /tmp/valu_rate.s 2>/dev/null
for m in v_fma_f32 v_add_f32 v_sub_f32 v_mul_f32 v_min_f32 v_mov_b32 v_med3_f32 v_mad_u32_u24 v_rndne_f32 v_cvt_i32_f32 \
         v_ldexp_f32 v_cmp_ge_f32 v_cndmask_b32 v_exp_f32 v_rcp_f32 v_mov_b32_dpp v_add_f32_dpp v_readlane_b32 s_and_b64 \
         s_cselect_b64 s_nop; do
  printf "%-16s %s\n" $m "$(grep -c "^\s*$m" /tmp/valu_rate.s)"
done
