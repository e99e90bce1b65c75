// Issue rate of the vector instructions the FAST kernel is made of: cycles per wave-instruction on one
// SIMD with 1, 2 and 4 waves per SIMD resident.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x12345678u, c = a + 77u, d = b + 99u;
    unsigned e = a ^ 0x5555u, f = b ^ 0xAAAAu, g = c ^ 0x3333u, h = d ^ 0xCCCCu;
    unsigned x = 0x07030501u + (threadIdx.x & 1), y = 0x01020304u + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        // eight independent chains, 16 x 8 = 128 instructions per iteration
        if (OP == 0) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 1) { REP16(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 2) { REP16(asm volatile("v_max3_u32 %0, %0, %8, %9\n v_max3_u32 %1, %1, %8, %9\n v_max3_u32 %2, %2, %8, %9\n v_max3_u32 %3, %3, %8, %9\n v_max3_u32 %4, %4, %8, %9\n v_max3_u32 %5, %5, %8, %9\n v_max3_u32 %6, %6, %8, %9\n v_max3_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 3) { REP16(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 6) { REP16(asm volatile("v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 7) { REP16(asm volatile("v_and_b32_e64 %0, %0, %8\n v_and_b32_e64 %1, %1, %8\n v_and_b32_e64 %2, %2, %8\n v_and_b32_e64 %3, %3, %8\n v_and_b32_e64 %4, %4, %8\n v_and_b32_e64 %5, %5, %8\n v_and_b32_e64 %6, %6, %8\n v_and_b32_e64 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 8) { REP16(asm volatile("v_min_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_min_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 9) { REP16(asm volatile("v_add_u32_e32 %0, 0x12345, %0\n v_add_u32_e32 %1, 0x12345, %1\n v_add_u32_e32 %2, 0x12345, %2\n v_add_u32_e32 %3, 0x12345, %3\n v_add_u32_e32 %4, 0x12345, %4\n v_add_u32_e32 %5, 0x12345, %5\n v_add_u32_e32 %6, 0x12345, %6\n v_add_u32_e32 %7, 0x12345, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 10) { REP16(asm volatile("v_max_u32_e32 %0, %8, %0\n v_max_u32_e32 %1, %8, %1\n v_max_u32_e32 %2, %8, %2\n v_max_u32_e32 %3, %8, %3\n v_max_u32_e32 %4, %8, %4\n v_max_u32_e32 %5, %8, %5\n v_max_u32_e32 %6, %8, %6\n v_max_u32_e32 %7, %8, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 11) { REP16(asm volatile("v_add_u32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 12) { REP16(asm volatile("v_min_u16_e32 %0, %8, %0\n v_min_u16_e32 %1, %8, %1\n v_min_u16_e32 %2, %8, %2\n v_min_u16_e32 %3, %8, %3\n v_min_u16_e32 %4, %8, %4\n v_min_u16_e32 %5, %8, %5\n v_min_u16_e32 %6, %8, %6\n v_min_u16_e32 %7, %8, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 13) { REP16(asm volatile("v_lshrrev_b32_e32 %0, 8, %0\n v_lshrrev_b32_e32 %1, 8, %1\n v_lshrrev_b32_e32 %2, 8, %2\n v_lshrrev_b32_e32 %3, 8, %3\n v_lshrrev_b32_e32 %4, 8, %4\n v_lshrrev_b32_e32 %5, 8, %5\n v_lshrrev_b32_e32 %6, 8, %6\n v_lshrrev_b32_e32 %7, 8, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 14) { REP16(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 15) { REP16(asm volatile("v_bfi_b32 %0, %0, %8, %9\n v_bfi_b32 %1, %1, %8, %9\n v_bfi_b32 %2, %2, %8, %9\n v_bfi_b32 %3, %3, %8, %9\n v_bfi_b32 %4, %4, %8, %9\n v_bfi_b32 %5, %5, %8, %9\n v_bfi_b32 %6, %6, %8, %9\n v_bfi_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 16) { REP16(asm volatile("v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %8, %9\n v_or3_b32 %5, %5, %8, %9\n v_or3_b32 %6, %6, %8, %9\n v_or3_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 17) { REP16(asm volatile("v_lshl_or_b32 %0, %0, %8, %9\n v_lshl_or_b32 %1, %1, %8, %9\n v_lshl_or_b32 %2, %2, %8, %9\n v_lshl_or_b32 %3, %3, %8, %9\n v_lshl_or_b32 %4, %4, %8, %9\n v_lshl_or_b32 %5, %5, %8, %9\n v_lshl_or_b32 %6, %6, %8, %9\n v_lshl_or_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 18) { REP16(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 19) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 20) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %8, %9\n v_alignbyte_b32 %1, %1, %8, %9\n v_alignbyte_b32 %2, %2, %8, %9\n v_alignbyte_b32 %3, %3, %8, %9\n v_alignbyte_b32 %4, %4, %8, %9\n v_alignbyte_b32 %5, %5, %8, %9\n v_alignbyte_b32 %6, %6, %8, %9\n v_alignbyte_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 21) { REP16(asm volatile("v_alignbit_b32 %0, %0, %8, %9\n v_alignbit_b32 %1, %1, %8, %9\n v_alignbit_b32 %2, %2, %8, %9\n v_alignbit_b32 %3, %3, %8, %9\n v_alignbit_b32 %4, %4, %8, %9\n v_alignbit_b32 %5, %5, %8, %9\n v_alignbit_b32 %6, %6, %8, %9\n v_alignbit_b32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 22) { REP16(asm volatile("v_bfe_u32 %0, %0, %8, %9\n v_bfe_u32 %1, %1, %8, %9\n v_bfe_u32 %2, %2, %8, %9\n v_bfe_u32 %3, %3, %8, %9\n v_bfe_u32 %4, %4, %8, %9\n v_bfe_u32 %5, %5, %8, %9\n v_bfe_u32 %6, %6, %8, %9\n v_bfe_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 23) { REP16(asm volatile("v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 24) { REP16(asm volatile("v_sad_u8 %0, %0, %8, %9\n v_sad_u8 %1, %1, %8, %9\n v_sad_u8 %2, %2, %8, %9\n v_sad_u8 %3, %3, %8, %9\n v_sad_u8 %4, %4, %8, %9\n v_sad_u8 %5, %5, %8, %9\n v_sad_u8 %6, %6, %8, %9\n v_sad_u8 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 25) { REP16(asm volatile("v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %8, %9\n v_xad_u32 %5, %5, %8, %9\n v_xad_u32 %6, %6, %8, %9\n v_xad_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 26) { REP16(asm volatile("v_med3_u32 %0, %0, %8, %9\n v_med3_u32 %1, %1, %8, %9\n v_med3_u32 %2, %2, %8, %9\n v_med3_u32 %3, %3, %8, %9\n v_med3_u32 %4, %4, %8, %9\n v_med3_u32 %5, %5, %8, %9\n v_med3_u32 %6, %6, %8, %9\n v_med3_u32 %7, %7, %8, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 27) { REP16(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 28) { REP16(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 29) { REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 30) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 31) { REP16(asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 32) { REP16(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 33) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 34) { REP16(asm volatile("v_lshlrev_b32 %0, %0, %8\n v_lshlrev_b32 %1, %1, %8\n v_lshlrev_b32 %2, %2, %8\n v_lshlrev_b32 %3, %3, %8\n v_lshlrev_b32 %4, %4, %8\n v_lshlrev_b32 %5, %5, %8\n v_lshlrev_b32 %6, %6, %8\n v_lshlrev_b32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 35) { REP16(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 36) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 37) { REP16(asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 38) { REP16(asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3\n v_mov_b32 %4, %4\n v_mov_b32 %5, %5\n v_mov_b32 %6, %6\n v_mov_b32 %7, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 39) { REP16(asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3\n v_not_b32 %4, %4\n v_not_b32 %5, %5\n v_not_b32 %6, %6\n v_not_b32 %7, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 40) { REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 41) { REP16(asm volatile("v_dot2_u32_u16 %0, %8, %9, %0\n v_dot2_u32_u16 %1, %8, %9, %1\n v_dot2_u32_u16 %2, %8, %9, %2\n v_dot2_u32_u16 %3, %8, %9, %3\n v_dot2_u32_u16 %4, %8, %9, %4\n v_dot2_u32_u16 %5, %8, %9, %5\n v_dot2_u32_u16 %6, %8, %9, %6\n v_dot2_u32_u16 %7, %8, %9, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 42) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "s20", "s21");) }
        if (OP == 43) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP16(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 44) { REP16(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_add_u32 %0, %0, %9\n v_cmp_lt_u32_e64 s[20:21], %1, %8\n v_add_u32 %1, %1, %9\n v_cmp_lt_u32_e64 s[20:21], %2, %8\n v_add_u32 %2, %2, %9\n v_cmp_lt_u32_e64 s[20:21], %3, %8\n v_add_u32 %3, %3, %9\n v_cmp_lt_u32_e64 s[20:21], %4, %8\n v_add_u32 %4, %4, %9\n v_cmp_lt_u32_e64 s[20:21], %5, %8\n v_add_u32 %5, %5, %9\n v_cmp_lt_u32_e64 s[20:21], %6, %8\n v_add_u32 %6, %6, %9\n v_cmp_lt_u32_e64 s[20:21], %7, %8\n v_add_u32 %7, %7, %9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "s20", "s21");) }
        if (OP == 45) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 46) { REP16(asm volatile("v_cmp_lt_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %9, vcc\n v_cmp_lt_u32_e32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %9, vcc\n v_cmp_lt_u32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %9, vcc\n v_cmp_lt_u32_e32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %9, vcc\n v_cmp_lt_u32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %9, vcc\n v_cmp_lt_u32_e32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %9, vcc\n v_cmp_lt_u32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %9, vcc\n v_cmp_lt_u32_e32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %9, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
        if (OP == 47) { REP16(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %1, %8\n v_cndmask_b32_e64 %1, %1, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %3, %8\n v_cndmask_b32_e64 %3, %3, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %5, %8\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %6, %8\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %7, %8\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "s20", "s21");) }
        if (OP == 48) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));) }
        if (OP == 49) { REP16(asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %8, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %8, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %8, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %8, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %8, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %8, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %8, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc");) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
    }
}

template <int OP>
void run(const char* name) {
    unsigned* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1 << 24);
    hipMalloc(&cyc, 8 * 4096);
    const int iters = 200;
    for (int waves_per_simd : {2, 8}) {
        // one workgroup of 4 * waves_per_simd waves per CU, 256 CUs
        const int threads = 64 * 4 * waves_per_simd;
        const int blocks = 256;
        dim3 b(threads > 1024 ? 1024 : threads), g(threads > 1024 ? blocks * (threads / 1024) : blocks);
        hipLaunchKernelGGL(k<OP>, g, b, 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, g, b, 0, 0, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(g.x);
        hipMemcpy(h.data(), cyc, 8 * g.x, hipMemcpyDeviceToHost);
        const double per_wave_instr = (double)h[0] / (iters * 128.0);
        // SIMD-cycles per wave-instruction assuming 2.4 GHz: total wave-instructions per SIMD = waves_per_simd * iters * 128
        const double ns_per = ms * 1e6 / (waves_per_simd * iters * 128.0);
        printf("%-14s waves/SIMD %d: %.2f memtime ticks per instruction of one wave, %.3f ns of SIMD time per wave-instruction (%.2f cycles at 2.4 GHz)\n",
               name, waves_per_simd, per_wave_instr, ns_per, ns_per * 2.4);
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<3>("v_and_b32");
    run<4>("v_add_u32");
    run<5>("v_fma_f32");
    run<0>("v_perm_b32");
    run<1>("v_pk_min_u16");
    run<2>("v_max3_u32");
    run<6>("v_dot4_u32_u8");
    run<7>("v_and_b32_e64");
    run<8>("v_min_u32_sdwa");
    run<9>("v_add_u32+lit");
    run<10>("v_max_u32_e32");
    run<11>("v_add_u32_dpp");
    run<12>("v_min_u16_e32");
    run<13>("v_lshrrev_e32");
    run<14>("v_and_or_b32");
    run<15>("v_bfi_b32");
    run<16>("v_or3_b32");
    run<17>("v_lshl_or_b32");
    run<18>("v_add3_u32");
    run<19>("v_mad_u32_u24");
    run<20>("v_alignbyte_b32");
    run<21>("v_alignbit_b32");
    run<22>("v_bfe_u32");
    run<23>("v_lshl_add_u32");
    run<24>("v_sad_u8");
    run<25>("v_xad_u32");
    run<26>("v_med3_u32");
    run<27>("v_sub_u32");
    run<28>("v_or_b32");
    run<29>("v_xor_b32");
    run<30>("v_mul_u32_u24");
    run<31>("v_bcnt_u32_b32");
    run<32>("v_pk_add_u16");
    run<33>("v_pk_mul_lo_u16");
    run<34>("v_lshlrev_b32");
    run<35>("v_min_u32");
    run<36>("v_cndmask_b32");
    run<37>("v_ffbl_b32");
    run<38>("v_mov_b32");
    run<39>("v_not_b32");
    run<40>("v_mov_dpp_wave_shr");
    run<41>("v_dot2_u32_u16");
    run<42>("v_cndmask_e64_sgpr");
    run<43>("v_cndmask_vcc_preset");
    run<44>("v_cmp_e64+v_add (pairs)");
    run<45>("v_mul_lo_u32");
    run<46>("v_cmp_e32 vcc + v_cndmask_e32 vcc (pairs)");
    run<47>("v_cmp_e64 sgpr + v_cndmask_e64 sgpr (pairs)");
    run<48>("v_cndmask_e64 with vcc");
    run<49>("v_addc_co_u32 vcc");
    return 0;
}
