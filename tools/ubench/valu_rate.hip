// Instruction issue-rate micro-benchmark for gfx950 (tools/ubench/valu_rate.hip): one kernel per instruction kind, 4 waves per
// SIMD, 8 independent destination registers so that only issue throughput is measured.  Prints SIMD cycles per wave-instruction
// relative to the wall clock of the launch (s_memtime ticks of wave 0 as the cycle count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REPT 64
#define ITER 2000
#define KERNEL(name, body) \
__global__ void __launch_bounds__(256, 4) k_##name(unsigned long long *out, int n) { \
	int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, a = v0 * 3, b = v0 * 5; \
	unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for(int i = 0; i < n; i++) { \
		asm volatile(".rept 8\n\t" body "\n\t.endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "m0"); \
	} \
	unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	if(threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; } \
	if(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 0x12345) { out[1] = 1; } \
}
// each body = 8 instructions on v0..v7 (%0..%7), inputs %8 %9
#define B8(ins, tail) ins " %0, " tail "\n\t" ins " %1, " tail "\n\t" ins " %2, " tail "\n\t" ins " %3, " tail "\n\t" ins " %4, " tail "\n\t" ins " %5, " tail "\n\t" ins " %6, " tail "\n\t" ins " %7, " tail
KERNEL(add, B8("v_add_u32", "%8, %9"))
KERNEL(add_sdwa, B8("v_add_u32_sdwa", "%8, %9 dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD"))
KERNEL(max3, B8("v_max3_i32", "%8, %9, %8"))
KERNEL(perm, B8("v_perm_b32", "%8, %9, %8"))
KERNEL(subclamp, B8("v_sub_i32", "%8, %9 clamp"))
KERNEL(dpp_wshr, B8("v_mov_b32_dpp", "%8 wave_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(dpp_wshr_bc, B8("v_mov_b32_dpp", "%8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"))
KERNEL(dpp_rshr, B8("v_mov_b32_dpp", "%8 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(cmp_vcc, "v_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9\n\tv_cmp_eq_u32 vcc, %8, %9")
KERNEL(cmp_sgpr, "v_cmp_eq_u32 s[40:41], %8, %9\n\tv_cmp_eq_u32 s[42:43], %8, %9\n\tv_cmp_eq_u32 s[44:45], %8, %9\n\tv_cmp_eq_u32 s[46:47], %8, %9\n\tv_cmp_eq_u32 s[40:41], %8, %9\n\tv_cmp_eq_u32 s[42:43], %8, %9\n\tv_cmp_eq_u32 s[44:45], %8, %9\n\tv_cmp_eq_u32 s[46:47], %8, %9")
KERNEL(addc_sgpr, B8("v_addc_co_u32", "vcc, %8, %9, s[40:41]"))
KERNEL(readlane, "v_readlane_b32 s40, %8, 0\n\tv_readlane_b32 s41, %8, 5\n\tv_readlane_b32 s42, %8, 9\n\tv_readlane_b32 s43, %8, 63\n\tv_readlane_b32 s44, %8, 0\n\tv_readlane_b32 s45, %8, 5\n\tv_readlane_b32 s46, %8, 9\n\tv_readlane_b32 s47, %8, 63")
KERNEL(writelane, B8("v_writelane_b32", "s4, 3"))
KERNEL(readfirst, "v_readfirstlane_b32 s40, %8\n\tv_readfirstlane_b32 s41, %9\n\tv_readfirstlane_b32 s42, %8\n\tv_readfirstlane_b32 s43, %9\n\tv_readfirstlane_b32 s44, %8\n\tv_readfirstlane_b32 s45, %9\n\tv_readfirstlane_b32 s46, %8\n\tv_readfirstlane_b32 s47, %9")
KERNEL(addc_vcc, B8("v_addc_co_u32_e32", "vcc, %8, %9, vcc"))
KERNEL(cndmask_vcc, B8("v_cndmask_b32_e32", "%8, %9, vcc"))
KERNEL(bfe, B8("v_bfe_i32", "%8, 0, 8"))
KERNEL(max_e32, B8("v_max_i32_e32", "%8, %9"))
KERNEL(and_e32, B8("v_and_b32_e32", "%8, %9"))
KERNEL(lshl_e32, B8("v_lshlrev_b32_e32", "3, %9"))
KERNEL(add3, B8("v_add3_u32", "%8, %9, %8"))
KERNEL(pk_add, B8("v_pk_add_i16", "%8, %9"))
KERNEL(pk_max, B8("v_pk_max_i16", "%8, %9"))
KERNEL(subco, B8("v_sub_co_u32_e32", "vcc, %8, %9"))
KERNEL(mov, B8("v_mov_b32_e32", "%8"))
KERNEL(add_sgpr, B8("v_add_u32_e32", "s4, %9"))
KERNEL(max_sdwa, B8("v_max_i32_sdwa", "sext(%8), sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"))
KERNEL(bperm, B8("ds_bpermute_b32", "%8, %9"))
KERNEL(salu_or64, "s_or_b64 s[40:41], s[42:43], s[44:45]\n\ts_or_b64 s[42:43], s[40:41], s[44:45]\n\ts_or_b64 s[46:47], s[42:43], s[44:45]\n\ts_or_b64 s[40:41], s[42:43], s[44:45]\n\ts_or_b64 s[42:43], s[40:41], s[44:45]\n\ts_or_b64 s[46:47], s[42:43], s[44:45]\n\ts_or_b64 s[40:41], s[42:43], s[44:45]\n\ts_or_b64 s[42:43], s[40:41], s[44:45]")
KERNEL(mix_valu_salu, "v_add_u32 %0, %8, %9\n\ts_or_b64 s[40:41], s[42:43], s[44:45]\n\tv_add_u32 %1, %8, %9\n\ts_or_b64 s[42:43], s[46:47], s[44:45]\n\tv_add_u32 %2, %8, %9\n\ts_or_b64 s[46:47], s[42:43], s[44:45]\n\tv_add_u32 %3, %8, %9\n\ts_or_b64 s[40:41], s[42:43], s[44:45]")
KERNEL(dep_chain, "v_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9\n\tv_add_u32 %0, %0, %9")
KERNEL(dep_chain_sdwa, "v_add_u32_sdwa %0, %0, %9 dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\tv_add_u32 %1, %1, %9\n\tv_add_u32_sdwa %0, %0, %9 dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\tv_add_u32 %1, %1, %9\n\tv_add_u32_sdwa %0, %0, %9 dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\tv_add_u32 %1, %1, %9\n\tv_add_u32_sdwa %0, %0, %9 dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\tv_add_u32 %1, %1, %9")
#define RUN(name, per_rept) { \
	hipMemset(d, 0, 16); hipEventRecord(e0); \
	hipLaunchKernelGGL(k_##name, dim3(grid), dim3(256), 0, 0, d, ITER); \
	hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
	unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
	double n_ins = (double)ITER * 8 * per_rept; \
	printf("%-16s wave0 ticks/instr %.2f   (4 waves/SIMD -> SIMD cycles per wave-instr %.2f)   launch %.3f ms\n", #name, h[0] / n_ins, h[0] / n_ins / 4.0, ms); }
int main() {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	int grid = p.multiProcessorCount * 4;
	unsigned long long *d; hipMalloc(&d, 16);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	RUN(add, 8) RUN(add, 8) RUN(add_sdwa, 8) RUN(max3, 8) RUN(perm, 8) RUN(subclamp, 8) RUN(dpp_wshr, 8) RUN(dpp_wshr_bc, 8) RUN(dpp_rshr, 8)
	RUN(cmp_vcc, 8) RUN(cmp_sgpr, 8) RUN(addc_sgpr, 8) RUN(readlane, 8) RUN(writelane, 8) RUN(readfirst, 8) RUN(addc_vcc, 8) RUN(cndmask_vcc, 8) RUN(bfe, 8) RUN(max_e32, 8) RUN(and_e32, 8) RUN(lshl_e32, 8) RUN(add3, 8) RUN(pk_add, 8) RUN(pk_max, 8) RUN(subco, 8) RUN(mov, 8) RUN(add_sgpr, 8) RUN(max_sdwa, 8) RUN(bperm, 8) RUN(salu_or64, 8) RUN(mix_valu_salu, 8) RUN(dep_chain, 8) RUN(dep_chain_sdwa, 8)
	return 0;
}
