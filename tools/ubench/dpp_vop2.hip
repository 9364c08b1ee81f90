// tools/ubench/dpp_vop2.hip -- does a VOP2 instruction that takes its src0 through the DPP wave shift give what a v_mov_b32_dpp copy followed by the plain
// instruction gives?  (The step of the DP fill copies the vector that moves with the step; folding the shift into the three readers of the copy would take one VALU
// instruction out of 33.)  Measured on gfx950: v_add_u32_dpp and v_sub_u32_dpp (dpp(src0) - src1) do; v_subrev_u32_dpp does NOT compute src1 - dpp(src0): it computes
// dpp(src1) - src0 -- the lane shift lands on the minuend in both forms, so "x - shifted(y)" has no single instruction, the fill needs it once per direction
// (gfv - shr(dh) going right, df - shl(dv) going down), and the copy stays.   hipcc --offload-arch=gfx950 -O3 -o dpp_vop2 tools/ubench/dpp_vop2.hip && ./dpp_vop2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
__global__ void k(uint32_t *out)
{
	const uint32_t l = threadIdx.x; uint32_t a = l * 3 + 1, b = l * 100 + 7, x, r;
	asm volatile("v_mov_b32_dpp %0, %1" SHR "\n\ts_nop 1" : "=&v"(x) : "v"(a)); out[0 * 64 + l] = x + b;
	asm volatile("v_add_u32_dpp %0, %1, %2" SHR "\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(b)); out[1 * 64 + l] = r;
	out[2 * 64 + l] = b - x;
	asm volatile("v_subrev_u32_dpp %0, %1, %2" SHR "\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(b)); out[3 * 64 + l] = r;
	asm volatile("v_mov_b32_dpp %0, %1" SHL "\n\ts_nop 1" : "=&v"(x) : "v"(a)); out[4 * 64 + l] = x + b;
	asm volatile("v_add_u32_dpp %0, %1, %2" SHL "\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(b)); out[5 * 64 + l] = r;
	out[6 * 64 + l] = b - x;
	asm volatile("v_subrev_u32_dpp %0, %1, %2" SHL "\n\ts_nop 1" : "=&v"(r) : "v"(a), "v"(b)); out[7 * 64 + l] = r;
	uint32_t c = a; asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %1" SHR "\n\ts_nop 1" : "+v"(c) : "v"(b)); out[8 * 64 + l] = c;          /* in place: against row 0 */
	uint32_t d = b; asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %0" SHR "\n\ts_nop 1" : "+v"(d) : "v"(a)); out[9 * 64 + l] = d;          /* dst = src1: against row 0 */
}
int main()
{
	uint32_t *d, h[10 * 64]; if(hipMalloc(&d, sizeof(h)) != hipSuccess) return 1; hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); if(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
	const int pairs[6][2] = { {0, 1}, {2, 3}, {4, 5}, {6, 7}, {0, 8}, {0, 9} };
	const char *nm[6] = { "v_add_u32_dpp shr", "v_subrev_u32_dpp shr", "v_add_u32_dpp shl", "v_subrev_u32_dpp shl", "v_add_u32_dpp shr, dst = src0", "v_add_u32_dpp shr, dst = src1" };
	for(int p = 0; p < 6; p++) { int bad = 0, first = -1; for(int l = 0; l < 64; l++) if(h[pairs[p][0] * 64 + l] != h[pairs[p][1] * 64 + l]) { if(first < 0) first = l; bad++; }
		printf("%-34s %2d lanes differ from copy + plain op", nm[p], bad); if(bad) printf(" (first: lane %d: %u against %u)", first, h[pairs[p][1] * 64 + first], h[pairs[p][0] * 64 + first]); printf("\n"); }
	return 0;
}
