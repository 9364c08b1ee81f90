// tools/ubench/mask_store.hip -- what would the traceback masks of the DP fill cost as SCALAR stores?  (the one change left that takes VALU
// instructions out of the step.)  The fill keeps four bit columns per lane and adds one bit per vector to each with v_addc_co_u32 (4 of the 33 VALU of a traced
// step); the compares that make the bits leave them in SGPR pairs, i.e. as the 64-bit lane masks the reference itself stores (gaba.c:308-315).  Variant B stores those
// pairs with s_store_dwordx2 instead -- no VALU at all -- at the price of four SMEM instructions per vector and of a block layout the traceback would have to follow.
// This kernel has the shape of the fill (8 waves per SIMD, 32 vectors per block, ~27 VALU of filler per vector with the compares in place) and runs both variants:
//   hipcc --offload-arch=gfx950 -O3 -o mask_store tools/ubench/mask_store.hip && ./mask_store
// prints ns per vector per wave for A (addc + one 1 KB store per block) and B (4 scalar stores per vector), and checks that what B wrote is what A accumulated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define BLOCKS_PER_WAVE 256
#define BLK_STRIDE 1296          /* bytes between the blocks of a wave's workspace */
#define FILLER \
	"v_add_u32 %[x0], %[x0], %[a]\n\t v_max_i32 %[x1], %[x1], %[x0]\n\t v_add_u32 %[x2], %[x2], %[x1]\n\t v_sub_u32 %[x3], %[x3], %[x2]\n\t" \
	"v_max3_i32 %[x0], %[x0], %[x3], %[b]\n\t v_add_u32 %[x1], %[x1], %[b]\n\t v_perm_b32 %[x2], %[x2], %[x0], %[a]\n\t v_add_u32 %[x3], %[x3], %[x1]\n\t" \
	"v_mov_b32_dpp %[x0], %[x0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t v_mov_b32_dpp %[x1], %[x1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
	"v_add_u32 %[x2], %[x2], %[x3]\n\t v_max_i32 %[x3], %[x3], %[x0]\n\t v_add_u32 %[x0], %[x0], %[x2]\n\t v_sub_i32 %[x1], %[x1], %[x3] clamp\n\t" \
	"v_add_u32 %[x2], %[x2], %[a]\n\t v_add_u32 %[x3], %[x3], %[b]\n\t v_max_i32 %[x0], %[x0], %[x1]\n\t"
#define COMPARES \
	"v_cmp_lt_i32 %[A], %[x0], %[x1]\n\t v_cmp_lt_i32 %[B], %[x1], %[x2]\n\t v_cmp_lt_i32 %[C], %[x2], %[x3]\n\t v_cmp_lt_i32 %[D], %[x3], %[x0]\n\t" \
	"v_cmp_ge_i32 %[E], %[x0], %[x2]\n\t v_cmp_ge_i32 %[F], %[x1], %[x3]\n\t s_or_b64 %[A], %[A], %[E]\n\t s_or_b64 %[C], %[C], %[F]\n\t"

template<int VARIANT, int FRESH>
__global__ void __launch_bounds__(256, 2) k_fill(uint32_t *lane_major, uint64_t *vec_major, unsigned long long *ticks, int n_blocks)
{
	const int lane = threadIdx.x & 63; const uint32_t wave = blockIdx.x * 4 + threadIdx.x / 64;
	int x0 = lane * 7 + wave, x1 = lane * 13 + 5, x2 = lane ^ 0x55, x3 = wave * 3 + lane, a = lane + 1, b = 3 - lane;
	uint32_t mh = 0, mv = 0, me = 0, mf = 0;
	const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
	const uint64_t span = FRESH ? (uint64_t)BLOCKS_PER_WAVE * BLK_STRIDE : BLK_STRIDE;          /* a wave's workspace: every block in new memory, or one block overwritten */
	uint8_t *ws = (uint8_t *)vec_major + (uint64_t)wave_u * span;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for(int blk = 0; blk < n_blocks; blk++) {
		mh = mv = me = mf = 0;
		uint8_t *blkp = ws + (FRESH ? (uint64_t)blk * BLK_STRIDE : 0);
		uint64_t *vmk = (uint64_t *)blkp; uint32_t *lm = (uint32_t *)blkp;
		for(int k = 0; k < 32; k++) {
			uint64_t A, B, C, D, E, F;
			if(VARIANT == 0) {
				asm volatile(FILLER COMPARES
					"v_addc_co_u32 %[mh], vcc, %[mh], %[mh], %[A]\n\t v_addc_co_u32 %[mv], vcc, %[mv], %[mv], %[B]\n\t v_addc_co_u32 %[me], vcc, %[me], %[me], %[C]\n\t v_addc_co_u32 %[mf], vcc, %[mf], %[mf], %[D]\n\t"
					: [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [mh] "+v"(mh), [mv] "+v"(mv), [me] "+v"(me), [mf] "+v"(mf),
					  [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [E] "=&s"(E), [F] "=&s"(F)
					: [a] "v"(a), [b] "v"(b) : "vcc", "scc");
			} else if(VARIANT == 1 || VARIANT == 2) {
				asm volatile(FILLER COMPARES
					"s_store_dwordx2 %[A], %[vm], 0x0\n\t s_store_dwordx2 %[B], %[vm], 0x8\n\t s_store_dwordx2 %[C], %[vm], 0x10\n\t s_store_dwordx2 %[D], %[vm], 0x18\n\t"
					: [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3),
					  [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [E] "=&s"(E), [F] "=&s"(F)
					: [a] "v"(a), [b] "v"(b), [vm] "s"(vmk) : "vcc", "scc", "memory");
				if(VARIANT == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
				vmk += 4;
			} else {
				asm volatile(FILLER COMPARES
					"s_mov_b64 s[40:41], %[A]\n\t s_mov_b64 s[42:43], %[B]\n\t s_mov_b64 s[44:45], %[C]\n\t s_mov_b64 s[46:47], %[D]\n\t"
					"s_store_dwordx4 s[40:43], %[vm], 0x0\n\t s_store_dwordx4 s[44:47], %[vm], 0x10\n\t"
					: [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3),
					  [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [E] "=&s"(E), [F] "=&s"(F)
					: [a] "v"(a), [b] "v"(b), [vm] "s"(vmk) : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
				if(VARIANT == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
				vmk += 4;
			}
		}
		if(VARIANT == 0) { lm[0 * 64 + lane] = mh; lm[1 * 64 + lane] = mv; lm[2 * 64 + lane] = me; lm[3 * 64 + lane] = mf; }
	}
	if(VARIANT != 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\t s_dcache_wb\n\t s_waitcnt lgkmcnt(0)" ::: "memory"); }
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if(lane == 0) { atomicMax(&ticks[VARIANT & 3], t1 - t0); }
	if(x0 + x1 + x2 + x3 == 0x7fffffff) { ticks[7] = 1; }
}

template<int V, int FR> static float run(int grid, uint32_t *lm, uint64_t *vm, unsigned long long *ticks)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
	for(int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL((k_fill<V, FR>), dim3(grid), dim3(256), 0, 0, lm, vm, ticks, BLOCKS_PER_WAVE); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
	return ms;
}
int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, grid = cus * 8, waves = grid * 4;          /* 8 waves per SIMD */
	uint32_t *lm; uint64_t *vm; unsigned long long *ticks;
	const size_t ws_bytes = (size_t)waves * BLOCKS_PER_WAVE * BLK_STRIDE + 4096;
	hipMalloc(&lm, 4096); hipMalloc(&vm, ws_bytes); hipMalloc(&ticks, 64); hipMemset(ticks, 0, 64); hipMemset(vm, 0, ws_bytes);
	const double vec = (double)waves * BLOCKS_PER_WAVE * 32;
	printf("%d CUs, %d waves, %d blocks of 32 vectors each; G vectors/s, one block overwritten | every block in new memory (stride %d B)\n", cus, waves, BLOCKS_PER_WAVE, BLK_STRIDE);
	const float a0 = run<0, 0>(grid, lm, vm, ticks), a1 = run<0, 1>(grid, lm, vm, ticks);
	printf("A  v_addc_co x 4 per vector + 1 KB per block        : %6.2f | %6.2f\n", vec / a0 * 1e-6, vec / a1 * 1e-6);
	{ const float b0 = run<1, 0>(grid, lm, vm, ticks), b1 = run<1, 1>(grid, lm, vm, ticks); printf("B  s_store_dwordx2 x 4, s_waitcnt behind them        : %6.2f (%+.1f %%) | %6.2f (%+.1f %%)\n", vec / b0 * 1e-6, (a0 / b0 - 1) * 100, vec / b1 * 1e-6, (a1 / b1 - 1) * 100); }
	{ const float b0 = run<2, 0>(grid, lm, vm, ticks), b1 = run<2, 1>(grid, lm, vm, ticks); printf("C  s_store_dwordx2 x 4, no wait                      : %6.2f (%+.1f %%) | %6.2f (%+.1f %%)\n", vec / b0 * 1e-6, (a0 / b0 - 1) * 100, vec / b1 * 1e-6, (a1 / b1 - 1) * 100); }
	{ const float b0 = run<3, 0>(grid, lm, vm, ticks), b1 = run<3, 1>(grid, lm, vm, ticks); printf("D  s_store_dwordx4 x 2 (+ 4 s_mov), s_waitcnt behind  : %6.2f (%+.1f %%) | %6.2f (%+.1f %%)\n", vec / b0 * 1e-6, (a0 / b0 - 1) * 100, vec / b1 * 1e-6, (a1 / b1 - 1) * 100); }
	{ const float b0 = run<4, 0>(grid, lm, vm, ticks), b1 = run<4, 1>(grid, lm, vm, ticks); printf("E  s_store_dwordx4 x 2 (+ 4 s_mov), no wait           : %6.2f (%+.1f %%) | %6.2f (%+.1f %%)\n", vec / b0 * 1e-6, (a0 / b0 - 1) * 100, vec / b1 * 1e-6, (a1 / b1 - 1) * 100); }
	if(hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
	return 0;
}
