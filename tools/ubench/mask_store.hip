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
#define FILLER \
	"v_add_u32 %[x0], %[x0], %[a]\n\t v_max_i32 %[x1], %[x1], %[x0]\n\t v_add_u32 %[x2], %[x2], %[x1]\n\t v_sub_u32 %[x3], %[x3], %[x2]\n\t" \
	"v_max3_i32 %[x0], %[x0], %[x3], %[b]\n\t v_add_u32 %[x1], %[x1], %[b]\n\t v_perm_b32 %[x2], %[x2], %[x0], %[a]\n\t v_add_u32 %[x3], %[x3], %[x1]\n\t" \
	"v_mov_b32_dpp %[x0], %[x0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t v_mov_b32_dpp %[x1], %[x1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
	"v_add_u32 %[x2], %[x2], %[x3]\n\t v_max_i32 %[x3], %[x3], %[x0]\n\t v_add_u32 %[x0], %[x0], %[x2]\n\t v_sub_i32 %[x1], %[x1], %[x3] clamp\n\t" \
	"v_add_u32 %[x2], %[x2], %[a]\n\t v_add_u32 %[x3], %[x3], %[b]\n\t v_max_i32 %[x0], %[x0], %[x1]\n\t"
#define COMPARES \
	"v_cmp_lt_i32 %[A], %[x0], %[x1]\n\t v_cmp_lt_i32 %[B], %[x1], %[x2]\n\t v_cmp_lt_i32 %[C], %[x2], %[x3]\n\t v_cmp_lt_i32 %[D], %[x3], %[x0]\n\t" \
	"v_cmp_ge_i32 %[E], %[x0], %[x2]\n\t v_cmp_ge_i32 %[F], %[x1], %[x3]\n\t s_or_b64 %[A], %[A], %[E]\n\t s_or_b64 %[C], %[C], %[F]\n\t"

template<int VARIANT>
__global__ void __launch_bounds__(256, 2) k_fill(uint32_t *lane_major, uint64_t *vec_major, unsigned long long *ticks, int n_blocks)
{
	const int lane = threadIdx.x & 63; const uint32_t wave = blockIdx.x * 4 + threadIdx.x / 64;
	int x0 = lane * 7 + wave, x1 = lane * 13 + 5, x2 = lane ^ 0x55, x3 = wave * 3 + lane, a = lane + 1, b = 3 - lane;
	uint32_t mh = 0, mv = 0, me = 0, mf = 0;
	uint32_t *lm = lane_major + (uint64_t)wave * 4 * 64;
	const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
	uint64_t *vm = vec_major + (uint64_t)wave_u * 32 * 4;          /* one block's worth per wave (overwritten block after block, as a workspace is) */
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for(int blk = 0; blk < n_blocks; blk++) {
		mh = mv = me = mf = 0;
		uint64_t *vmk = vm;
		for(int k = 0; k < 32; k++) {
			uint64_t A, B, C, D, E, F;
			if(VARIANT == 0) {
				asm volatile(FILLER COMPARES
					"v_addc_co_u32 %[mh], vcc, %[mh], %[mh], %[A]\n\t v_addc_co_u32 %[mv], vcc, %[mv], %[mv], %[B]\n\t v_addc_co_u32 %[me], vcc, %[me], %[me], %[C]\n\t v_addc_co_u32 %[mf], vcc, %[mf], %[mf], %[D]\n\t"
					: [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [mh] "+v"(mh), [mv] "+v"(mv), [me] "+v"(me), [mf] "+v"(mf),
					  [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [E] "=&s"(E), [F] "=&s"(F)
					: [a] "v"(a), [b] "v"(b) : "vcc", "scc");
			} else {
				asm volatile("s_waitcnt lgkmcnt(0)\n\t" FILLER COMPARES
					"s_store_dwordx2 %[A], %[vm], 0x0\n\t s_store_dwordx2 %[B], %[vm], 0x8\n\t s_store_dwordx2 %[C], %[vm], 0x10\n\t s_store_dwordx2 %[D], %[vm], 0x18\n\t"
					: [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3),
					  [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [E] "=&s"(E), [F] "=&s"(F)
					: [a] "v"(a), [b] "v"(b), [vm] "s"(vmk) : "vcc", "scc", "memory");
				vmk += 4;
			}
		}
		if(VARIANT == 0) { lm[0 * 64 + lane] = mh; lm[1 * 64 + lane] = mv; lm[2 * 64 + lane] = me; lm[3 * 64 + lane] = mf; }
	}
	if(VARIANT == 1) { asm volatile("s_waitcnt lgkmcnt(0)\n\t s_dcache_wb\n\t s_waitcnt lgkmcnt(0)" ::: "memory"); }
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if(lane == 0) { atomicMax(&ticks[VARIANT], t1 - t0); }
	if(x0 + x1 + x2 + x3 == 0x7fffffff) { ticks[7] = 1; }
}

int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, grid = cus * 8, waves = grid * 4;          /* 8 waves per SIMD */
	uint32_t *lm; uint64_t *vm; unsigned long long *ticks;
	hipMalloc(&lm, (size_t)waves * 4 * 64 * 4); hipMalloc(&vm, (size_t)waves * 32 * 4 * 8); hipMalloc(&ticks, 64); hipMemset(ticks, 0, 64); hipMemset(vm, 0, (size_t)waves * 32 * 4 * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float ms[2];
	for(int rep = 0; rep < 2; rep++) {
		hipEventRecord(e0); hipLaunchKernelGGL(k_fill<0>, dim3(grid), dim3(256), 0, 0, lm, vm, ticks, BLOCKS_PER_WAVE); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
		hipEventRecord(e0); hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, 0, lm, vm, ticks, BLOCKS_PER_WAVE); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
	}
	if(hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
	const double vec = (double)waves * BLOCKS_PER_WAVE * 32;
	printf("%d CUs, %d waves, %d blocks of 32 vectors each\n", cus, waves, BLOCKS_PER_WAVE);
	printf("A  v_addc_co x 4 per vector + 1 KB per block : %.3f ms  %.2f G vectors/s\n", ms[0], vec / ms[0] * 1e-6);
	printf("B  s_store_dwordx2 x 4 per vector            : %.3f ms  %.2f G vectors/s   (%+.1f %%)\n", ms[1], vec / ms[1] * 1e-6, (ms[0] / ms[1] - 1) * 100);
	/* what B left in memory: the last block of every wave, vector-major; A's last block lane-major -- the same bits transposed */
	std::vector<uint32_t> hl((size_t)waves * 256); std::vector<uint64_t> hv((size_t)waves * 128);
	hipMemcpy(hl.data(), lm, hl.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hv.data(), vm, hv.size() * 8, hipMemcpyDeviceToHost);
	size_t bad = 0;
	for(int w = 0; w < waves; w++) for(int m = 0; m < 4; m++) for(int l = 0; l < 64; l++) {
		uint32_t col = 0; for(int k = 0; k < 32; k++) col = (col << 1) | (uint32_t)((hv[(size_t)w * 128 + k * 4 + m] >> l) & 1);
		if(col != hl[(size_t)w * 256 + m * 64 + l]) bad++;
	}
	printf("bit columns of A against the masks B stored: %zu of %zu differ\n", bad, (size_t)waves * 256);
	return bad != 0;
}
