// tools/ubench/traffic_cal.hip -- what do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) read on kernels whose HBM traffic is known?  (the
// extension kernel's traffic_over_algorithmic was "somewhere in 1.06 - 1.45" because MI355X_MICROARCH.md calibrates the doubling of FETCH_SIZE on wide coalesced
// streaming reads only, and the extension kernel's reads are 32-byte pieces and scratch reloads.)  Five kernels, each moving a known number of bytes over a buffer far
// larger than the 256 MB of last-level cache, so that what the counters see is memory traffic:
//   stream_read      every lane reads 16 B, coalesced (1 KB per wave and instruction)                                   known: bytes read
//   strided32_read   every lane reads 32 B at a stride of 1 296 B (the block trailer + mask pair shape of the DP blocks)  known: useful bytes; 64-B and 128-B granule bytes
//   stream_write     every lane writes 16 B, coalesced                                                                  known: bytes written
//   block_write      a wave writes 1 296-B blocks one after the other (1 KB coalesced + 256 B + 16 B from lane 0)        known: bytes written
//   scratch_toy      every lane fills 3 KB of a private array and reads it back through a data-dependent index          known: scratch bytes written / read per lane
// Run under tools/pmc_calibrate.sh, which takes the counters per kernel and prints counter / known.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while(0)

__global__ void __launch_bounds__(256) stream_read(const uint4 *src, uint64_t n16, unsigned long long *sink)
{
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; const uint64_t step = (uint64_t)gridDim.x * 256;
	uint32_t acc = 0;
	for(; i < n16; i += step) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if(acc == 0x12345678u) { atomicAdd(sink, 1ull); }
}
__global__ void __launch_bounds__(256) strided32_read(const uint8_t *src, uint64_t n_items, uint64_t stride, unsigned long long *sink)
{
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; const uint64_t step = (uint64_t)gridDim.x * 256;
	uint32_t acc = 0;
	for(; i < n_items; i += step) { const uint4 *p = (const uint4 *)(src + i * stride); const uint4 a = p[0], b = p[1]; acc ^= a.x ^ a.w ^ b.y ^ b.z; }
	if(acc == 0x12345678u) { atomicAdd(sink, 1ull); }
}
__global__ void __launch_bounds__(256) stream_write(uint4 *dst, uint64_t n16)
{
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; const uint64_t step = (uint64_t)gridDim.x * 256;
	for(; i < n16; i += step) { dst[i] = make_uint4((uint32_t)i, 1, 2, 3); }
}
__global__ void __launch_bounds__(256) block_write(uint8_t *dst, uint64_t n_blocks)
{
	const uint32_t lane = threadIdx.x & 63; uint64_t w = (uint64_t)blockIdx.x * 4 + threadIdx.x / 64; const uint64_t step = (uint64_t)gridDim.x * 4;
	for(; w < n_blocks; w += step) {
		uint8_t *b = dst + w * 1296;
		for(int m = 0; m < 4; m++) { ((uint32_t *)b)[m * 64 + lane] = lane + m; }
		((uint32_t *)(b + 1024))[lane] = lane * 3;
		if(lane == 0) { *(uint4 *)(b + 1280) = make_uint4(1, 2, 3, 4); }
	}
}
__global__ void __launch_bounds__(256) scratch_toy(uint32_t *out, int rounds)
{
	volatile uint32_t priv[768];          /* 3 KB per lane: more than the register file gives a lane */
	const uint32_t t = blockIdx.x * 256 + threadIdx.x; uint32_t acc = t;
	for(int r = 0; r < rounds; r++) {
		for(int i = 0; i < 768; i++) { priv[i] = acc + (uint32_t)i; }
		for(int i = 0; i < 768; i++) { acc += priv[(i * 7 + (acc & 3)) % 768]; }
	}
	out[t] = acc;
}

int main(int argc, char **argv)
{
	const uint64_t gb = argc > 1 ? (uint64_t)atoll(argv[1]) : 8;          /* buffer size */
	const uint64_t bytes = gb << 30;
	uint8_t *buf; unsigned long long *sink; uint32_t *out;
	CHECK(hipMalloc(&buf, bytes + 4096)); CHECK(hipMalloc(&sink, 8)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMemset(sink, 0, 8));
	const int grid = 256 * 16;
	CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
	hipLaunchKernelGGL(stream_read, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
	const uint64_t stride = 1296, n_items = bytes / stride;
	hipLaunchKernelGGL(strided32_read, dim3(grid), dim3(256), 0, 0, (const uint8_t *)buf, n_items, stride, sink);
	hipLaunchKernelGGL(stream_write, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
	hipLaunchKernelGGL(block_write, dim3(grid), dim3(256), 0, 0, buf, bytes / 1296);
	const int rounds = 64;
	hipLaunchKernelGGL(scratch_toy, dim3(grid), dim3(256), 0, 0, out, rounds);
	CHECK(hipDeviceSynchronize());
	printf("{\"buffer_bytes\": %llu, \"stream_read\": {\"read\": %llu}, \"strided32_read\": {\"useful\": %llu, \"granule64\": %llu, \"granule128\": %llu, \"items\": %llu},"
		" \"stream_write\": {\"written\": %llu}, \"block_write\": {\"written\": %llu}, \"scratch_toy\": {\"lanes\": %llu, \"written\": %llu, \"read\": %llu}}\n",
		(unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)(n_items * 32), (unsigned long long)(n_items * 64), (unsigned long long)(n_items * 128), (unsigned long long)n_items,
		(unsigned long long)bytes, (unsigned long long)(bytes / 1296 * 1296), (unsigned long long)grid * 256, (unsigned long long)grid * 256 * 768 * 4 * rounds, (unsigned long long)grid * 256 * 768 * 4 * rounds);
	return 0;
}
