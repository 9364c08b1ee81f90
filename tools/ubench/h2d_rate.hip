/* h2d_rate.hip -- how fast does text in ordinary (pageable) host memory get into HBM?  The ways the reader could take, timed on one buffer:
 *   pageable   hipMemcpy straight from the buffer (the runtime stages it itself)
 *   pinned     hipMemcpyAsync from a hipHostMalloc'd copy (the ceiling of the link)
 *   register   hipHostRegister of the whole buffer (timed), then hipMemcpyAsync from it, then hipHostUnregister (timed)
 *   staged     T host threads copy 32 MB pieces into a ring of R pinned buffers, hipMemcpyAsync from there (what the library's reader does)
 * usage: h2d_rate [GB = 4]     build: hipcc --offload-arch=gfx950 -O2 -o h2d_rate h2d_rate.hip -lpthread */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>

static double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
#define OK(e) do { hipError_t r_ = (e); if(r_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while(0)

struct CopyPool {
	std::vector<std::thread> th; std::mutex mu; std::condition_variable cv, dcv;
	const char *src = nullptr; char *dst = nullptr; size_t n = 0; unsigned long gen = 0; unsigned left = 0, nth = 0; bool stop = false;
	void start(unsigned want) { nth = want; for(unsigned t = 0; t < nth; t++) th.emplace_back([this, t]() { run(t); }); }
	void run(unsigned t)
	{
		unsigned long seen = 0;
		while(true) {
			std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || gen != seen; }); if(stop) return;
			seen = gen; const char *sp = src; char *dp = dst; const size_t bytes = n; lk.unlock();
			const size_t lo = (bytes * t / nth) & ~(size_t)63, hi = t + 1 == nth ? bytes : ((bytes * (t + 1) / nth) & ~(size_t)63);
			if(hi > lo) memcpy(dp + lo, sp + lo, hi - lo);
			lk.lock(); if(--left == 0) dcv.notify_all();
		}
	}
	void copy(char *d, const char *sp, size_t bytes)
	{
		if(nth == 0) { memcpy(d, sp, bytes); return; }
		std::unique_lock<std::mutex> lk(mu); src = sp; dst = d; n = bytes; left = nth; gen++; cv.notify_all();
		dcv.wait(lk, [&]() { return left == 0; });
	}
	~CopyPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for(auto &t : th) t.join(); }
};

int main(int argc, char **argv)
{
	const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 4, n = gb << 30;
	char *host = (char *)malloc(n); if(!host) return 1;
	{ std::vector<std::thread> th; for(int t = 0; t < 16; t++) th.emplace_back([&, t]() { for(size_t i = n / 16 * t; i < n / 16 * (t + 1); i += 4096) host[i] = (char)(i >> 12); }); for(auto &x : th) x.join(); }
	char *dev = nullptr; OK(hipMalloc(&dev, n)); OK(hipMemset(dev, 0, n)); OK(hipDeviceSynchronize());
	hipStream_t st; OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	double t0;
	for(int rep = 0; rep < 2; rep++) { t0 = now_ms(); OK(hipMemcpy(dev, host, n, hipMemcpyHostToDevice)); printf("pageable hipMemcpy          : %6.1f GB/s\n", n * 1e-6 / (now_ms() - t0)); }
	{
		char *pin = nullptr; const size_t pn = n < (2ull << 30) ? n : (2ull << 30); OK(hipHostMalloc(&pin, pn, hipHostMallocDefault)); memcpy(pin, host, pn);
		for(int rep = 0; rep < 2; rep++) { t0 = now_ms(); OK(hipMemcpyAsync(dev, pin, pn, hipMemcpyHostToDevice, st)); OK(hipStreamSynchronize(st)); printf("pinned source (ceiling)     : %6.1f GB/s\n", pn * 1e-6 / (now_ms() - t0)); }
		OK(hipHostFree(pin));
	}
	for(int rep = 0; rep < 2; rep++) {
		t0 = now_ms(); hipError_t r = hipHostRegister(host, n, hipHostRegisterDefault); const double t_reg = now_ms() - t0;
		if(r != hipSuccess) { printf("hipHostRegister failed: %s\n", hipGetErrorString(r)); break; }
		t0 = now_ms(); OK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st)); OK(hipStreamSynchronize(st)); const double t_cp = now_ms() - t0;
		t0 = now_ms(); OK(hipHostUnregister(host)); const double t_un = now_ms() - t0;
		printf("register %.0f ms (%5.1f GB/s) + copy %6.1f GB/s + unregister %.0f ms: %6.1f GB/s all told\n", t_reg, n * 1e-6 / t_reg, n * 1e-6 / t_cp, t_un, n * 1e-6 / (t_reg + t_cp + t_un));
	}
	/* registering piece by piece just ahead of the copy (what a reader over a mapped file could do) */
	for(size_t piece : { (size_t)64 << 20, (size_t)256 << 20 }) {
		t0 = now_ms();
		for(size_t o = 0; o < n; o += piece) { const size_t nb = n - o < piece ? n - o : piece; OK(hipHostRegister(host + o, nb, hipHostRegisterDefault)); OK(hipMemcpyAsync(dev + o, host + o, nb, hipMemcpyHostToDevice, st)); OK(hipStreamSynchronize(st)); OK(hipHostUnregister(host + o)); }
		printf("register / copy / unregister in pieces of %3zu MB: %6.1f GB/s\n", piece >> 20, n * 1e-6 / (now_ms() - t0));
	}
	for(size_t piece : { (size_t)8 << 20, (size_t)32 << 20, (size_t)64 << 20 }) for(unsigned T : { 1u, 2u, 4u, 8u, 12u, 16u, 24u }) for(int R : { 2, 4 }) {
		if(R == 2 && T != 8) continue;
		CopyPool cp; cp.start(T);
		std::vector<char *> pin(R); std::vector<hipEvent_t> ev(R);
		for(int i = 0; i < R; i++) { OK(hipHostMalloc(&pin[i], piece, hipHostMallocDefault)); memset(pin[i], 0, piece); OK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
		double best = 0;
		for(int rep = 0; rep < 2; rep++) {
			t0 = now_ms(); size_t k = 0;
			for(size_t o = 0; o < n; o += piece, k++) {
				const size_t nb = n - o < piece ? n - o : piece; const int pi = (int)(k % R);
				OK(hipEventSynchronize(ev[pi])); cp.copy(pin[pi], host + o, nb);
				OK(hipMemcpyAsync(dev + o, pin[pi], nb, hipMemcpyHostToDevice, st)); OK(hipEventRecord(ev[pi], st));
			}
			OK(hipStreamSynchronize(st));
			const double r = n * 1e-6 / (now_ms() - t0); if(r > best) best = r;
		}
		printf("staged: pieces of %2zu MB, ring of %d, %2u copy threads: %6.1f GB/s\n", piece >> 20, R, T, best);
		for(int i = 0; i < R; i++) { OK(hipHostFree(pin[i])); OK(hipEventDestroy(ev[i])); }
	}
	return 0;
}
