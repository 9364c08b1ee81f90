/*
 * mm_index.hpp -- index construction on the device (mm_idx_gen, minialign.c:2951; workers :2767-2944; the per-bucket unstable sort ksort.h:84-131).
 *
 *   I1  mm_ref_sketch_kernel    wave per stretch of a reference sequence (2^18 positions, a warm-up of 64 in front): the (w,k)-minimizers with the emission
 *                               rule and position decoding K1 applies to reads (sketch_h / sketch_window_min are shared with it), counted, then written in
 *                               reference order
 *   I2  mm_idx_hist_kernel      wave per tile of that array: how many of its minimizers fall into each of the 2^b buckets (hash & mask)
 *       mm_idx_colscan_kernel   thread per bucket: the tiles' counts turned into write positions down the column (tile order inside a bucket = reference order)
 *       mm_idx_scatter_kernel   wave per tile, in order: (hash >> b, pos, rid) into its bucket -- the arrays the reference's ordered drain fills (:2854-2859)
 *   I3  mm_idx_sort_kernel      wave per bucket: radix_sort_128x on the remaining hash bits, replayed as a permutation of 4-byte entries (digit << 24 | index)
 *                               exactly as K2s does for seeds -- the cycle-leader walk of ksort.h:101-116 decides the order of the positions of one
 *                               minimizer, which is the order of its hits in every read's seed array (:2882, :2919-2927) -- then the records move once
 *   I4  mm_idx_runs_kernel      thread per element: run lengths of equal keys, a histogram of them (the occurrence thresholds are quantiles of it, :2981-2986)
 *       mm_idx_cut_kernel       per bucket the first key above the last threshold: the reference's fill cursor stops there and drops the rest of the bucket (Q2, :2927-2931)
 *       mm_idx_fill_kernel      thread per kept key: into the open-addressing table the mapper probes (compare-and-swap on the key word); a value list is the run itself
 *                               inside the sorted value array
 * All integer / byte work, bound by HBM latency (the sort's walks) and bandwidth (everything else); nothing here is a contraction.
 */
#pragma once
#include "mm_device.hpp"

namespace mm {

struct IdxMini { uint64_t hash; uint32_t pos, rid; };          /* a minimizer of the reference: full hash, k-mer start, sequence << 1 | strand; after the scatter: hash >> b */
struct RefStretch { uint64_t off; uint32_t len, begin, end, seq; uint64_t out; uint32_t host; uint32_t pad; };      /* sequence at `off` in the arena; positions [begin, end); first output slot; host = 1: circular sequence, sketched by the host */

struct I1Args { gaba::SeqArena ar; RefStretch *st; uint32_t n; uint32_t k, w; IdxMini *out; uint32_t *count; uint32_t emit; uint32_t *counter; };
__global__ void __launch_bounds__(256) mm_ref_sketch_kernel(I1Args a)
{
	const int lane = lane_id();
	const uint32_t k = a.k, w = a.w; const uint64_t kmask = (1ull << 2 * k) - 1;
	while(true) {
		uint32_t t = 0;
		if(lane == 0) { t = atomicAdd(a.counter, 1u); }
		t = (uint32_t)rdfirst((int)t);
		if(t >= a.n) { break; }
		const RefStretch &s = a.st[t];
		if(rdfirst((int)s.host)) { continue; }
		const uint64_t off = rdfirst64(s.off); const uint32_t len = (uint32_t)rdfirst((int)s.len), begin = (uint32_t)rdfirst((int)s.begin), end = (uint32_t)rdfirst((int)s.end), seq = (uint32_t)rdfirst((int)s.seq);
		IdxMini *out = a.out + rdfirst64(s.out);
		uint64_t h_prev = ~0ull, v_last = 0; uint32_t n_out = 0;
		/* one block of 64 positions in front of the stretch warms the window up: h is a function of the k + 1 bases that end at a position, the window holds w <= 31
		 * of them, u is the minimum of the position before -- all exact by the end of that block */
		for(uint32_t base = begin >= 64 ? begin - 64 : 0; base < end; base += 64) {
			const uint32_t p = base + (uint32_t)lane;
			const uint64_t h = p < end ? sketch_h(a.ar, off, p, len, k, w, kmask) : ~0ull;
			const uint64_t v = sketch_window_min(h, h_prev, w, lane);
			uint64_t vp = shfl_up64(v, 1);
			const uint64_t v63 = ((uint64_t)(uint32_t)rdlane((int)(v >> 32), 63) << 32) | (uint32_t)rdlane((int)v, 63);
			if(lane == 0) { vp = v_last; }
			if(p == k - 1) { vp = 0; }
			const bool emit = p >= k - 1 && p >= begin && p < end && ((v == h) || (v != vp));
			v_last = v63; h_prev = h;
			const uint64_t em = __ballot(emit);
			if(emit && a.emit) {
				const uint32_t iv = (uint32_t)(v & 0x7f), ip = (p - (k - 1)) % w;
				out[n_out + (uint32_t)__popcll(em & ((1ull << lane) - 1))] = IdxMini{ v >> 8, (p - (k - 1)) - ((ip + w - iv) % w), (seq << 1) | (uint32_t)((v >> 7) & 1) };
			}
			n_out += (uint32_t)__popcll(em);
		}
		if(lane == 0 && !a.emit) { a.count[t] = n_out; }
	}
}

/* ---- I2: stable partition into 2^b buckets ---- */
struct I2Args { const IdxMini *in; uint64_t n; uint32_t tile; uint32_t n_tiles; uint32_t bbits; uint32_t *hist; /* [tile][bucket] */ uint64_t *bofs; /* [bucket + 1] */ IdxMini *out; };
__global__ void __launch_bounds__(64) mm_idx_hist_kernel(I2Args a)
{
	const uint32_t t = blockIdx.x; const int lane = lane_id();
	const uint32_t nb = 1u << a.bbits, bmask = nb - 1;
	uint32_t *row = a.hist + (uint64_t)t * nb;
	for(uint32_t i = (uint32_t)lane; i < nb; i += 64) { row[i] = 0; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
	const uint64_t lo = (uint64_t)t * a.tile, hi = lo + a.tile < a.n ? lo + a.tile : a.n;
	for(uint64_t i = lo + (uint64_t)lane; i < hi; i += 64) { atomicAdd(&row[(uint32_t)a.in[i].hash & bmask], 1u); }
}
__global__ void __launch_bounds__(256) mm_idx_colscan_kernel(I2Args a)
{
	const uint32_t b = blockIdx.x * 256u + threadIdx.x, nb = 1u << a.bbits;
	if(b >= nb) { return; }
	uint32_t run = 0;
	for(uint32_t t = 0; t < a.n_tiles; t++) { uint32_t *p = a.hist + (uint64_t)t * nb + b; const uint32_t c = *p; *p = run; run += c; }
	a.bofs[b + 1] = run;          /* bucket sizes; the host turns them into offsets (2^b numbers) */
}
__global__ void __launch_bounds__(64) mm_idx_scatter_kernel(I2Args a)
{
	const uint32_t t = blockIdx.x; const int lane = lane_id();
	const uint32_t nb = 1u << a.bbits, bmask = nb - 1;
	uint32_t *row = a.hist + (uint64_t)t * nb;
	const uint64_t lo = (uint64_t)t * a.tile, hi = lo + a.tile < a.n ? lo + a.tile : a.n;
	for(uint64_t i0 = lo; i0 < hi; i0 += 64) {
		const uint64_t i = i0 + (uint64_t)lane; const bool act = i < hi;
		IdxMini m = act ? a.in[i] : IdxMini{ 0, 0, 0 };
		const uint32_t b = (uint32_t)m.hash & bmask;
		/* lanes with the same bucket: in lane (= reference) order behind each other */
		uint64_t todo = __ballot(act), mine = 0;
		while(todo) { const int l = __builtin_ctzll(todo); const uint32_t bl = (uint32_t)rdlane((int)b, l); const uint64_t g = __ballot(act && b == bl); if(act && b == bl) { mine = g; } todo &= ~g; }
		uint32_t base = 0;
		const int leader = mine ? __builtin_ctzll(mine) : 0;
		if(act && lane == leader) { base = atomicAdd(&row[b], (uint32_t)__popcll(mine)); }
		base = (uint32_t)__shfl((int)base, leader);
		if(act) { a.out[a.bofs[b] + base + (uint32_t)__popcll(mine & ((1ull << lane) - 1))] = IdxMini{ m.hash >> a.bbits, m.pos, m.rid }; }
	}
}

/* ---- I3: radix_sort_128x per bucket, replayed (see K2s in mm_device.hpp for the method; here the entries live in HBM: a bucket of a human-size reference has tens of
 * thousands of elements, and with the walks bound by latency either way it is the number of buckets in flight that counts -- a wave needs 7 KB of LDS for its tables) ---- */
constexpr uint32_t IS_STACK = 1024, IS_IDX_BITS = 24, IS_IDX_MASK = (1u << IS_IDX_BITS) - 1;
struct I3Args { const IdxMini *in; const uint64_t *bofs; uint32_t n_buckets; uint32_t key_bits; uint32_t *ent; uint64_t *hrem; uint64_t *val; uint32_t *counter; uint32_t *err; };
__device__ __forceinline__ void is_small_buckets(uint32_t *e, const LU32 *bs, const LU32 *be, uint32_t beg, uint32_t end, const IdxMini *g, int lane)
{
	uint32_t pos = beg;
	while(pos < end) {
		const uint32_t slot = pos + (uint32_t)lane; const bool valid = slot < end;
		const uint32_t x = valid ? e[slot] : 0u, d = x >> IS_IDX_BITS;
		const uint32_t b0 = valid ? bs[d] : 0u, b1 = valid ? be[d] : 0u;
		const uint64_t m_inc = __ballot(valid && b1 > pos + 64);
		const uint32_t cut = m_inc ? pos + (uint32_t)__builtin_ctzll(m_inc) : (pos + 64 < end ? pos + 64 : end);
		if(cut == pos) { pos = (uint32_t)rdfirst((int)b1); continue; }           /* a bucket of more than 64 (on the stack): step over it */
		const bool act = slot < cut && b1 - b0 >= 2;
		uint32_t rank = 0;
		if(__ballot(act)) {
			const uint64_t key = act ? g[x & IS_IDX_MASK].hash : 0ull;
			const uint32_t size = act ? b1 - b0 : 0u;
			for(uint32_t j = 0; __ballot(j < size); j++) {
				const int ol = (int)(b0 + j - pos) & 63;
				const uint64_t ok = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), ol) << 32) | (uint32_t)__shfl((int)key, ol);
				if(j < size && (ok < key || (ok == key && b0 + j < slot))) { rank++; }
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if(act) { e[b0 + rank] = x; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		pos = cut;
	}
}
__global__ void __launch_bounds__(64) mm_idx_sort_kernel(I3Args a)
{
	__shared__ uint32_t tab[768 + 2 * IS_STACK];
	LU32 *cnt = (LU32 *)tab, *bb = cnt + 256, *be = bb + 256, *stk = be + 256, *stsh = stk + IS_STACK;          /* stk: begin of a pending range, stsh: its end << 8 | shift / 8 */
	const int lane = lane_id();
	while(true) {
		uint32_t bi = 0;
		if(lane == 0) { bi = atomicAdd(a.counter, 1u); }
		bi = (uint32_t)rdfirst((int)bi);
		if(bi >= a.n_buckets) { break; }
		const uint64_t o0 = rdfirst64(a.bofs[bi]), o1 = rdfirst64(a.bofs[bi + 1]);
		if(o1 - o0 > IS_IDX_MASK) { if(lane == 0) { atomicOr(a.err, 1u); } continue; }
		const uint32_t n = (uint32_t)(o1 - o0);
		if(n == 0) { continue; }
		const IdxMini *g = a.in + o0; uint32_t *e = a.ent + o0;
		for(uint32_t i = (uint32_t)lane; i < n; i += 64) { e[i] = i; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		uint32_t sp = 0, err = 0;
		if(n <= 64) {
			if(lane == 0) { bb[0] = 0; be[0] = n; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			is_small_buckets(e, bb, be, 0, n, g, lane);
		} else {
			if(lane == 0) { stk[0] = 0u; stsh[0] = (n << 8) | ((a.key_bits - 8) >> 3); }
			sp = 1;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		while(sp > 0) {
			sp--;
			const uint32_t beg = (uint32_t)rdfirst((int)stk[sp]), es = (uint32_t)rdfirst((int)stsh[sp]);
			const uint32_t end = es >> 8, m = end - beg; const int sh = (int)(es & 255u) * 8;
			for(int k = lane; k < 256; k += 64) { cnt[k] = 0; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			for(uint32_t i = beg + (uint32_t)lane; i < end; i += 64) {
				const uint32_t src = e[i] & IS_IDX_MASK;
				const uint32_t d = (uint32_t)(g[src].hash >> sh) & 255u;
				e[i] = d << IS_IDX_BITS | src;
				atomicAdd((uint32_t *)&cnt[d], 1u);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			const uint32_t d0 = (uint32_t)rdfirst((int)e[beg]) >> IS_IDX_BITS;
			if((uint32_t)rdfirst((int)cnt[d0]) == m) {
				/* every element has the same digit: the level leaves the range as it is (the upper digits of hash >> b are all zero) */
				if(sh) { if(lane == 0) { stk[sp] = beg; stsh[sp] = (end << 8) | (uint32_t)((sh > 8 ? sh - 8 : 0) >> 3); } sp++; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				continue;
			}
			const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
			{
				uint32_t incl = c0 + c1 + c2 + c3;
				for(int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, dd); if(lane >= dd) { incl += o; } }
				uint32_t acc = beg + incl - (c0 + c1 + c2 + c3);
				bb[4 * lane] = acc; acc += c0; be[4 * lane] = acc; bb[4 * lane + 1] = acc; acc += c1; be[4 * lane + 1] = acc;
				bb[4 * lane + 2] = acc; acc += c2; be[4 * lane + 2] = acc; bb[4 * lane + 3] = acc; acc += c3; be[4 * lane + 3] = acc;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			/* the in-place cycle-leader permutation (ksort.h:101-116), literally, on the 4-byte entries: buckets in ascending order; inside a bucket the cursor walks to its
			 * end, and every element that is not at home starts a cycle -- it goes to the cursor of its own bucket, the element it displaces to the cursor of *its* bucket
			 * (at home there or not) -- until an element of the current bucket turns up.  Hash digits are uniform: nearly every element is displaced, one lane walks. */
			if(lane == 0) {
				for(uint32_t k = 0; k < 256; k++) {
					uint32_t b = bb[k]; const uint32_t ee = be[k];
					while(b != ee) {
						const uint32_t x = e[b];
						if((x >> IS_IDX_BITS) == k) { b++; continue; }
						uint32_t tmp = x, l_ = x >> IS_IDX_BITS;
						do { const uint32_t p = bb[l_]; const uint32_t y = e[p]; e[p] = tmp; bb[l_] = p + 1; tmp = y; l_ = y >> IS_IDX_BITS; } while(l_ != k);
						e[b] = tmp; b++;
					}
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			for(int k = lane; k < 256; k += 64) { bb[k] = k == 0 ? beg : be[k - 1]; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			if(sh) {
				const int ns = sh > 8 ? sh - 8 : 0;
				for(int k0 = 0; k0 < 256; k0 += 64) {
					const int k = k0 + lane; const uint32_t nb = be[k] - bb[k];
					const uint64_t mm_ = __ballot(nb > 64);
					const uint32_t slot = sp + (uint32_t)__popcll(mm_ & ((1ull << lane) - 1));
					if(nb > 64) { if(slot < IS_STACK) { stk[slot] = bb[k]; stsh[slot] = (be[k] << 8) | (uint32_t)(ns >> 3); } else { err = 1; } }
					sp += (uint32_t)__popcll(mm_);
				}
				if(sp > IS_STACK) { sp = IS_STACK; }
				is_small_buckets(e, bb, be, beg, end, g, lane);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		if(__ballot(err != 0) && lane == 0) { atomicOr(a.err, 2u); }
		/* the records move once: remaining hash bits and (pos | rid << 32) in sorted order */
		for(uint32_t i = (uint32_t)lane; i < n; i += 64) { const IdxMini v = g[e[i] & IS_IDX_MASK]; a.hrem[o0 + i] = v.hash; a.val[o0 + i] = (uint64_t)v.pos | ((uint64_t)v.rid << 32); }
	}
}

/* ---- I4: keys, occurrence counts, table ---- */
constexpr uint32_t IDX_HB = 1u << 16;          /* counts below this go into the histogram, the few above into a list */
struct I4Args {
	const uint64_t *hrem; const uint64_t *val; uint64_t n; const uint64_t *bofs; uint32_t n_buckets; uint32_t bbits;
	uint32_t *runlen;                          /* per element: length of the run of equal keys it starts, 0 inside a run */
	unsigned long long *hist;                  /* [IDX_HB + 1] */
	uint32_t *big; uint32_t big_cap; uint32_t *n_big;
	uint64_t *cut;                             /* per bucket: first element of the first key with more than max_cnt occurrences (bofs[b + 1] when none) */
	uint32_t max_cnt;
	unsigned long long *n_keys;
	IdxSlot *slot; uint64_t mask;
};
__device__ __forceinline__ uint32_t bucket_of(const uint64_t *bofs, uint32_t n_buckets, uint64_t i)
{
	uint32_t lo = 0, hi = n_buckets;          /* bofs[lo] <= i < bofs[hi] */
	while(hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if(bofs[mid] <= i) { lo = mid; } else { hi = mid; } }
	return lo;
}
__global__ void __launch_bounds__(256) mm_idx_runs_kernel(I4Args a)
{
	/* nearly every key occurs once or a few times: the counts below 256 are added up in LDS first (one global atomic per block and count instead of one per key) */
	__shared__ uint32_t lh[256];
	lh[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if(i < a.n) {
		const uint32_t b = bucket_of(a.bofs, a.n_buckets, i);
		const uint64_t lo = a.bofs[b], hi = a.bofs[b + 1]; const uint64_t key = a.hrem[i];
		if(i > lo && a.hrem[i - 1] == key) { a.runlen[i] = 0; }
		else {
			uint64_t e = i + 1; while(e < hi && a.hrem[e] == key) { e++; }
			const uint32_t len = (uint32_t)(e - i);
			a.runlen[i] = len;
			if(len < 256) { atomicAdd(&lh[len], 1u); }
			else if(len < IDX_HB) { atomicAdd(&a.hist[len], 1ull); }
			else { atomicAdd(&a.hist[IDX_HB], 1ull); const uint32_t q = atomicAdd(a.n_big, 1u); if(q < a.big_cap) { a.big[q] = len; } }
		}
	}
	__syncthreads();
	if(lh[threadIdx.x]) { atomicAdd(&a.hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]); }
}
__global__ void __launch_bounds__(256) mm_idx_cut_kernel(I4Args a)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if(i >= a.n) { return; }
	if(a.runlen[i] > a.max_cnt) { atomicMin((unsigned long long *)&a.cut[bucket_of(a.bofs, a.n_buckets, i)], (unsigned long long)i); }
}
__global__ void __launch_bounds__(256) mm_idx_fill_kernel(I4Args a)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	const uint32_t len = i < a.n ? a.runlen[i] : 0;
	bool key = false;
	if(len) {
		const uint32_t b = bucket_of(a.bofs, a.n_buckets, i);
		key = i < a.cut[b];
		if(key && a.slot) {
			const uint64_t minier = (a.hrem[i] << a.bbits) | b;
			const uint64_t value = len == 1 ? a.val[i] : ((1ull << 63) | (i << 24) | (uint64_t)len);
			uint64_t s = idx_hash(minier) & a.mask;
			while(true) {
				const unsigned long long old = atomicCAS((unsigned long long *)&a.slot[s].key, 0ull, (unsigned long long)(minier + 1));
				if(old == 0ull) { a.slot[s].val = value; break; }
				s = (s + 1) & a.mask;
			}
		}
	}
	/* (first pass, slot == NULL: the number of keys, which sizes the table) */
	if(!a.slot) { const uint64_t m = __ballot(key); if(m && lane_id() == __builtin_ctzll(m)) { atomicAdd(a.n_keys, (unsigned long long)__popcll(m)); } }
}

} /* namespace mm */
