/* K2: the unstable radix sort of the seeds replayed (ksort.h:84-131) and the chain sweep (mm_chain_seeds, mm_circularize, mm_chain) -- part of mm_device.hpp (included from there, inside namespace mm; split out in round 6 so that each stage can be read on its own) */
/* =====================================================================================================
 * K2: seed sort + chaining, one lane per read (serial by nature; 64 reads per wavefront)
 * ===================================================================================================== */
/* mm_circularize (minialign.c:3632-3696), after the chains of a read are known and before they are sorted: a chain whose root seed lies within the
 * window of the end of a circular reference is linked to a leaf seed just behind the origin -- the far chain is switched off (top bit of plen), its
 * length and root seed pass to the near one.  Serial, one lane; only reads that hit a circular reference get here.  Leaf view of a Seed:
 * upos = rsid, rid, vpos = lsid, lid = cid. */
__device__ inline void circularize(Seed *s, Root *c, uint32_t n_seed, uint32_t tlid, uint32_t n_root, const uint32_t *seq_len, const uint8_t *seq_circ, uint32_t twlen)
{
	uint32_t blid = n_seed + 1;
	for(uint32_t rcid = 0; rcid < n_root; rcid++) {
		const uint32_t rlid = c[rcid].lid, rsid = s[rlid].upos, rid = s[rlid].rid;
		if(seq_circ[rid] == 0 || (uint32_t)(seq_len[rid] - (uint32_t)AS(s[rsid])) > twlen) { continue; }
		const uint32_t rlen = seq_len[rid];
		const int32_t uofs = (int32_t)(rlen << 1), vofs = -(int32_t)rlen;              /* _ud(rlen, 0), _vd(rlen, 0) */
		while(blid < tlid && s[s[blid].vpos].rid < rid) { blid++; }
		const uint32_t vub = s[rsid].vpos - (uint32_t)vofs + twlen;
		while(blid < tlid && s[s[blid].vpos].vpos > vub) { blid++; }
		/* window of the root seed moved by one turn: (u <= uub, rid <= rid, v <= vub, v > vlb), signed */
		const int32_t w_u = (int32_t)(s[rsid].upos + twlen - (uint32_t)uofs), w_r = (int32_t)s[rsid].rid;
		const int32_t w_vub = (int32_t)(s[rsid].vpos + twlen - (uint32_t)vofs), w_vlb = (int32_t)(s[rsid].vpos - (uint32_t)vofs);
		uint64_t best = ~0ull;
		for(uint32_t lid = blid; lid < tlid; lid++) {
			const Seed &f = s[s[lid].vpos];
			if(!((int32_t)f.upos <= w_u && (int32_t)f.rid <= w_r && (int32_t)f.vpos <= w_vub && (int32_t)f.vpos > w_vlb)) { continue; }
			const uint32_t cid = s[lid].lid;
			if(cid == 0xffffffffu || (c[cid].plen & 0x80000000u)) { continue; }
			const uint64_t cand = ((uint64_t)c[cid].plen << 32) | lid;
			best = cand < best ? cand : best;
		}
		if(best == ~0ull) { continue; }
		const uint32_t pd = (uint32_t)(best >> 32), llid = (uint32_t)best, lcid = s[llid].lid;
		c[lcid].lid = rlid; c[lcid].plen |= 0x80000000u;
		s[s[llid].vpos].lid = ~s[rlid].upos;
		c[rcid].plen -= (uint32_t)OFS((int32_t)pd);
		s[rlid].upos = s[llid].upos;
	}
}

/* ---- ksort.h:84-131 restated over 16-byte records with a 64-bit key (first 8 bytes) ---- */
struct U128 { uint64_t k, v; };
__device__ __forceinline__ void ins_sort_128(U128 *beg, U128 *end)
{
	for(U128 *i = beg + 1; i < end; ++i) {
		if(i->k < (i - 1)->k) {
			U128 *j, tmp = *i;
			for(j = i; j > beg && tmp.k < (j - 1)->k; --j) { *j = *(j - 1); }
			*j = tmp;
		}
	}
}
/*
 * radix_sort_128x: MSD, 8 bits per level starting at bit 56, in-place cycle-leader permutation (UNSTABLE) with
 * insertion sort for buckets of <= 64.  The recursion order of sibling buckets is irrelevant (disjoint ranges), so an
 * explicit stack of pending ranges replaces it; one 256-entry bucket table lives in the per-lane scratch.
 * Returns false if the scratch stack overflowed.
 */
__device__ inline bool radix_sort_128(U128 *p, uint32_t l, uint32_t *scratch, uint32_t scratch_words)
{
	if(l <= 64) { ins_sort_128(p, p + l); return true; }
	uint32_t *bb = scratch, *be = scratch + 256;             /* bucket begin / end (element indices relative to p) */
	uint32_t *stack = scratch + 512; uint32_t cap = (scratch_words - 512) / 3, sp = 0;
	stack[0] = 0; stack[1] = l; stack[2] = 56; sp = 1;
	while(sp > 0) {
		sp--;
		uint32_t beg = stack[3 * sp], end = stack[3 * sp + 1]; int s = (int)stack[3 * sp + 2];
		for(int k = 0; k < 256; k++) { bb[k] = be[k] = beg; }
		for(uint32_t i = beg; i != end; ++i) { ++be[(p[i].k >> s) & 255]; }
		for(int k = 1; k < 256; k++) { be[k] += be[k - 1] - beg; bb[k] = be[k - 1]; }
		for(int k = 0; k < 256;) {
			if(bb[k] != be[k]) {
				int l_ = (int)((p[bb[k]].k >> s) & 255);
				if(l_ != k) {
					U128 tmp = p[bb[k]], swap;
					do { swap = tmp; tmp = p[bb[l_]]; p[bb[l_]++] = swap; l_ = (int)((tmp.k >> s) & 255); } while(l_ != k);
					p[bb[k]++] = tmp;
				} else { ++bb[k]; }
			} else { ++k; }
		}
		bb[0] = beg; for(int k = 1; k < 256; k++) { bb[k] = be[k - 1]; }
		if(s) {
			int ns = s > 8 ? s - 8 : 0;
			for(int k = 0; k < 256; k++) {
				uint32_t n = be[k] - bb[k];
				if(n > 64) { if(sp >= cap) { return false; } stack[3 * sp] = bb[k]; stack[3 * sp + 1] = be[k]; stack[3 * sp + 2] = (uint32_t)ns; sp++; }
				else if(n > 1) { ins_sort_128(p + bb[k], p + be[k]); }
			}
		}
	}
	return true;
}
/* radix_sort_64x (key = low 32 bits of an 8-byte record): same algorithm, 4 key bytes */
struct U64R { uint32_t k, v; };
__device__ __forceinline__ void ins_sort_64(U64R *beg, U64R *end)
{
	for(U64R *i = beg + 1; i < end; ++i) {
		if(i->k < (i - 1)->k) {
			U64R *j, tmp = *i;
			for(j = i; j > beg && tmp.k < (j - 1)->k; --j) { *j = *(j - 1); }
			*j = tmp;
		}
	}
}
__device__ inline bool radix_sort_64(U64R *p, uint32_t l, uint32_t *scratch, uint32_t scratch_words)
{
	if(l <= 64) { ins_sort_64(p, p + l); return true; }
	uint32_t *bb = scratch, *be = scratch + 256;
	uint32_t *stack = scratch + 512; uint32_t cap = (scratch_words - 512) / 3, sp = 0;
	stack[0] = 0; stack[1] = l; stack[2] = 24; sp = 1;
	while(sp > 0) {
		sp--;
		uint32_t beg = stack[3 * sp], end = stack[3 * sp + 1]; int s = (int)stack[3 * sp + 2];
		for(int k = 0; k < 256; k++) { bb[k] = be[k] = beg; }
		for(uint32_t i = beg; i != end; ++i) { ++be[(p[i].k >> s) & 255]; }
		for(int k = 1; k < 256; k++) { be[k] += be[k - 1] - beg; bb[k] = be[k - 1]; }
		for(int k = 0; k < 256;) {
			if(bb[k] != be[k]) {
				int l_ = (int)((p[bb[k]].k >> s) & 255);
				if(l_ != k) {
					U64R tmp = p[bb[k]], swap;
					do { swap = tmp; tmp = p[bb[l_]]; p[bb[l_]++] = swap; l_ = (int)((tmp.k >> s) & 255); } while(l_ != k);
					p[bb[k]++] = tmp;
				} else { ++bb[k]; }
			} else { ++k; }
		}
		bb[0] = beg; for(int k = 1; k < 256; k++) { bb[k] = be[k - 1]; }
		if(s) {
			int ns = s > 8 ? s - 8 : 0;
			for(int k = 0; k < 256; k++) {
				uint32_t n = be[k] - bb[k];
				if(n > 64) { if(sp >= cap) { return false; } stack[3 * sp] = bb[k]; stack[3 * sp + 1] = be[k]; stack[3 * sp + 2] = (uint32_t)ns; sp++; }
				else if(n > 1) { ins_sort_64(p + bb[k], p + be[k]); }
			}
		}
	}
	return true;
}

/* window vectors of the reference's v4i32 code (minialign.c:3366-3402): e0 = upos, e1 = rid, e2 = e3 = vpos */
struct V4 { int32_t e0, e1, e2, e3; };
__device__ __forceinline__ V4 load_pv(const Seed &s) { return V4{ (int32_t)s.upos, (int32_t)s.rid, (int32_t)s.vpos, (int32_t)s.vpos }; }
__device__ __forceinline__ V4 add_win(V4 a, int32_t len) { return V4{ (int32_t)((uint32_t)a.e0 + (uint32_t)len), a.e1, (int32_t)((uint32_t)a.e2 + (uint32_t)len), a.e3 }; }
/* _inside_wv: (v > vlb, v <= vub, rid <= rid, u <= uub) <=> gt-mask == 0xf000 */
__device__ __forceinline__ bool inside_wv(const V4 &u, const V4 &d) { return !(d.e0 > u.e0) && !(d.e1 > u.e1) && !(d.e2 > u.e2) && (d.e3 > u.e3); }
__device__ __forceinline__ bool inside_uub(const V4 &u, const V4 &d) { return !(d.e0 > u.e0) && !(d.e1 > u.e1); }
__device__ __forceinline__ V4 update_wv(V4 w, const V4 &f)
{
	uint32_t d0 = (uint32_t)w.e0 - (uint32_t)f.e0, d2 = (uint32_t)w.e2 - (uint32_t)f.e2;
	w.e0 = (int32_t)((uint32_t)w.e0 - d2); w.e2 = (int32_t)((uint32_t)w.e2 - d0);
	return w;
}
__device__ __forceinline__ int32_t pdiff(const V4 &w, const V4 &f) { return (int32_t)(((uint32_t)w.e0 - (uint32_t)f.e0) + ((uint32_t)w.e2 - (uint32_t)f.e2)); }
/* double -> uint32 as the reference's x86-64 build does it (cvttsd2si r64 + truncation) */
__device__ __forceinline__ uint32_t d2u32(double d) { if(!(d > -9.2e18 && d < 9.2e18)) { return 0; } return (uint32_t)(long long)d; }
__device__ __forceinline__ uint32_t f2u32(float f) { if(!(f > -9.2e18f && f < 9.2e18f)) { return 0; } return (uint32_t)(long long)f; }


typedef __attribute__((address_space(3))) Seed LSeed;
typedef __attribute__((address_space(3))) uint32_t LU32;
/* -----------------------------------------------------------------------------------------------------
 * K2s: radix_sort_128x (ksort.h:84-131) of the seed array, one wavefront per read, as a permutation of small indices.
 *
 * The reference's sort is an MSD radix sort with an in-place cycle-leader permutation per level (unstable: its exact element order decides ties) and a
 * stable insertion sort for buckets of <= 64.  Which element ends where in one level depends on the digits alone, so the level is replayed on 4-byte
 * entries (digit << 16 | index of the seed) in LDS instead of on the 16-byte seeds: 4 B of LDS per seed, which lets a CU hold a dozen reads instead of
 * three -- the replay is a chain of dependent LDS round trips, and the only way to make it cheap is to have many of them in flight.  The 64-bit keys stay
 * where K1 wrote them (HBM / L2) and are fetched once per level, 64 at a time; the stable sort of the small buckets is a rank computation (lane = element,
 * shuffles over its bucket; a stable sort has one answer, so any stable method gives the insertion sort's); the seeds themselves move once, at the end.
 * ----------------------------------------------------------------------------------------------------- */
constexpr uint32_t K2S_MAX_N = 24576;                  /* seeds + sentinel a read may have here (104 KB of LDS: what a CU has left beside eight extension workgroups, see K2C_MAX_LDS_KB); larger reads: in-HBM path of K2a */
constexpr uint32_t K2S_STACK = 512;                    /* pending ranges (each > 64 elements, disjoint) */
constexpr uint32_t K2S_TABLE_WORDS = 768 + 2 * K2S_STACK;
constexpr uint64_t K2S_SENTINEL_KEY = 0x7fffffff80000000ull;      /* { upos = INT32_MIN, rid = INT32_MAX }, minialign.c:3531 */
__host__ __device__ inline uint32_t k2s_bytes(uint32_t n_all) { return 4u * ((n_all + 63u) & ~63u) + 4u * K2S_TABLE_WORDS; }
struct K2sArgs {
	ReadState *st; const uint32_t *work; uint32_t n_work;
	Seed *seed_pool;
	uint32_t lds_bytes, n_lo, n_hi;   /* this launch takes the reads with n_lo < k2s_bytes(n + 1) <= n_hi */
	uint32_t *counter;
	unsigned long long *prof;         /* [0] wave cycles */
	uint32_t start_shift;             /* the first radix level at which two seeds of a read can differ (56: none skipped): with fewer than 2^8 (2^16) reference sequences the levels of bits 56, 48, 40 (56, 48)
	                                   * see one digit on every seed -- a counting pass, a scan and a walk over the whole array each, which move nothing and hand the same range on */
};
__device__ __forceinline__ uint64_t k2s_key(const Seed *gs, uint32_t src, uint32_t n)
{
	if(src >= n) { return K2S_SENTINEL_KEY; }
	const uint2 v = *(const uint2 *)&gs[src];          /* { upos, rid } */
	return (uint64_t)v.x | ((uint64_t)v.y << 32);
}
/* stable sort by the full key of every bucket of 2 .. 64 elements in [beg, end); bs / be = bucket begin / end by digit, e = entries (digit << 16 | index) */
__device__ __forceinline__ void k2s_small_buckets(LU32 *e, const LU32 *bs, const LU32 *be, uint32_t beg, uint32_t end, const Seed *gs, uint32_t n, int lane)
{
	uint32_t pos = beg;
	while(pos < end) {
		const uint32_t slot = pos + (uint32_t)lane; const bool valid = slot < end;
		const uint32_t x = valid ? e[slot] : 0u, d = x >> 16;
		const uint32_t b0 = valid ? bs[d] : 0u, b1 = valid ? be[d] : 0u;
		const uint64_t m_inc = __ballot(valid && b1 > pos + 64);
		uint32_t cut = m_inc ? pos + (uint32_t)__builtin_ctzll(m_inc) : (pos + 64 < end ? pos + 64 : end);
		if(cut == pos) { pos = (uint32_t)rdfirst((int)b1); continue; }           /* a bucket of more than 64 (it went on the stack): step over it */
		const bool act = slot < cut && b1 - b0 >= 2;
		uint32_t rank = 0;
		if(__ballot(act)) {
			const uint64_t key = act ? k2s_key(gs, x & 0xffffu, n) : 0ull;
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
__global__ void __launch_bounds__(64, MM_SHORT_KERNEL_WAVES) mm_sort_kernel(K2sArgs a)
{
	__builtin_amdgcn_s_setprio(2);          /* short and latency bound beside the extension waves of the other lanes (which run at 0 or 1, the few heaviest reads of a launch at 3) */
	extern __shared__ uint8_t lds_raw[];
	LU32 *e = (LU32 *)lds_raw;
	LU32 *cnt = (LU32 *)(lds_raw + a.lds_bytes - 4 * K2S_TABLE_WORDS), *bb = cnt + 256, *be = bb + 256, *stk = be + 256, *stsh = stk + K2S_STACK;
	const int lane = lane_id();
	const unsigned long long cy_begin = __builtin_amdgcn_s_memtime();
	while(true) {
		uint32_t wi = 0;
		if(lane == 0) { wi = atomicAdd(a.counter, 1u); }
		wi = (uint32_t)rdfirst((int)wi);
		if(wi >= a.n_work) { break; }
		ReadState *st = &a.st[a.work[wi]];
		const uint32_t n = (uint32_t)rdfirst((int)st->seed_n0), n_all = n + 1;
		if(n == 0 || n_all > K2S_MAX_N) { continue; }
		const uint32_t need = k2s_bytes(n_all);
		if(need <= a.n_lo || need > a.n_hi) { continue; }                    /* another size class */
		Seed *gs = a.seed_pool + rdfirst64(st->seed_off);
		uint32_t err = 0;
		for(uint32_t i = (uint32_t)lane; i < n_all; i += 64) { e[i] = i; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		uint32_t sp = 0;
		if(n_all <= 64) {
			/* one insertion sort over everything (ksort.h:126): a single "bucket" */
			if(lane == 0) { bb[0] = 0; be[0] = n_all; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			k2s_small_buckets(e, bb, be, 0, n_all, gs, n, lane);
		} else {
			/* (a level whose digit is the same on every seed leaves the range as it is -- the sentinel, the one element with another digit, already stands behind the others --
			 * and passes it on to the next level because it holds more than 64 elements: starting at start_shift on the range without the sentinel is the same walk) */
			if(lane == 0) { if(a.start_shift < 56u && n > 64u) { stk[0] = 0u | (n << 16); stsh[0] = a.start_shift; } else { stk[0] = 0u | (n_all << 16); stsh[0] = 56; } }
			sp = 1;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		while(sp > 0) {
			sp--;
			const uint32_t rg = (uint32_t)rdfirst((int)stk[sp]); const int sh = rdfirst((int)stsh[sp]);
			const uint32_t beg = rg & 0xffffu, end = rg >> 16, m = end - beg;
			/* digits of this level, histogram */
			for(int k = lane; k < 256; k += 64) { cnt[k] = 0; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			for(uint32_t i = beg + (uint32_t)lane; i < end; i += 64) {
				const uint32_t src = e[i] & 0xffffu;
				const uint32_t d = (uint32_t)(k2s_key(gs, src, n) >> sh) & 255u;
				e[i] = d << 16 | src;
				atomicAdd((uint32_t *)&cnt[d], 1u);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			const uint32_t d0 = (uint32_t)rdfirst((int)e[beg]) >> 16;
			if((uint32_t)rdfirst((int)cnt[d0]) == m) {
				/* every element has the same digit: the level leaves the range as it is */
				if(sh) { if(lane == 0) { stk[sp] = rg; stsh[sp] = (uint32_t)(sh > 8 ? sh - 8 : 0); } sp++; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				continue;
			}
			/* bucket bounds: each lane owns four consecutive buckets, one wave-wide exclusive scan */
			const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
			{
				uint32_t incl = c0 + c1 + c2 + c3;
				for(int dd = 1; dd < 64; dd <<= 1) { uint32_t o = (uint32_t)__shfl_up((int)incl, dd); if(lane >= dd) { incl += o; } }
				uint32_t acc = beg + incl - (c0 + c1 + c2 + c3);
				bb[4 * lane] = acc; acc += c0; be[4 * lane] = acc; bb[4 * lane + 1] = acc; acc += c1; be[4 * lane + 1] = acc;
				bb[4 * lane + 2] = acc; acc += c2; be[4 * lane + 2] = acc; bb[4 * lane + 3] = acc; acc += c3; be[4 * lane + 3] = acc;
			}
			/* non-empty buckets as four 64-bit masks (bucket 4 * lane + j -> bit lane of mask j) */
			const uint64_t nz0 = __ballot(c0 != 0), nz1 = __ballot(c1 != 0), nz2 = __ballot(c2 != 0), nz3 = __ballot(c3 != 0);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			/* how many elements already sit in their bucket decides how the permutation is replayed */
			uint32_t n_home = 0;
			for(uint32_t i0 = beg; i0 < end; i0 += 64) {
				const uint32_t i = i0 + (uint32_t)lane; bool home = false;
				if(i < end) { const uint32_t d = e[i] >> 16; home = i >= bb[d] && i < be[d]; }
				n_home += (uint32_t)__popcll(__ballot(home));
			}
			/*
			 * The in-place cycle-leader permutation (ksort.h:101-116), literally, on the 4-byte entries.  Buckets in ascending order; inside a bucket the
			 * cursor walks to its end, and every element that is not at home starts a cycle: it goes to the cursor of its own bucket, the element it
			 * displaces goes to the cursor of *its* bucket (whether it was at home there or not), until an element of the current bucket turns up.
			 */
			if(2 * n_home <= m) {
				/* mostly displaced elements: one lane, no hand-overs */
				if(lane == 0) {
					for(int l4 = 0; l4 < 64; l4++) {
						const uint32_t any = (uint32_t)((nz0 >> l4) & 1) | (uint32_t)((nz1 >> l4) & 1) << 1 | (uint32_t)((nz2 >> l4) & 1) << 2 | (uint32_t)((nz3 >> l4) & 1) << 3;
						for(int j = 0; j < 4; j++) {
							if(!((any >> j) & 1)) { continue; }
							const uint32_t k = (uint32_t)(4 * l4 + j);
							uint32_t b = bb[k]; const uint32_t ee = be[k];
							while(b != ee) {
								const uint32_t x = e[b];
								if((x >> 16) == k) { b++; continue; }
								uint32_t tmp = x, l_ = x >> 16;
								do { const uint32_t p = bb[l_]; const uint32_t y = e[p]; e[p] = tmp; bb[l_] = p + 1; tmp = y; l_ = y >> 16; } while(l_ != k);
								e[b] = tmp; b++;
							}
						}
					}
				}
			} else {
				/* mostly at home (seeds of one strand arrive in diagonal order): stretches at home are stepped over 64 at a time, lane 0 runs the cycles */
				for(int l4 = 0; l4 < 64; l4++) {
					const uint32_t any = (uint32_t)((nz0 >> l4) & 1) | (uint32_t)((nz1 >> l4) & 1) << 1 | (uint32_t)((nz2 >> l4) & 1) << 2 | (uint32_t)((nz3 >> l4) & 1) << 3;
					for(int j = 0; j < 4; j++) {
						if(!((any >> j) & 1)) { continue; }
						const uint32_t k = (uint32_t)(4 * l4 + j);
						uint32_t b = (uint32_t)rdfirst((int)bb[k]); const uint32_t ee = (uint32_t)rdfirst((int)be[k]);
						while(b != ee) {
							const uint32_t idx = b + (uint32_t)lane;
							const bool away = idx < ee && (e[idx] >> 16) != k;
							const uint64_t m_away = __ballot(away);
							if(m_away == 0) { b = b + 64 < ee ? b + 64 : ee; continue; }
							b += (uint32_t)__builtin_ctzll(m_away);
							if(lane == 0) {
								uint32_t tmp = e[b], l_ = tmp >> 16;
								do { const uint32_t p = bb[l_]; const uint32_t y = e[p]; e[p] = tmp; bb[l_] = p + 1; tmp = y; l_ = y >> 16; } while(l_ != k);
								e[b] = tmp;
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
							b++;
						}
					}
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			for(int k = lane; k < 256; k += 64) { bb[k] = k == 0 ? beg : be[k - 1]; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			if(sh) {
				const int ns = sh > 8 ? sh - 8 : 0;
				/* large buckets go back on the stack (order of siblings is irrelevant), small ones are sorted by rank */
				for(int k0 = 0; k0 < 256; k0 += 64) {
					const int k = k0 + lane; const uint32_t nb = be[k] - bb[k];
					const uint64_t mm = __ballot(nb > 64);
					const uint32_t slot = sp + (uint32_t)__popcll(mm & ((1ull << lane) - 1));
					if(nb > 64) { if(slot < K2S_STACK) { stk[slot] = bb[k] | (be[k] << 16); stsh[slot] = (uint32_t)ns; } else { err |= ERR_STACK; } }
					sp += (uint32_t)__popcll(mm);
				}
				if(sp > K2S_STACK) { sp = K2S_STACK; }
				k2s_small_buckets(e, bb, be, beg, end, gs, n, lane);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		/* the seeds move once: through the (still unused) leaf half of the read's region, then back in order */
		for(uint32_t i = (uint32_t)lane; i < n_all; i += 64) {
			const uint32_t src = e[i] & 0xffffu;
			Seed v = Seed{ 0x80000000u, 0x7fffffffu, 0x80000000u, 0x7fffffffu };
			if(src < n) { v = gs[src]; }
			gs[n_all + i] = v;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		for(uint32_t i = (uint32_t)lane; i < n_all; i += 64) { gs[i] = gs[n_all + i]; }
		if(__ballot(err != 0) && lane == 0) { st->err |= ERR_STACK; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	if(lane == 0) { atomicAdd(&a.prof[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - cy_begin)); }
}


/* -----------------------------------------------------------------------------------------------------
 * K2p + K2c: mm_chain_seeds (minialign.c:3547-3625) over the sorted seed array, in two launches.
 *
 *   K2p  mm_chain_scan_kernel   What one step of the chain sweep finds from seed i -- the last seed inside the shrinking window (succ, 0 = none) and the
 *                               first seed it passes over (seen) -- depends on the sorted array alone, not on what earlier chains have marked.  So the
 *                               window scans of all seeds run first, one seed per lane, straight from HBM / L2 (neighbouring lanes scan overlapping
 *                               stretches), with no LDS and therefore at full occupancy.  pdiff() of the reference is evaluated on the window it has just
 *                               updated and is always 0: "the largest (pdiff, sid)" is simply the last accepted sid.  Results go to the (still unused) leaf
 *                               half of the read's seed region, 8 B per seed.
 *   K2c  mm_chain_kernel        the sequential sweep itself -- a pointer chase over those tables, one lane per read does it -- on a compact image of the read
 *                               in LDS: 12 B per seed ({ succ | seen << 16, leaf mark } read in one piece, upos + vpos) and 12 B per leaf / chain, half of what the
 *                               16-byte seeds and leaves took, so that a CU holds four to six reads; nothing in the loop touches HBM.  Seeds' marks, leaves
 *                               and chain roots are written out afterwards, in parallel; mm_circularize, the root sort and the prediction for the carried
 *                               reference length follow as before.
 * ----------------------------------------------------------------------------------------------------- */
typedef __attribute__((address_space(3))) uint16_t LU16;
struct K2pArgs { ReadState *st; const uint32_t *work; uint32_t n_work; Seed *seed_pool; uint32_t twlen; uint32_t *counter; unsigned long long *prof; };
__global__ void __launch_bounds__(64) mm_chain_scan_kernel(K2pArgs a)
{
	const int lane = lane_id();
	const unsigned long long cy_begin = __builtin_amdgcn_s_memtime();
	const int32_t tw = (int32_t)a.twlen;
	while(true) {
		uint32_t wi = 0;
		if(lane == 0) { wi = atomicAdd(a.counter, 1u); }
		wi = (uint32_t)rdfirst((int)wi);
		if(wi >= a.n_work) { break; }
		const ReadState *st = &a.st[a.work[wi]];
		const uint32_t n = (uint32_t)rdfirst((int)st->seed_n0), n_all = n + 1;
		if(n == 0 || n_all > K2S_MAX_N) { continue; }
		const Seed *s = a.seed_pool + rdfirst64(st->seed_off);
		uint2 *ss = (uint2 *)(a.seed_pool + rdfirst64(st->seed_off) + n_all);
		const uint32_t tsid = n;
		for(uint32_t i0 = 0; i0 < tsid; i0 += 64) {
			const uint32_t i = i0 + (uint32_t)lane;
			if(i < tsid) {
				V4 wv = add_win(load_pv(s[i]), tw);
				uint32_t last = 0, first_out = 0xffffffffu;
				for(uint32_t jx = i + 1; jx <= tsid; jx++) {          /* the sentinel at tsid always ends the scan */
					const V4 fv = load_pv(s[jx]);
					if(inside_wv(wv, fv)) { wv = update_wv(wv, fv); last = jx; continue; }
					first_out = first_out < jx ? first_out : jx;
					if(!inside_uub(wv, fv)) { break; }
				}
				ss[i] = uint2{ last, first_out };
			}
		}
	}
	if(lane == 0) { atomicAdd(&a.prof[1], (unsigned long long)(__builtin_amdgcn_s_memtime() - cy_begin)); }
}
/* the largest LDS image mm_chain_kernel takes: what a CU has left beside eight workgroups of the extension kernel (K3_LDS_BYTES each) -- a launch that asks for all
 * 160 KB finds no CU to start on until an extension launch of another lane ends, whether it has a read to sweep or not; larger reads go the in-HBM way of K2a */
#ifndef K2C_MAX_LDS_KB
#define K2C_MAX_LDS_KB 108u
#endif
__host__ __device__ inline uint32_t k2c_leafcap(uint32_t n_all, uint32_t shift) { return (n_all >> shift) + 64; }
__host__ __device__ inline uint32_t k2c_bytes(uint32_t n_all, uint32_t leafcap) { return 12u * ((n_all + 63u) & ~63u) + 12u * ((leafcap + 63u) & ~63u) + 1536u * 4u; }

/* -----------------------------------------------------------------------------------------------------
 * K2w: the same sweep, one LANE per read, everything in HBM / L2.  The sweep is a chain of dependent look-ups (one round trip per chained seed) whichever memory
 * it runs in; in LDS a CU holds four or five reads' images, i.e. four or five chases in flight per CU, and the launches wait for LDS and wave slots beside the
 * extension waves of the other lanes (17 ms alone, three times that in the mix).  Here every read of the batch is in flight at once -- 64 per wave, a few hundred
 * waves, no LDS -- and a round trip costs an HBM access instead of an LDS access: the launch lasts as long as the read with the most seeds (a few thousand steps).
 * Per step the step table entry ss[nx] (K2p) and the mark gs[nx].lid are fetched together; marks are written in place (n_all + leaf number: what mm_chain_seeds
 * leaves), leaves and chain roots go through a scratch area the size of the read's seed region (they would overwrite the step table where the leaves end up) and are
 * written out behind the sweep.  Reads with more than K2S_MAX_N seeds stay with K2a.
 * ----------------------------------------------------------------------------------------------------- */
struct K2wArgs {
	ReadState *st; const uint32_t *work; uint32_t n_work;
	Seed *seed_pool; Root *root_pool; uint8_t *scratch;          /* scratch: 16 B per seed of a read (its leaf tables, later the table of its root sort), handed out as the reads come */
	unsigned long long *scratch_top; uint64_t scratch_bytes;      /* cursor (zeroed before the launch) and size: the host sizes it for a third of the seed pool's capacity -- reads carry
	                                                               * a fifth to a twelfth of their caps -- and a read that finds it exhausted reports ERR_SEED_CAP (the batch is redone with larger pools) */
	double mcoef; uint32_t min_score, twlen;
	const uint32_t *seq_len; const uint8_t *seq_circ;
};
__global__ void __launch_bounds__(64, MM_SHORT_KERNEL_WAVES) mm_chain_sweep_kernel(K2wArgs a)
{
	__builtin_amdgcn_s_setprio(2);
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if(t >= a.n_work) { return; }
	ReadState *st = &a.st[a.work[t]];
	const uint32_t n = st->seed_n0, n_all = n + 1;
	if(n == 0) { st->n_seed = 0; st->n_root = 0; st->pred_rid = gaba::NIL; return; }
	if(n_all > K2S_MAX_N) { return; }
	Seed *gs = a.seed_pool + st->seed_off; Root *c = a.root_pool + st->root_off;
	const uint2 *ss = (const uint2 *)(gs + n_all);
	const unsigned long long sc_need = 16ull * n_all, sc_off = atomicAdd(a.scratch_top, sc_need);
	if(sc_off + sc_need > a.scratch_bytes) { st->err |= ERR_SEED_CAP; st->n_seed = n; st->n_root = 0; st->pred_rid = gaba::NIL; return; }
	uint16_t *lrs = (uint16_t *)(a.scratch + sc_off), *lls = lrs + n_all, *lcid = lls + n_all, *rlid = lcid + n_all;
	uint32_t *rplen = (uint32_t *)(rlid + n_all + (n_all & 1));
	st->n_seed = n; st->n_root = 0; st->pred_rid = gaba::NIL;
	const uint32_t UNM = 0x7fffffffu;
	uint32_t ncid = 0, nleaf = 0, nlsid = 0; const uint32_t tsid = n;
	while(nlsid < tsid) {
		const uint32_t lf = nleaf++, lsid0 = nlsid;
		uint2 x = ss[lsid0]; const Seed s0 = gs[lsid0];
		const uint32_t plen0 = s0.upos + s0.vpos; uint32_t scnt = 1;
		lrs[lf] = (uint16_t)lsid0; lls[lf] = (uint16_t)lsid0; lcid[lf] = 0xffffu;
		uint32_t nrsid = lsid0, hl = s0.lid;          /* hl: the mark of the seed the chain stands on when it stops */
		nlsid = 0xffffffffu;
		while(true) {
			const uint32_t nx = x.x, sm = x.y;
			nlsid = nlsid < sm ? nlsid : sm;
			if(nx == 0) { break; }                        /* nothing inside the window: the chain ends on the seed it stands on */
			const uint2 ex = ss[nx]; const uint32_t ey = gs[nx].lid;          /* (two independent loads, one round trip) */
			nrsid = nx; hl = ey;
			if(ey != UNM) { break; }                      /* marked by an earlier leaf: the chain runs into that one */
			gs[nx].lid = n_all + lf; hl = n_all + lf;
			scnt++;
			if(nlsid <= nx) { nlsid = 0xffffffffu; }
			x = ex;
		}
		if(nrsid == lsid0) { continue; }
		uint32_t cid = 0xffffu;
		if(hl != UNM && hl - n_all < lf) {
			nrsid = lrs[hl - n_all];                      /* leaf.rsid */
			cid = lcid[gs[nrsid].lid - n_all];            /* leaf.cid of the leaf that marks it */
		}
		bool fresh = false;
		if(cid == 0xffffu) { cid = ncid++; fresh = true; }
		const Seed se = gs[nrsid]; const uint32_t eu = se.upos + se.vpos;
		const uint32_t plen = (uint32_t)OFS((int32_t)d2u32((1.0 - 1.0 / (double)scnt) * (double)(uint32_t)(eu - plen0)));
		uint32_t best = fresh ? (uint32_t)OFS(0) : rplen[cid];
		if(fresh) { rlid[cid] = (uint16_t)lf; }
		lcid[lf] = (uint16_t)cid; lrs[lf] = (uint16_t)nrsid;
		if(plen < best) { best = plen; rlid[cid] = (uint16_t)lf; }
		if(fresh || plen == best) { rplen[cid] = best; }
	}
	/* write out: the sentinel stays where the sort put it, the leaves { rsid, rid, lsid, cid } (over the step table, which is done with), the chain roots */
	for(uint32_t lf = 0; lf < nleaf; lf++) {
		const uint32_t ls = lls[lf], ci = lcid[lf];
		gs[n_all + lf] = Seed{ (uint32_t)lrs[lf], gs[ls].rid, ls, ci == 0xffffu ? 0xffffffffu : ci };
	}
	for(uint32_t ci = 0; ci < ncid; ci++) { c[ci] = Root{ rplen[ci], n_all + (uint32_t)rlid[ci] }; }
	const uint32_t nlid = n_all + nleaf;
	st->seed_n = nlid; st->n_root = ncid;
	if(ncid) {
		if(a.seq_circ) { circularize(gs, c, n, nlid, ncid, a.seq_len, a.seq_circ, a.twlen); }
		if(ncid <= 64) { ins_sort_64((U64R *)c, (U64R *)c + ncid); }          /* longest first (minialign.c:3719); radix_sort_64x is an insertion sort up to 64 elements */
		else {
			/* the read's own scratch area is free again (leaves and roots are written out): 4 words per seed, i.e. at least 8 per chain -- the
			 * 512 bucket words and 3 per pending range (at most one per 65 chains) of radix_sort_64x fit from 65 chains on */
			if(!radix_sort_64((U64R *)c, ncid, (uint32_t *)lrs, 4u * n_all)) { st->err |= ERR_STACK; }
		}
		uint32_t pred = gaba::NIL, n_pass = 0, w_pass = 0;          /* chains that pass the length test of mm_search_load_root and their summed lengths: what the extension will cost */
		for(uint32_t kq = 0; kq < ncid; kq++) {
			const uint32_t pl = (uint32_t)OFS((int32_t)c[kq].plen);
			if(pl * a.mcoef < 2.0 * a.min_score) { break; }
			pred = gs[gs[c[kq].lid].upos].rid; n_pass++; w_pass += pl;
		}
		st->pred_rid = pred; st->n_pass = n_pass; st->w_pass = w_pass;
	}
}

/* -----------------------------------------------------------------------------------------------------
 * K2a: the same stage, one *wavefront* per read with the seed / leaf array staged in LDS (first round only; reads whose
 * arrays do not fit the LDS budget, and the rescue rounds, take the lane-per-read kernel above).
 *   - radix levels whose digit is constant over the range are identity permutations and are skipped (one parallel
 *     histogram decides); the cycle-leader permutation itself stays serial (lane 0, in LDS); the <= 64-element buckets
 *     left by a level are insertion-sorted one bucket per lane (insertion sort is stable, so any order of buckets and any
 *     stable method give the reference's result);
 *   - the chaining sweep keeps the reference's sequential semantics but tests 64 candidate seeds per step.
 * ----------------------------------------------------------------------------------------------------- */
struct K2aArgs {
	ReadState *st; const uint32_t *work; uint32_t n_work;
	Seed *seed_pool; Root *root_pool;
	uint32_t lds_bytes;               /* dynamic LDS of this launch (tables included); <= 1536 * 4: sort in place in HBM */
	uint32_t n_lo, n_hi;              /* this launch takes the reads with n_lo < k2a_bytes(seed_n) <= n_hi (size class, bytes) */
	uint32_t retry;                   /* 1: take the reads whose leaf area overflowed in their class (flagged n_root = ~0), with room for 2 (n + 1) */
	uint32_t leaf_shift;              /* first attempt: a leaf area of (n + 1) >> leaf_shift elements (2: a quarter; the host lowers it when a batch needed many retries) */
	uint32_t big_only;                /* 1: only the reads mm_sort_kernel / mm_chain_kernel leave out (n + 1 > K2S_MAX_N) */
	uint32_t presorted;               /* 1: mm_sort_kernel has already sorted the seed arrays of the reads it takes (n + 1 <= K2S_MAX_N) */
	uint32_t *counter;                /* work-list cursor of this launch */
	uint32_t twlen; double mcoef; uint32_t min_score;
	const uint32_t *seq_len; const uint8_t *seq_circ;   /* reference lengths and circular flags (NULL: no circular reference) for mm_circularize */
	unsigned long long *prof;         /* [0] sort [1] chain [2] whole wave (s_memtime ticks) [3] reads that did not fit the LDS [5] reads whose leaf area overflowed (retried) */
};
/* elements a read is given in its first attempt: the seeds, the sentinel and a leaf area of a quarter of that (the worst case of
 * one leaf per seed is left to the retry launch) */
__host__ __device__ inline uint32_t k2a_need(uint32_t seed_n, uint32_t leaf_shift) { return (seed_n + 1) + ((seed_n + 1) >> leaf_shift) + 64; }
/* LDS bytes of a read: 16 B per element + the two u32 step tables of the chain sweep + the sort tables */
__host__ __device__ inline uint32_t k2a_bytes(uint32_t seed_n, uint32_t elems) { return 16u * elems + 8u * (seed_n + 1) + 1536u * 4u; }

__device__ __forceinline__ uint64_t lkey(const LSeed *p) { return (uint64_t)p->upos | ((uint64_t)p->rid << 32); }
__device__ __forceinline__ Seed lds_ld(const LSeed *p) { Seed r; r.upos = p->upos; r.rid = p->rid; r.vpos = p->vpos; r.lid = p->lid; return r; }
__device__ __forceinline__ void lds_st(LSeed *p, const Seed &v) { p->upos = v.upos; p->rid = v.rid; p->vpos = v.vpos; p->lid = v.lid; }
__device__ __forceinline__ uint64_t lkey(const Seed *p) { return (uint64_t)p->upos | ((uint64_t)p->rid << 32); }
__device__ __forceinline__ Seed lds_ld(const Seed *p) { return *p; }
__device__ __forceinline__ void lds_st(Seed *p, const Seed &v) { *p = v; }
template<typename S>
__device__ __forceinline__ void lds_ins_sort(S *beg, S *end)
{
	for(S *i = beg + 1; i < end; ++i) {
		uint64_t ki = lkey(i);
		if(ki < lkey(i - 1)) {
			Seed tmp = lds_ld(i); S *j;
			for(j = i; j > beg && ki < lkey(j - 1); --j) { lds_st(j, lds_ld(j - 1)); }
			lds_st(j, tmp);
		}
	}
}

/* sort + chain over a seed array that lives either in LDS (S = LSeed) or in HBM (S = Seed); returns false if the leaf area overflowed */
template<typename S>
__device__ __forceinline__ bool sort_chain_wave(S *s, uint32_t cap, uint32_t seed_n, LU32 *cnt, LU32 *bb, LU32 *be, LU32 *stack,
	Root *c, const K2aArgs &a, int lane, uint32_t &nlid_out, uint32_t &ncid_out, unsigned long long &cy_sort, unsigned long long &cy_chain,
	const bool pre, LU32 *succ, LU32 *seen, const bool sorted, const bool chain = true)
{
	const uint32_t n_all = seed_n + 1;
	const unsigned long long cy0 = __builtin_amdgcn_s_memtime();
		/* ---- radix_sort_128x ---- */
		if(sorted) { /* done by mm_sort_kernel */ }
		else if(n_all <= 64) { if(lane == 0) { lds_ins_sort(s, s + n_all); } }
		else {
			uint32_t sp = 1;
			if(lane == 0) { stack[0] = 0; stack[1] = n_all; stack[2] = 56; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			while(sp > 0) {
				sp--;
				const uint32_t beg = (uint32_t)rdfirst((int)stack[3 * sp]), end = (uint32_t)rdfirst((int)stack[3 * sp + 1]); const int sh = rdfirst((int)stack[3 * sp + 2]);
				for(int k = lane; k < 256; k += 64) { cnt[k] = 0; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				for(uint32_t i = beg + (uint32_t)lane; i < end; i += 64) { atomicAdd((uint32_t *)&cnt[(lkey(&s[i]) >> sh) & 255], 1u); }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				/* a level whose elements all share the digit leaves the range untouched */
				const uint32_t d0 = (uint32_t)((lkey(&s[beg]) >> sh) & 255);
				const bool single = (uint32_t)rdfirst((int)cnt[d0]) == end - beg;
				if(single) {
					if(sh) { if(lane == 0) { stack[3 * sp] = beg; stack[3 * sp + 1] = end; stack[3 * sp + 2] = (uint32_t)(sh > 8 ? sh - 8 : 0); } sp++; }
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
					continue;
				}
				/* bucket bounds: each lane owns four consecutive buckets, one wave-wide exclusive scan */
				{
					const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
					uint32_t incl = c0 + c1 + c2 + c3;
					for(int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl_up((int)incl, d); if(lane >= d) { incl += o; } }
					uint32_t acc = beg + incl - (c0 + c1 + c2 + c3);
					bb[4 * lane] = acc; acc += c0; be[4 * lane] = acc; bb[4 * lane + 1] = acc; acc += c1; be[4 * lane + 1] = acc;
					bb[4 * lane + 2] = acc; acc += c2; be[4 * lane + 2] = acc; bb[4 * lane + 3] = acc; acc += c3; be[4 * lane + 3] = acc;
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				{
					/* the in-place cycle-leader permutation (ksort.h:101-116): inherently sequential, and its exact element order is
					 * what decides ties, so it is replayed as is -- with one shortcut that changes nothing: a stretch of elements that
					 * already sit in their bucket only advances that bucket's cursor, so the stretch is found 64 elements at a time
					 * (one ballot) and the serial code (lane 0) runs for the displaced elements only.  Seeds arrive roughly in
					 * diagonal order, i.e. nearly sorted for forward-strand hits. */
					for(int k = 0; k < 256; k++) {
						uint32_t b = (uint32_t)rdfirst((int)bb[k]); const uint32_t e = (uint32_t)rdfirst((int)be[k]);
						while(b != e) {
							const uint32_t idx = b + (uint32_t)lane;
							const bool away = idx < e && (int)((lkey(&s[idx]) >> sh) & 255) != k;
							const uint64_t m_away = __ballot(away);
							if(m_away == 0) { b = b + 64 < e ? b + 64 : e; continue; }
							b += (uint32_t)__builtin_ctzll(m_away);               /* everything in front of it is home */
							if(lane == 0) {
								Seed tmp = lds_ld(&s[b]), swp;
								int l_ = (int)((((uint64_t)tmp.upos | ((uint64_t)tmp.rid << 32)) >> sh) & 255);
								do { swp = tmp; uint32_t d = bb[l_]; tmp = lds_ld(&s[d]); lds_st(&s[d], swp); bb[l_] = d + 1;
								     l_ = (int)((((uint64_t)tmp.upos | ((uint64_t)tmp.rid << 32)) >> sh) & 255); } while(l_ != k);
								lds_st(&s[b], tmp);
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
							b++;
						}
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				for(int k = lane; k < 256; k += 64) { bb[k] = k == 0 ? beg : be[k - 1]; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				if(sh) {
					const int ns = sh > 8 ? sh - 8 : 0;
					/* large buckets go back on the stack (order of siblings is irrelevant), small ones are sorted one per lane */
					for(int k0 = 0; k0 < 256; k0 += 64) {
						const int k = k0 + lane; const uint32_t nb = be[k] - bb[k];
						const uint64_t m = __ballot(nb > 64);
						if(nb > 64) { const uint32_t slot = sp + (uint32_t)__popcll(m & ((1ull << lane) - 1)); stack[3 * slot] = bb[k]; stack[3 * slot + 1] = be[k]; stack[3 * slot + 2] = (uint32_t)ns; }
						sp += (uint32_t)__popcll(m);
					}
					for(int k = lane; k < 256; k += 64) { uint32_t n = be[k] - bb[k]; if(n > 1 && n <= 64) { lds_ins_sort(s + bb[k], s + be[k]); } }
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		const unsigned long long cy1 = __builtin_amdgcn_s_memtime(); cy_sort += cy1 - cy0;
		if(!chain) { nlid_out = seed_n + 1; ncid_out = 0; return true; }

		/* ---- mm_chain_seeds (minialign.c:3547-3625) ---- */
		uint32_t ncid = 0, nlid = seed_n + 1, nlsid = 0; const uint32_t tsid = seed_n;
		const int32_t tw = (int32_t)a.twlen;
		bool overflow = false;
		if(pre) {
			/*
			 * What one step of the chain sweep finds from a seed i -- the last seed inside the shrinking window (succ, 0 = none)
			 * and the first seed it passes over (seen) -- depends on the sorted array alone, not on what earlier chains have
			 * marked (the marks only decide where a chain stops).  So the window scans of all seeds run here, one seed per lane,
			 * and the sequential sweep below is left with a pointer chase.  pdiff() of the reference is evaluated on the window
			 * it has just updated and is therefore always 0: "the largest (pdiff, sid)" is simply the last accepted sid.
			 */
			for(uint32_t i0 = 0; i0 < tsid; i0 += 64) {
				const uint32_t i = i0 + (uint32_t)lane;
				if(i < tsid) {
					V4 wv = add_win(load_pv(lds_ld(&s[i])), tw);
					uint32_t last = 0, first_out = 0xffffffffu;
					for(uint32_t jx = i + 1; jx <= tsid; jx++) {          /* the sentinel at tsid always ends the scan */
						const V4 fv = load_pv(lds_ld(&s[jx]));
						if(inside_wv(wv, fv)) { wv = update_wv(wv, fv); last = jx; continue; }
						first_out = first_out < jx ? first_out : jx;
						if(!inside_uub(wv, fv)) { break; }
					}
					succ[i] = last; seen[i] = first_out;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		while(nlsid < tsid) {
			const uint32_t lid = nlid++;
			if(lid >= cap) { overflow = true; break; }
			/* the seed the chain currently stands on is carried in (uniform) registers: each step takes it from the lanes of the
			 * chunk it has just scanned instead of reading it back from the array */
			Seed rs_ = lds_ld(&s[nlsid]);
			int32_t rs_u = rdfirst((int)rs_.upos), rs_r = rdfirst((int)rs_.rid), rs_v = rdfirst((int)rs_.vpos);
			const uint32_t l_rid = (uint32_t)rs_r;
			if(lane == 0) { lds_st(&s[lid], Seed{ nlsid, l_rid, nlsid, 0xffffffffu }); }
			uint32_t plen = (uint32_t)(rs_u + rs_v), scnt = 1;
			const uint32_t lsid0 = nlsid;
			uint64_t nrsid = nlsid; nlsid = 0xffffffffu;
			while(pre) {
				/* pointer chase over the precomputed steps (same bookkeeping as the scanning form below) */
				const uint32_t rsid = (uint32_t)nrsid;
				const uint32_t nx = (uint32_t)rdfirst((int)succ[rsid]), sm = (uint32_t)rdfirst((int)seen[rsid]);
				nlsid = nlsid < sm ? nlsid : sm;
				if(nx == 0) { nrsid = rsid; break; }
				const uint32_t cl = (uint32_t)rdfirst((int)s[nx].lid);
				nrsid = nx;
				if(cl != 0x7fffffffu) { break; }
				if(lane == 0) { s[nx].lid = lid; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				scnt++;
				if(nlsid <= nx) { nlsid = 0xffffffffu; }
			}
			while(!pre) {
				const uint32_t rsid = (uint32_t)nrsid; nrsid = 0;
				V4 wv = add_win(V4{ rs_u, rs_r, rs_v, rs_v }, tw);
				int32_t b_u = 0, b_r = 0, b_v = 0; uint32_t b_lid = 0;          /* record of the seed nrsid points at */
				bool stop = false;
				for(uint32_t base = rsid + 1; !stop; base += 64) {
					const uint32_t sid = base + (uint32_t)lane;
					const bool valid = sid <= tsid;                      /* the sentinel at tsid always ends the scan */
					Seed cs = Seed{ 0, 0x7fffffffu, 0, 0 }; if(valid) { cs = lds_ld(&s[sid]); }
					V4 fv = load_pv(cs);
					uint64_t pending = __ballot(valid);
					while(pending) {
						const bool mine = (pending >> lane) & 1;
						const bool in = mine && inside_wv(wv, fv);
						const bool brk = mine && !in && !inside_uub(wv, fv);
						const uint64_t m_in = __ballot(in), m_brk = __ballot(brk), m_out = __ballot(mine && !in);
						const uint32_t f_in = m_in ? (uint32_t)__builtin_ctzll(m_in) : 64u, f_brk = m_brk ? (uint32_t)__builtin_ctzll(m_brk) : 64u;
						const uint32_t lim = f_in < f_brk ? f_in : f_brk;
						/* non-inside candidates met before the next event (the breaking one included) pull nlsid down */
						const uint64_t below = lim >= 63 ? ~0ull : ((2ull << lim) - 1);
						const uint64_t m_seen = m_out & below;
						if(m_seen) { const uint32_t fs = base + (uint32_t)__builtin_ctzll(m_seen); nlsid = nlsid < fs ? nlsid : fs; }
						if(f_brk < f_in) { stop = true; break; }
						if(f_in == 64) { break; }                       /* nothing left in this chunk */
						V4 af = V4{ rdlane(fv.e0, (int)f_in), rdlane(fv.e1, (int)f_in), rdlane(fv.e2, (int)f_in), rdlane(fv.e3, (int)f_in) };
						wv = update_wv(wv, af);
						const int64_t di = (int64_t)(((uint64_t)(int64_t)pdiff(wv, af) << 32) | (uint64_t)(base + f_in));
						if(di > (int64_t)nrsid) { nrsid = (uint64_t)di; b_u = af.e0; b_r = af.e1; b_v = af.e2; b_lid = (uint32_t)rdlane((int)cs.lid, (int)f_in); }
						pending &= f_in >= 63 ? 0ull : ~((2ull << f_in) - 1);
					}
					if(!stop && base + 64 > tsid + 1) { stop = true; }       /* ran past the sentinel (cannot happen: the sentinel breaks) */
				}
				if(nrsid == 0) { nrsid = rsid; break; }
				const uint32_t cand = (uint32_t)nrsid;
				if(b_lid != 0x7fffffffu) { nrsid = cand; break; }           /* s[cand].lid: nothing ahead of the chain has been marked by this leaf */
				if(lane == 0) { s[cand].lid = lid; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				scnt++;
				if((uint64_t)nlsid <= nrsid) { nlsid = 0xffffffffu; }
				rs_u = b_u; rs_r = b_r; rs_v = b_v;
			}
			if(nrsid == lsid0) { continue; }
			uint32_t cid = 0xffffffffu;
			const uint32_t hl = (uint32_t)rdfirst((int)s[nrsid].lid);
			if(hl < lid) {
				nrsid = (uint32_t)rdfirst((int)s[hl].upos);                       /* leaf.rsid */
				cid = (uint32_t)rdfirst((int)s[(uint32_t)rdfirst((int)s[nrsid].lid)].lid);   /* leaf.cid */
			}
			bool fresh = false;
			if(cid == 0xffffffffu) { cid = ncid++; fresh = true; }
			const uint32_t eu = (uint32_t)rdfirst((int)(s[nrsid].upos + s[nrsid].vpos));
			plen = (uint32_t)OFS((int32_t)d2u32((1.0 - 1.0 / (double)scnt) * (double)(uint32_t)(eu - plen)));
			if(lane == 0) {
				if(fresh) { c[cid] = Root{ (uint32_t)OFS(0), lid }; }
				s[lid].lid = cid; s[lid].upos = (uint32_t)nrsid;
				if(plen < c[cid].plen) { c[cid] = Root{ plen, lid }; }
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
	nlid_out = nlid; ncid_out = ncid;
	cy_chain += __builtin_amdgcn_s_memtime() - cy1;
	return !overflow;
}

__global__ void __launch_bounds__(64) mm_sort_chain_lds_kernel(K2aArgs a)
{
	extern __shared__ uint8_t lds_raw[];
	/* [seeds + leaves: cap x 16 B][succ, seen: (seed_n + 1) x 4 B each][tables: 1536 words at the end of the block] */
	LSeed *ls = (LSeed *)lds_raw;
	LU32 *cnt = (LU32 *)(lds_raw + a.lds_bytes - 1536 * 4);      /* 256 counters */
	LU32 *bb = cnt + 256, *be = bb + 256;             /* bucket begin / end */
	LU32 *stack = be + 256;                           /* pending ranges: (beg, end, shift) x 256 */
	const int lane = lane_id();
	unsigned long long cy_sort = 0, cy_chain = 0, n_big = 0; const unsigned long long cy_begin = __builtin_amdgcn_s_memtime();
	while(true) {
		uint32_t wi = 0;
		if(lane == 0) { wi = atomicAdd(a.counter, 1u); }
		wi = (uint32_t)rdfirst((int)wi);
		if(wi >= a.n_work) { break; }
		ReadState *st = &a.st[a.work[wi]];
		const uint32_t seed_n = (uint32_t)rdfirst((int)st->seed_n0);     /* not seed_n: launches of other classes update that concurrently */
		/* big_only: what mm_chain_kernel cannot take -- more than K2S_MAX_N seeds, an LDS image of more than 160 KB, or leaves that did not fit even the retry */
		if(a.big_only == 2 && seed_n + 1 <= K2S_MAX_N) { continue; }          /* (the lane-per-read sweep took everything else) */
		if(a.big_only == 1 && seed_n + 1 <= K2S_MAX_N && k2c_bytes(seed_n + 1, k2c_leafcap(seed_n + 1, a.leaf_shift)) <= K2C_MAX_LDS_KB * 1024u && (uint32_t)rdfirst((int)st->n_root) != 0xfffffffeu) { continue; }
		if(seed_n == 0) { if(a.n_lo == 0 && !a.retry && lane == 0) { st->n_seed = 0; st->n_root = 0; st->pred_rid = gaba::NIL; } continue; }
		bool fits; uint32_t lcap = 0;
		if(a.retry) {
			if((uint32_t)rdfirst((int)st->n_root) != 0xffffffffu) { continue; }
			fits = k2a_bytes(seed_n, 2 * (seed_n + 1)) <= a.lds_bytes;
		} else {
			const uint32_t need = k2a_bytes(seed_n, k2a_need(seed_n, a.leaf_shift));
			if(need <= a.n_lo || need > a.n_hi) { continue; }              /* another size class */
			fits = a.lds_bytes > 1536 * 4;
		}
		if(fits) { lcap = (a.lds_bytes - 1536u * 4u - 8u * (seed_n + 1)) / 16u; }      /* all the room of the class goes to the leaf area */
		LU32 *succ = (LU32 *)(ls + lcap), *seen = succ + (seed_n + 1);
		Seed *gs = a.seed_pool + rdfirst64(st->seed_off);
		Root *c = a.root_pool + rdfirst64(st->root_off);
		const uint32_t gcap = (uint32_t)rdfirst((int)st->seed_cap);
		if(lane == 0) { st->n_seed = seed_n; st->n_root = 0; st->pred_rid = gaba::NIL; }
		uint32_t nlid = 0, ncid = 0; bool ok;
		if(fits) {
			for(uint32_t i = (uint32_t)lane; i < seed_n; i += 64) { lds_st(&ls[i], gs[i]); }
			if(lane == 0) { lds_st(&ls[seed_n], Seed{ 0x80000000u, 0x7fffffffu, 0x80000000u, 0x7fffffffu }); }      /* sentinel, minialign.c:3531 */
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			ok = sort_chain_wave<LSeed>(ls, lcap, seed_n, cnt, bb, be, stack, c, a, lane, nlid, ncid, cy_sort, cy_chain, true, succ, seen, a.presorted && seed_n + 1 <= K2S_MAX_N);
			if(ok) { for(uint32_t i = (uint32_t)lane; i < nlid; i += 64) { gs[i] = lds_ld(&ls[i]); } }
		} else {
			/* too large for LDS: same algorithm in place in HBM */
			if(lane == 0) { gs[seed_n] = Seed{ 0x80000000u, 0x7fffffffu, 0x80000000u, 0x7fffffffu }; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			n_big++;
			ok = sort_chain_wave<Seed>(gs, gcap, seed_n, cnt, bb, be, stack, c, a, lane, nlid, ncid, cy_sort, cy_chain, false, succ, seen, a.presorted && seed_n + 1 <= K2S_MAX_N);
		}
		if(!ok) {
			/* leaf area exhausted: the seed array in HBM is untouched (LDS case), so the retry launch redoes the read with full room */
			if(lane == 0) { if(!a.retry && fits) { st->n_root = 0xffffffffu; atomicAdd(&a.prof[5], 1ull); } else { st->err |= ERR_SEED_CAP; } }
			continue;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if(lane == 0) {
			st->seed_n = nlid; st->n_root = ncid;
			if(ncid) {
				if(a.seq_circ) { circularize(gs, c, seed_n, nlid, ncid, a.seq_len, a.seq_circ, a.twlen); }
				if(!radix_sort_64((U64R *)c, ncid, (uint32_t *)cnt, 1536)) { st->err |= ERR_STACK; }              /* longest first (minialign.c:3719); LDS tables reused as scratch */
				uint32_t pred = gaba::NIL, n_pass = 0, w_pass = 0;          /* chains that pass the length test of mm_search_load_root and their summed lengths: what the extension will cost */
				for(uint32_t kq = 0; kq < ncid; kq++) {
					uint32_t pl = (uint32_t)OFS((int32_t)c[kq].plen);
					if(pl * a.mcoef < 2.0 * a.min_score) { break; }
					pred = gs[gs[c[kq].lid].upos].rid; n_pass++; w_pass += pl;
				}
				st->pred_rid = pred; st->n_pass = n_pass; st->w_pass = w_pass;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	if(lane == 0) {
		atomicAdd(&a.prof[0], cy_sort); atomicAdd(&a.prof[1], cy_chain); atomicAdd(&a.prof[2], (unsigned long long)(__builtin_amdgcn_s_memtime() - cy_begin));
		atomicAdd(&a.prof[3], n_big);
	}
}
