/* K1: minimizer sketch, index lookup, seed expansion (mm_sketch, mm_idx_get, mm_collect_seed, mm_expand) -- part of mm_device.hpp (included from there, inside namespace mm; split out in round 6 so that each stage can be read on its own) */
/* =====================================================================================================
 * K1: sketch + lookup + expand
 * ===================================================================================================== */
__device__ __forceinline__ uint32_t crc32c_u64(uint32_t crc, uint64_t v)        /* _mm_crc32_u64; only reached for k > 16 */
{
	for(int i = 0; i < 64; i++) { uint32_t b = (crc ^ (uint32_t)(v >> i)) & 1u; crc = (crc >> 1) ^ (b ? 0x82f63b78u : 0u); }
	return crc;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d)
{
	return ((uint64_t)(uint32_t)__shfl_up((int)(v >> 32), d) << 32) | (uint32_t)__shfl_up((int)v, d);
}

struct K1Args {
	DevIndex idx; gaba::SeqArena qar;
	const ReadIn *in; ReadState *st; uint32_t n_reads;
	MinRec *min_pool;
	Seed *seed_pool; uint64_t seed_pool_cap; unsigned long long *seed_top;
	Resc *resc_pool; uint64_t resc_pool_cap; unsigned long long *resc_top;
	Root *root_pool; uint64_t root_pool_cap; unsigned long long *root_top;
	uint32_t *counter;
	unsigned long long *stats;     /* [0] minimizers probed, [1] seeds */
	const uint32_t *work;          /* read indices to process (n_reads entries) */
	uint64_t *tap;                 /* stage tap (tests): when set, the stream word of every minimizer (hash << 8 | strand << 7 | position mod w, minialign.c:2402) beside its record */
	/* room for the minimizer records of the few reads that emit more than their share (a read inside a satellite array or a homopolymer run emits one per position where
	 * the typical read emits 2 / (w + 1) per base): a region behind the reads' own in min_pool, handed out by a cursor; the records are scratch of this kernel, so a read that
	 * overflows its share takes min(qlen, ...) records there and runs its first pass again (NULL: no such region, the read reports ERR_SEED_CAP as before) */
	unsigned long long *min_over_top; uint64_t min_over_base, min_over_cap;
	unsigned long long *note;      /* pinned HOST memory (or NULL): the last wave of the launch leaves the three pool cursors there -- what the reads of the launch asked for -- so that the host
	                                * has them when the launch is over without a copy of its own (a 24-byte D2H is a blit kernel that waits for a wave slot beside the extension waves: 17 ms per batch) */
};

/* code (0..3, 4 = N) of base p of the read */
__device__ __forceinline__ uint32_t q_code(const gaba::SeqArena &ar, uint64_t p)
{
	uint32_t c = (ar.pk[p >> 4] >> (2 * (p & 15))) & 3;
	uint32_t n = (ar.nm[p >> 5] >> (p & 31)) & 1;
	return n ? 4 : c;
}

/* h of position p of a sequence at `off` in a packed arena (~0 where no k-mer ends): hash << 8 | (k-mer start mod w) | strand << 7 (minialign.c:2394-2402) */
__device__ __forceinline__ uint64_t sketch_h(const gaba::SeqArena &ar, uint64_t q_off, uint32_t p, uint32_t qlen, uint32_t k, uint32_t w, uint64_t kmask)
{
	uint64_t h = ~0ull;
	if(p >= k - 1 && p < qlen) {
		/* forward / reverse k-mers ending at p.  N is pushed as 4 (minialign.c:2391-2392): it ORs into the neighbouring
		 * 2-bit slots, and k1 is never masked, so the recurrence is replayed over the k + 1 bases that can still
		 * influence the registers at p (one base before the window leaves one bit behind in k1) */
		uint64_t k0 = 0, k1 = 0;
		uint32_t start = p >= k ? p - k : 0;
		/* without an N among those k + 1 bases the two registers are plain functions of the k packed bases ending at p:
		 * k1 = their complement in array order, k0 = the same 2-bit groups in reverse order -- two word loads instead of
		 * replaying the recurrence base by base (the replay stays for windows that contain an N, and for k > 16) */
		bool plain = false;
		if(k <= 16) {
			const uint64_t nb = q_off + start, nn = (uint64_t)(p - start + 1);                    /* N bits of bases [start, p] */
			const uint64_t nw = (uint64_t)ar.nm[nb >> 5] | ((uint64_t)ar.nm[(nb >> 5) + 1] << 32);
			plain = ((nw >> (nb & 31)) & ((1ull << nn) - 1)) == 0;
		}
		if(plain) {
			const uint64_t fb = q_off + p - (k - 1);
			const uint64_t ww = (uint64_t)ar.pk[fb >> 4] | ((uint64_t)ar.pk[(fb >> 4) + 1] << 32);
			const uint64_t W = (ww >> (2 * (fb & 15))) & kmask;
			k1 = ~W & kmask;
			uint64_t rv = __brevll(W) >> (64 - 2 * k);                                            /* bit i -> bit 2k - 1 - i */
			k0 = ((rv >> 1) & 0x5555555555555555ull) | ((rv & 0x5555555555555555ull) << 1);      /* ... and the two bits of each base back in order */
		} else {
			for(uint32_t j = start; j <= p; j++) {
				uint64_t c = q_code(ar, q_off + j);
				k0 = (k0 << 2 | c) & kmask;
				k1 = (k1 >> 2) | ((3ull ^ c) << (2 * (k - 1)));
			}
		}
		uint64_t km = k0 < k1 ? k0 : k1, kx = k0 < k1 ? k1 : k0, m = k0 < k1 ? 0 : 0x80;
		/* hash64 (minialign.c:2353): a CRC32C seeded with the low word of its own input is zero unless the high word is set */
		uint64_t crc = (kx >> 32) ? (uint64_t)crc32c_u64((uint32_t)kx, kx) : 0ull;
		uint64_t hv = (crc ^ km) & kmask;
		uint32_t i = (p - (k - 1)) % w;
		h = hv << 8 | i | m;
	}
	return h;
}
/* minimum of h over the last w positions: lane i holds position base + i of the current 64, h_prev the same lanes of the 64 before */
__device__ __forceinline__ uint64_t sketch_window_min(uint64_t h, uint64_t h_prev, uint32_t w, int lane)
{
	/* window minimum over the last w positions (forward-min of the current block + backward-min of the previous one,
	 * minialign.c:2394-2421, is the minimum over [p - w + 1, p]) */
	uint64_t v = h;
	for(uint32_t j = 1; j < w; j++) {
		int src_lane = lane - (int)j;
		uint64_t from_cur = ((uint64_t)(uint32_t)__shfl((int)(h >> 32), src_lane & 63) << 32) | (uint32_t)__shfl((int)h, src_lane & 63);
		uint64_t from_prev = ((uint64_t)(uint32_t)__shfl((int)(h_prev >> 32), src_lane & 63) << 32) | (uint32_t)__shfl((int)h_prev, src_lane & 63);
		uint64_t src = src_lane >= 0 ? from_cur : from_prev;
		v = src < v ? src : v;
	}
	return v;
}

__global__ void __launch_bounds__(256, MM_SHORT_KERNEL_WAVES) mm_sketch_seed_kernel(K1Args a)
{
	__builtin_amdgcn_s_setprio(2);          /* short and latency bound beside the extension waves of the other lanes (which run at 0 or 1, the few heaviest reads of a launch at 3) */
	const int lane = lane_id();
	const DevIndex &ix = a.idx;
	const uint32_t k = ix.k, w = ix.w;
	const uint64_t kmask = (1ull << 2 * k) - 1;
	const uint32_t max_occ = ix.occ[ix.n_occ - 1], resc_occ = ix.occ[0];
	unsigned long long n_probe = 0, n_seedtot = 0;
	while(true) {
		uint32_t r = 0;
		if(lane == 0) { r = atomicAdd(a.counter, 1u); }
		r = (uint32_t)rdfirst((int)r);
		if(r >= a.n_reads) { break; }
		r = (uint32_t)rdfirst((int)a.work[r]);
		ReadState *st = &a.st[r];
		const uint64_t q_off = rdfirst64(a.in[r].q_off);
		const uint32_t qlen = (uint32_t)rdfirst((int)a.in[r].qlen);
		MinRec *rec = a.min_pool + rdfirst64(st->min_off);
		uint32_t min_cap = (uint32_t)rdfirst((int)st->min_cap);
		uint32_t n_rec = 0;            /* uniform */
		uint32_t n_seed = 0, n_resc = 0, n_resc_hits = 0;
		bool in_share = true;          /* the records are in the read's own share of the pool (the stage tap is parallel to that) */

		pass1_again:
		n_rec = 0;
		/* pass 1: minimizers in order, probe the index, keep (qs, n, ref) records */
		uint64_t h_prev = ~0ull;       /* h of the previous 64 positions (lane i = position base - 64 + i) */
		uint64_t v_last = 0;           /* v of the last position of the previous chunk: u of the reference, initial cap value 0 (minialign.c:2412) */
		for(uint32_t base = 0; base < qlen; base += 64) {
			uint32_t p = base + (uint32_t)lane;
			const uint64_t h = sketch_h(a.qar, q_off, p, qlen, k, w, kmask);
			const uint64_t v = sketch_window_min(h, h_prev, w, lane);
			uint64_t vp = shfl_up64(v, 1);
			uint64_t v63 = ((uint64_t)(uint32_t)rdlane((int)(v >> 32), 63) << 32) | (uint32_t)rdlane((int)v, 63);
			if(lane == 0) { vp = v_last; }
			if(p == k - 1) { vp = 0; }                       /* u of the first evaluated position is the initial cap value 0 (minialign.c:2412) */
			bool valid = p >= k - 1 && p < qlen;
			bool emit = valid && ((v == h) || (v != vp));
			/* last valid lane's v feeds the next chunk */
			v_last = v63;
			h_prev = h;
			uint64_t em = __ballot(emit);
			uint32_t my = (uint32_t)__popcll(em & ((1ull << lane) - 1));
			if(emit) {
				uint32_t iv = (uint32_t)(v & 0x7f), ip = (p - (k - 1)) % w;
				uint32_t qpos = (p - (k - 1)) - ((ip + w - iv) % w);          /* = base + u of the reference's decoder (minialign.c:3471-3475) */
				uint64_t fr = (v >> 7) & 1, hh = v >> 8;
				/* mm_idx_get: probe */
				uint64_t s = idx_hash(hh) & ix.mask; uint32_t n = 0; uint64_t ref = 0;
				while(true) {
					IdxSlot sl = ix.slot[s];
					if(sl.key == 0) { break; }
					if(sl.key == hh + 1) { if((int64_t)sl.val >= 0) { n = 1; ref = sl.val; } else { n = (uint32_t)(sl.val & 0xffffff); ref = sl.val; } break; }
					s = (s + 1) & ix.mask;
				}
				uint32_t pos = (uint32_t)((qpos + (k & (uint32_t)-(int32_t)fr)) ^ (uint32_t)-(int32_t)fr);   /* minialign.c:3482 */
				uint32_t slot_i = n_rec + my;
				if(slot_i < min_cap) { rec[slot_i] = MinRec{ pos, n > max_occ ? 0u : n, ref }; if(a.tap && in_share) { a.tap[(uint64_t)(rec - a.min_pool) + slot_i] = hh << 8 | fr << 7 | (uint64_t)(qpos % w); } }
			}
			n_rec += (uint32_t)__popcll(em);
			n_probe += (unsigned long long)__popcll(em);
		}
		if(n_rec > min_cap) {
			if(in_share && a.min_over_top != nullptr) {
				/* more minimizers than the read's share holds: room for one per position from the overflow region, and the pass again */
				const uint64_t need = ((uint64_t)qlen + 63u) & ~63ull; unsigned long long off = 0;
				if(lane == 0) { off = atomicAdd(a.min_over_top, (unsigned long long)need); }
				off = rdfirst64(off);
				if(off + need <= a.min_over_cap) { rec = a.min_pool + a.min_over_base + off; min_cap = (uint32_t)need; in_share = false; goto pass1_again; }
			}
			n_rec = min_cap; if(lane == 0) { st->err |= ERR_SEED_CAP; }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		/* totals */
		for(uint32_t i = (uint32_t)lane; i < n_rec + 63 - ((n_rec + 63) % 64); i += 64) {
			uint32_t n = i < n_rec ? rec[i].n : 0;
			uint32_t s_ = (n != 0 && n <= resc_occ) ? n : 0, rr = n > resc_occ ? 1u : 0u, rh = n > resc_occ ? n : 0;
			for(int o = 32; o > 0; o >>= 1) { s_ += (uint32_t)__shfl_xor((int)s_, o); rr += (uint32_t)__shfl_xor((int)rr, o); rh += (uint32_t)__shfl_xor((int)rh, o); }
			n_seed += s_; n_resc += rr; n_resc_hits += rh;
		}
		n_seed = (uint32_t)rdfirst((int)n_seed); n_resc = (uint32_t)rdfirst((int)n_resc); n_resc_hits = (uint32_t)rdfirst((int)n_resc_hits);
		/* claim space: seeds + sentinel + leaves, doubled as the reference reserves (minialign.c:3709) */
		uint32_t seed_cap = 2 * (n_seed + n_resc_hits + 2);
		uint32_t root_cap = n_seed + n_resc_hits + 2;
		unsigned long long so = 0, ro = 0, to = 0;
		if(lane == 0) { so = atomicAdd(a.seed_top, (unsigned long long)seed_cap); ro = atomicAdd(a.resc_top, (unsigned long long)n_resc + 1); to = atomicAdd(a.root_top, (unsigned long long)root_cap); }
		so = rdfirst64(so); ro = rdfirst64(ro); to = rdfirst64(to);
		bool ok = so + seed_cap <= a.seed_pool_cap && ro + n_resc + 1 <= a.resc_pool_cap && to + root_cap <= a.root_pool_cap;
		if(!ok) { if(lane == 0) { st->err |= ERR_SEED_CAP; st->done = 1; st->seed_n = 0; st->seed_n0 = 0; st->n_seed = 0; st->n_resc = 0; } continue; }
		Seed *seed = a.seed_pool + so; Resc *resc = a.resc_pool + ro;
		/* pass 2: expand in order (mm_expand, minialign.c:3420-3447) */
		uint32_t sp = 0, rp = 0;
		for(uint32_t base = 0; base < n_rec; base += 64) {
			uint32_t i = base + (uint32_t)lane;
			MinRec m = i < n_rec ? rec[i] : MinRec{ 0, 0, 0 };
			uint32_t ns = (m.n != 0 && m.n <= resc_occ) ? m.n : 0, nr = m.n > resc_occ ? 1u : 0u;
			/* exclusive prefix sums across the wave */
			uint32_t ps = ns, pr = nr;
			for(int o = 1; o < 64; o <<= 1) { uint32_t x = (uint32_t)__shfl_up((int)ps, o), y = (uint32_t)__shfl_up((int)pr, o); if(lane >= o) { ps += x; pr += y; } }
			uint32_t tot_s = (uint32_t)rdlane((int)ps, 63), tot_r = (uint32_t)rdlane((int)pr, 63);
			ps -= ns; pr -= nr;
			if(nr) { resc[rp + pr] = Resc{ m.qs, m.n, m.ref }; }
			for(uint32_t j = 0; j < ns; j++) {
				uint64_t hit = (int64_t)m.ref >= 0 ? m.ref : ix.val[((m.ref & 0x7fffffffffffffffull) >> 24) + j];
				uint32_t rid = (uint32_t)(hit >> 32), rs = (uint32_t)hit;
				uint32_t rmask = (uint32_t)-(int32_t)(rid & 1);
				int32_t _rs = (int32_t)(rs + (k & rmask)), _qs = (int32_t)(m.qs ^ rmask);
				seed[sp + ps + j] = Seed{ U_(_rs, _qs), rid >> 1, V_(_rs, _qs), 0x7fffffffu };
			}
			sp += tot_s; rp += tot_r;
		}
		n_seedtot += n_seed;
		if(lane == 0) {
			st->n_min = n_rec;
			st->seed_off = so; st->seed_cap = seed_cap; st->seed_n = n_seed; st->seed_n0 = n_seed; st->n_seed = 0;
			st->resc_off = ro; st->n_resc = n_resc; st->presc = 0;
			st->root_off = to; st->root_cap = root_cap; st->n_root = 0; st->n_res = 0;
		}
	}
	if(lane == 0) {
		atomicAdd(&a.stats[0], n_probe); atomicAdd(&a.stats[1], n_seedtot);
		if(a.note) {
			__threadfence();
			const uint32_t prev = atomicAdd(a.counter + 1, 1u);          /* waves that are through (the word behind the work counter; zeroed with it) */
			if(prev + 1 == gridDim.x * (blockDim.x / 64)) { a.note[0] = atomicAdd(a.seed_top, 0ull); a.note[1] = atomicAdd(a.resc_top, 0ull); a.note[2] = atomicAdd(a.root_top, 0ull); }
		}
	}
}
