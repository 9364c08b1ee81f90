/* K3: the extension driver (mm_extend, mm_search_*, the occurrence-threshold rounds, jobs, workspace rings, watchdog ticks) over the DP of gaba_device.hpp -- part of mm_device.hpp (included from there, inside namespace mm; split out in round 6 so that each stage can be read on its own) */
/* =====================================================================================================
 * K3: extension driver, one wavefront per read
 * ===================================================================================================== */
struct KhSlot { uint64_t k, v; };
#define MM_NEXT_STRIDE(_cap) (2ull * (_cap) + MM_NEXT_SCRATCH)          /* per wave: next[cap], the sort's scratch, a copy of next[] for the look-ahead of the retry jobs */
#define MM_NEXT_SCRATCH 1024u          /* u64 words behind each wave's next[] array: 512 bucket words + 512 pending ranges for radix_sort_64 */
struct AlnRec {                /* what the host needs of a gaba_alignment_t (gaba.h:205-220) */
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, dcnt, slen, plen;
	uint32_t seg_off;          /* index into the segment pool */
	uint64_t path_off;         /* word offset into the path pool; two header words {plen, 0x40000000} precede it (gaba.h:217) */
};
/* one class of DP workspaces as a launch sees it: `slabs` = every workspace of the class, numbered; two rings of free numbers per XCD (k3_ring_try / k3_ring_give) --
 * the SHARED ring of the device (ctr / ring, n numbers per XCD: 0 .. 8 n - 1) that the launches of all lanes take from, and the PRIVATE ring of the lane that launches
 * (pctr / pring, pn numbers per XCD: from 8 n on, the lane's own stretch).  A wave only ever WAITS for a number of its own launch's private ring: the waves of another
 * launch sit on another hardware queue, and a queue can be switched out with everything its waves hold (the hang of round 5) */
struct K3Class { uint8_t *slabs; uint64_t bytes; unsigned long long *ctr; uint32_t *ring; uint32_t n; uint32_t qmax; uint32_t pn; uint32_t pad; unsigned long long *pctr; uint32_t *pring; };
struct K3Args {
	DevIndex idx; gaba::Consts gc; const uint8_t *roots; gaba::SeqArena ar_ref, ar_q;
	const ReadIn *in; ReadState *st; const uint32_t *work; uint32_t n_work;
	Seed *seed_pool; Root *root_pool;
	uint8_t *slabs; uint64_t slab_bytes;                 /* DP workspace per wave */
	/* non-NULL: the workspaces are shared by every launch of every lane -- a wave takes a free one when it starts and gives it back when it ends.  One ring of
	 * free workspace numbers per XCD (a wave only ever takes from the ring of the XCD it runs on, HW_REG_XCC_ID): the L2s of different XCDs are not coherent
	 * with each other inside a launch, so a workspace must not wander between them while kernels are running */
	unsigned long long *ring_ctr; uint32_t *ring; uint32_t ring_n;      /* per XCD x: ring_ctr[2x] = takes, [2x + 1] = returns; ring[x * ring_n ..] = numbers (~0 = taken) */
	/* workspace classes (ring mode; table in device memory, n_cls >= 1, class 0 = the fields above): class c serves the reads of up to cls[c].qmax bases, the last one
	 * the longest read of the input.  A long tail of read lengths (ONT) would otherwise size every workspace for the longest read and leave room for a
	 * fraction of the waves; a wave changes class when the read it takes asks for another one */
	const K3Class *cls; uint32_t n_cls;
	KhSlot *kh_pool; uint32_t kh_cap;                    /* per read (work index) */
	unsigned long long *kh_top; uint64_t kh_base, kh_pool_cap;      /* larger tables for the reads with many chains: handed out behind the fixed regions (from kh_base on) */
	uint32_t round;
	uint64_t *next_pool; uint32_t next_cap;              /* per wave: (pdiff, sid) */
	uint64_t *bin_pool; uint64_t bin_pool_cap; unsigned long long *bin_top; uint32_t bin_cap_per_read;
	AlnRec *aln_pool; uint64_t aln_pool_cap; unsigned long long *aln_top; uint32_t aln_cap_per_read;
	gaba::Segment *seg_pool; uint64_t seg_pool_cap; unsigned long long *seg_top;
	uint32_t *path_pool; uint64_t path_pool_cap; unsigned long long *path_top;
	uint32_t tglen; double mcoef; float min_ratio; uint32_t min_score;
	uint32_t *counter; unsigned long long *stats;        /* [2] fills, [3] vectors, [4] blocks, [5] traces, [6] trace steps */
	uint32_t seg_beg[8], seg_len[8]; uint32_t *seg_cnt;  /* the work list by workspace class: reads of class c at work[seg_beg[c] .. + seg_len[c]), cursor seg_cnt[c] (one class: everything in [0]) */
	/* rounds in the kernel: a read left without a result goes straight on to the next occurrence threshold on the wave that holds it (mm_align_seq's loop,
	 * minialign.c:4444-4448) -- rescued minimizers expanded, seeds sorted and chained again in HBM by that wave, then extended -- instead of coming back
	 * through the host for another round of launches */
	uint32_t inkernel_rounds; Resc *resc_pool; uint32_t twlen;
	/* chain-level parallelism inside the heaviest reads of a launch (the first 64th of the work list: dozens of chains each, one of them is the critical path of the
	 * launch): the first trial of every chain of such a read -- downward extension from its root seed, max search, upward extension, traceback: a pure function of
	 * (reference, cp_a, cp_b, strand) -- is a job any wave of the launch takes BEFORE the waves start on the reads; the wave that later walks the read's chains in order,
	 * with the real hash and bins, takes a job's result where the inputs of the trial it is about to run are the job's (agent-scope release / acquire between the two
	 * waves).  Same results by construction.  NULL: no jobs */
	const struct SpecJob *jobs; struct SpecMemo *memo; unsigned long long *job_top;      /* job_top[0] = jobs enumerated (mm_spec_jobs_kernel), [1] = cursor, [2] staged path words, [3] staged segments (= stage_top), [4] memo hits */
	uint64_t job_cap; uint32_t *spath; uint64_t spath_cap; gaba::Segment *sseg; uint64_t sseg_cap;
	/* retry jobs: after a recorded alignment whose chain has length to spare, mm_search_load_next hands out up to eight more seeds of the chain, one per trial, and nearly every one of
	 * those trials is a full downward pass that ends in a maximum already in the hash (a duplicate, thrown away, minialign.c:3969) -- the tail of a launch is a read doing that on one
	 * wave.  The start points and band widths of these trials follow from the next-seed list alone as long as each is a duplicate, so the wave that is about to run the first of them
	 * works the list ahead on a copy, publishes the rest as jobs (rjobs / rstate / rmemo, agent-scope hand-off as for the chain jobs), and waves that have run out of reads take them
	 * (they stay in the launch until the last read is done: reads_done).  The owner takes a result where its inputs are the trial's, runs a job itself where nobody has claimed it, and
	 * works on a later job of its own while one it needs is in another wave's hands.  NULL: none */
	uint32_t rq_helper_mask;             /* one wave in (mask + 1) is a helper for the retry jobs (one in 128 by default): the first wave of one workgroup in (mask + 1) / 4 of every XCD; every helper holds a wave slot the other lanes' launches wait for */
	struct SpecJob *rjobs; struct SpecMemo *rmemo; uint32_t *rstate; uint32_t rq_cap; unsigned int *rq_ctl;      /* rq_ctl[0] = published, [1] = the takers' cursor, [2] = reads done, [3] = results taken, [4] = reads being walked that have published the chains of a round, [5] = the cursor of the waves that take chain jobs between their reads */
	unsigned long long *stage_top;       /* cursors of the staging area (spath / sseg) that the traced jobs of either kind write to: [0] path words, [1] segments */
	uint32_t round_jobs;                 /* n > 0: a read publishes the chains of a round that was chained inside the launch as jobs (rjobs, JOB_FULL) when it has n or more of them (at least 2) */
	uint32_t dyn0_min;                   /* experiment (MM_K3_DYN_ROUND0 = n, off = 0): a read with n or more passing chains in the round the launch starts with that got no chain jobs before the launch publishes them itself when its wave takes it */
	/* the watchdog's window into the launch (pinned host memory the device writes to while the kernel runs; NULL: none): wd[0] != 0 = the host has called the launch off --
	 * every wave that is waiting for something leaves, its read marked ERR_ABORT; wd[K3_WD_HEAD + wave] = where that wave is (K3_WD_* << 28 | detail), written when a read
	 * is taken and from inside every wait that lasts (k3_wd_tick).  No wait of the kernel is without this way out */
	uint32_t *wd; uint32_t wd_n;
	uint32_t test_hang;                  /* test hook (MM_TEST_K3_HANG): the wave that takes entry test_hang - 1 of the work list waits for something that never comes */
};
enum : uint32_t { K3_WD_HEAD = 16,
	K3_WD_TAKE = 1,          /* looking for a DP workspace: none on offer on its XCD (detail: the class, for a wave that holds a read; bit 24 | the cursor of the work list for one that holds none) */
	K3_WD_GIVE = 2,          /* giving a workspace back: the slot of its give ticket still holds the number of the turn before */
	K3_WD_TRY = 3,           /* the take without waiting: the number of its ticket is on its way into the slot */
	K3_WD_LDS = 4,           /* the tables of the rescue round (one set per workgroup) */
	K3_WD_CARRY = 5,         /* the carried value of the read in front (detail: that read) */
	K3_WD_MEMO = 6,          /* a chain job enumerated before the launch that another wave is running (detail: memo index) */
	K3_WD_CJOB = 7,          /* a chain job published inside the launch that another wave has claimed (detail: slot) */
	K3_WD_RJOB = 8,          /* a retry job another wave has claimed (detail: slot) */
	K3_WD_IDLE = 9,          /* a wave without reads looking for published jobs (detail: reads done) */
	K3_WD_TEST = 10,         /* the test hook */
	K3_WD_JOB = 13,          /* running a job (detail: slot) */
	K3_WD_RAN = 14,          /* back at work after a wait that lasted */
	K3_WD_READ = 15 };       /* took a read (detail: its place in the work list) */
/* called by the polling lane from inside a wait loop: every 1 024th turn it says where the wave is and looks whether the host has called the launch off (true) */
__device__ __forceinline__ bool k3_wd_tick(uint32_t *w, uint32_t wave, uint32_t &st, uint32_t site, uint32_t detail)
{
	st++;
	if((st & 0x3ffu) != 0u || w == nullptr) { return false; }
	__hip_atomic_store(&w[K3_WD_HEAD + wave], (site << 28) | (detail & 0x0fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	st |= 0x40000000u;
	return __hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}
__device__ __forceinline__ void k3_wd_mark(uint32_t *w, uint32_t wave, uint32_t site, uint32_t detail) { if(w != nullptr) { __hip_atomic_store(&w[K3_WD_HEAD + wave], (site << 28) | (detail & 0x0fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } }
__device__ __forceinline__ void k3_wd_ran(uint32_t *w, uint32_t wave, uint32_t &st) { if(st & 0x40000000u) { k3_wd_mark(w, wave, K3_WD_RAN, 0); } st = 0; }

/* the per-read position hash, kh_t (minialign.c:341-683), literal */
struct Kh { KhSlot *a; uint32_t mask, cnt, ub, cap; };
__device__ inline void kh_clear(Kh &h) { h.mask = 255; h.cnt = 0; h.ub = (uint32_t)(256 * 0.4); for(int i = 0; i < 256; i++) { h.a[i].k = ~0ull; h.a[i].v = ~0ull; } }
__device__ inline uint64_t kh_allocate(KhSlot *a, uint64_t k, uint64_t v, uint64_t mask, uint32_t *is_new)
{
	#define KH_POLL(_i, _b0, _k1) { long long _b = (long long)(_b0); while(true) { (_k1) = a[_i].k; \
		if(_b <= (long long)((_k1) & mask) + (long long)((_k1) + 2 < 2)) { break; } _b -= (long long)(((_i) + 1) & (mask + 1)); (_i) = ((_i) + 1) & mask; } }
	uint64_t i = k & mask, k0 = k, v0 = v, k1;
	KH_POLL(i, i, k1);
	if(k0 == k1) { *is_new = 0; return i; }
	uint64_t j = i;
	a[i].k = k0;
	while(k1 + 2 >= 2) {
		uint64_t v1 = a[i].v; a[i].v = v0; k0 = k1; v0 = v1;
		i = (i + 1) & mask;
		KH_POLL(i, k0 & mask, k1);
		a[i].k = k0;
	}
	a[i].v = v0;
	*is_new = 1;
	return j;
	#undef KH_POLL
}
__device__ inline bool kh_extend(Kh &h)
{
	uint64_t prev = (uint64_t)h.mask + 1, size = 2 * prev, mask = size - 1;
	if(size > h.cap) { return false; }
	h.mask = (uint32_t)mask; h.ub = (uint32_t)(size * 0.4);
	for(uint64_t i = 0; i < prev; i++) { h.a[i + prev].k = ~0ull; h.a[i + prev].v = ~0ull; }
	for(uint64_t i = 0; i < size; i++) {
		uint64_t k = h.a[i].k;
		if(k + 2 < 2 || (k & mask) == i) { continue; }
		uint64_t v = h.a[i].v;
		h.a[i].k = ~0ull - 1; h.a[i].v = ~0ull;
		uint32_t dummy; kh_allocate(h.a, k, v, mask, &dummy);
	}
	return true;
}
/* kh_put_ptr: returns the slot index whose value word the caller reads / writes */
__device__ inline uint64_t kh_put(Kh &h, uint64_t key, bool extend, uint32_t *err)
{
	if(extend && h.cnt >= h.ub) { if(!kh_extend(h)) { *err |= ERR_KH_CAP; } }
	/* the table cannot grow any further in its slot of the pool: the read is given up here (the host enlarges the slots and runs the batch again);
	 * inserting on would fill the table and the probe loop would never find a free slot */
	if((*err & ERR_KH_CAP) || h.cnt + 2 >= h.mask) { *err |= ERR_KH_CAP; return 0; }
	uint32_t nw; uint64_t idx = kh_allocate(h.a, key, ~0ull, h.mask, &nw);
	h.cnt += nw;
	return idx;
}
__device__ __forceinline__ uint64_t mm_key(uint64_t x, uint64_t y) { return x ^ (x >> 29) ^ y ^ __builtin_bswap64(y); }    /* minialign.c:3362 */

struct Search {                 /* mm_search_t, minialign.c:3218 */
	uint32_t cp_a, cp_b, tp_a, tp_b;
	uint32_t aid, bid, iid, eid, sid, rev;
	int64_t prem; uint32_t pacc, crem, srem, narrow, min_score;
};
constexpr uint32_t MM_CREM = 50000, MM_SREM = 8;
struct SpecJob { uint32_t r, aid, cp_a, cp_b, rev, rlen, rcirc, pad; };          /* pad: band width class of the trial (sr.narrow: 0 .. 2) | JOB_FULL */
constexpr uint32_t JOB_FULL = 0x100u;          /* the whole first trial of a chain (downward pass, max search, upward pass, traceback into the staging area); without it: downward pass + max search only (a retry trial) */
struct SpecMemo {
	uint32_t state;                  /* 0: not done yet; bit 31: done, bit 0: downward pass + max search valid, bit 1: upward pass (+ traceback when mmax1 >= min_score) valid */
	uint32_t aid, cp_a, cp_b, rev;   /* the inputs it was computed for */
	uint32_t pp_apos, pp_bpos, bw; uint64_t pp_plen; int64_t mmax0;
	int64_t mmax1; uint64_t tplen; uint64_t path_off; uint32_t seg_off;
	gaba::AlnOut ao;
};

/*
 * The DP phases run as real (non-inlined) device functions from the extension driver: the driver keeps ~150 scalars of
 * state (search state, four section descriptors, pool pointers), and letting them stay live across the DP loops makes the
 * compiler spill SGPRs into VGPR lanes *inside* those loops.  Across a call they are saved once.  Arguments and results go
 * by value; uniform values are re-scalarised on entry (arguments travel in VGPRs).
 */
struct DpIn {                 /* what a DP phase needs of the wave's context */
	gaba::Consts c; gaba::SeqArena ar0, ar1; uint8_t *slab; uint32_t top, cap;
};
struct DpOut { uint32_t top; int err; uint32_t n_vec, n_blk, n_tr; };
__device__ __forceinline__ void dp_ctx_open(gaba::Ctx &x, gaba::SeqArena *ar, const DpIn &in)
{
	const uint32_t *src = (const uint32_t *)&in.c; uint32_t *dst = (uint32_t *)&x.c;
	for(uint32_t i = 0; i < sizeof(gaba::Consts) / 4; i++) { dst[i] = (uint32_t)rdfirst((int)src[i]); }
	ar[0].pk = (const uint32_t *)rdfirst64((uint64_t)in.ar0.pk); ar[0].nm = (const uint32_t *)rdfirst64((uint64_t)in.ar0.nm);
	ar[1].pk = (const uint32_t *)rdfirst64((uint64_t)in.ar1.pk); ar[1].nm = (const uint32_t *)rdfirst64((uint64_t)in.ar1.nm);
	x.ar = ar; x.slab = (uint8_t *)rdfirst64((uint64_t)in.slab); x.top = (uint32_t)rdfirst((int)in.top); x.cap = (uint32_t)rdfirst((int)in.cap);
	x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
}
__device__ __forceinline__ gaba::Sec sec_uniform(const gaba::Sec &s)
{
	gaba::Sec r; r.id = (uint32_t)rdfirst((int)s.id); r.len = (uint32_t)rdfirst((int)s.len); r.off = rdfirst64(s.off);
	r.arena = (uint32_t)rdfirst((int)s.arena); r.rev = (uint32_t)rdfirst((int)s.rev); return r;
}
struct ExtOut { DpOut d; uint32_t m; int64_t mmax; uint32_t n_fill; };
__device__ __attribute__((noinline)) ExtOut k3_extend_core(DpIn in, int bw, gaba::Sec ca, uint32_t apos, gaba::Sec cb, uint32_t bpos, int no_trace, int circ)
{
	gaba::Ctx x; gaba::SeqArena ar[2]; dp_ctx_open(x, ar, in);
	x.no_trace = rdfirst(no_trace) != 0;
	const gaba::Sec tailsec = { 0xfffffffeu, 96, 0, 2, 0 };
	ExtOut o; o.n_fill = 0;
	const gaba::Sec cau = sec_uniform(ca);
	o.m = gaba::extend_core(x, rdfirst(bw), cau, (uint32_t)rdfirst((int)apos), sec_uniform(cb), (uint32_t)rdfirst((int)bpos), rdfirst(circ) ? cau : tailsec, tailsec, o.mmax, o.n_fill);
	o.d = DpOut{ x.top, x.err, x.n_vec, x.n_blk, x.n_tr };
	return o;
}
struct LeafOut { DpOut d; gaba::Leaf lf; uint64_t plen; gaba::PosPair pp; };
__device__ __attribute__((noinline)) LeafOut k3_leaf_search(DpIn in, uint32_t tail, int want_pos)
{
	gaba::Ctx x; gaba::SeqArena ar[2]; dp_ctx_open(x, ar, in);
	LeafOut o;
	tail = (uint32_t)rdfirst((int)tail);
	int64_t fbpos = (int64_t)rdfirst64(gaba::tail_at(x, tail)->f.bpos);
	o.plen = (!want_pos && fbpos < gaba::INIT_FETCH_POS) ? 0 : gaba::leaf_search(x, tail, o.lf);
	if(want_pos) { o.pp = gaba::search_max_walk(x, tail, o.lf, o.plen); }
	o.d = DpOut{ x.top, x.err, x.n_vec, x.n_blk, x.n_tr };
	return o;
}
struct TraceOut { DpOut d; gaba::AlnOut ao; };
__device__ __attribute__((noinline)) TraceOut k3_trace(DpIn in, uint32_t tail, gaba::Leaf lf, uint64_t plen, uint32_t *path, gaba::Segment *seg)
{
	gaba::Ctx x; gaba::SeqArena ar[2]; dp_ctx_open(x, ar, in);
	TraceOut o;
	uint32_t *lfw = (uint32_t *)&lf; for(uint32_t i = 0; i < sizeof(gaba::Leaf) / 4; i++) { lfw[i] = (uint32_t)rdfirst((int)lfw[i]); }
	o.ao = gaba::dp_trace_finish(x, (uint32_t)rdfirst((int)tail), lf, rdfirst64(plen), (uint32_t *)rdfirst64((uint64_t)path), (gaba::Segment *)rdfirst64((uint64_t)seg), 8);
	o.d = DpOut{ x.top, x.err, x.n_vec, x.n_blk, x.n_tr };
	return o;
}

/*
 * One job: a trial of a chain as a pure function of its inputs (reference, cp_a, cp_b, strand, band width; minialign.c:4134-4166 up to the duplicate test, and with
 * JOB_FULL on through the upward pass and the traceback, whose path words and segments go to a staging area).  Run by whichever wave of the launch takes the job --
 * the chain jobs enumerated before the launch (K3Args.jobs), the chains a read finds in a later occurrence-threshold round and the retry trials behind a recorded
 * alignment (K3Args.rjobs) -- on the workspace that wave holds (flushed by the caller).  The result goes out with plain stores, an agent-scope release, the drain the
 * compiler may drop, then the flag (MI355X_MICROARCH.md, inter-workgroup visibility: the wave that takes it may sit on another XCD): flag_in_memo -> the memo's own
 * state word (bit 31 | valid bits; it reads 0 until then), else *flag = flag_val with the valid bits in the memo.
 */
struct JobOut { DpOut d; uint32_t n_fill, n_trace; };
__device__ __attribute__((noinline)) JobOut k3_run_job(DpIn din, SpecJob j, uint32_t qlen, uint64_t q_off, uint64_t roff, uint32_t min_score,
	SpecMemo *mo_out, uint32_t *flag, uint32_t flag_val, int flag_in_memo, uint32_t *spath, uint64_t spath_cap, gaba::Segment *sseg, uint64_t sseg_cap, unsigned long long *stage_top)
{
	const int lane = lane_id();
	const uint32_t aid = (uint32_t)rdfirst((int)j.aid), cp_a = (uint32_t)rdfirst((int)j.cp_a), cp_b = (uint32_t)rdfirst((int)j.cp_b);
	const uint32_t rev = (uint32_t)rdfirst((int)j.rev), rlen = (uint32_t)rdfirst((int)j.rlen), kind = (uint32_t)rdfirst((int)j.pad); const int rcirc = rdfirst((int)j.rcirc);
	const int bw = (int)(kind & 0xffu); const bool full = (kind & JOB_FULL) != 0;
	qlen = (uint32_t)rdfirst((int)qlen); q_off = rdfirst64(q_off); roff = rdfirst64(roff); min_score = (uint32_t)rdfirst((int)min_score);
	mo_out = (SpecMemo *)rdfirst64((uint64_t)mo_out); flag = (uint32_t *)rdfirst64((uint64_t)flag); flag_val = (uint32_t)rdfirst((int)flag_val); flag_in_memo = rdfirst(flag_in_memo);
	spath = (uint32_t *)rdfirst64((uint64_t)spath); spath_cap = rdfirst64(spath_cap); sseg = (gaba::Segment *)rdfirst64((uint64_t)sseg); sseg_cap = rdfirst64(sseg_cap);
	stage_top = (unsigned long long *)rdfirst64((uint64_t)stage_top);
	const gaba::Sec rsec_f = gaba::Sec{ aid << 1, rlen, roff, 0, 0 }, rsec_r = gaba::Sec{ (aid << 1) + 1, rlen, roff, 0, 1 };
	const gaba::Sec qsec_f = gaba::Sec{ 0, qlen, q_off, 1, 0 }, qsec_r = gaba::Sec{ 1, qlen, q_off, 1, 1 };
	SpecMemo mo; mo.state = 0; mo.aid = aid; mo.cp_a = cp_a; mo.cp_b = cp_b; mo.rev = rev; mo.bw = (uint32_t)bw; mo.mmax0 = 0; mo.mmax1 = 0; mo.tplen = 0; mo.path_off = 0; mo.seg_off = 0;
	mo.pp_apos = mo.pp_bpos = 0; mo.pp_plen = 0;
	mo.ao.status = 0; mo.ao.score = 0; mo.ao.identity = 0; mo.ao.agcnt = mo.ao.bgcnt = mo.ao.dcnt = mo.ao.slen = mo.ao.plen = 0;
	JobOut o; o.n_fill = 0; o.n_trace = 0; o.d.n_vec = 0; o.d.n_blk = 0; o.d.n_tr = 0;
	ExtOut eo = k3_extend_core(din, bw, rsec_f, cp_a, rev ? qsec_r : qsec_f, cp_b, 1, rcirc);
	uint32_t top = (uint32_t)rdfirst((int)eo.d.top); int err = rdfirst(eo.d.err);
	o.d.n_vec += (uint32_t)rdfirst((int)eo.d.n_vec); o.d.n_blk += (uint32_t)rdfirst((int)eo.d.n_blk); o.n_fill += (uint32_t)rdfirst((int)eo.n_fill);
	uint32_t m = (uint32_t)rdfirst((int)eo.m); int64_t mmax = (int64_t)rdfirst64((uint64_t)eo.mmax);
	bool go = err == 0;
	if(go) { mo.mmax0 = mmax; mo.state = 1; if(mmax == 0) { go = false; } }
	if(go) {
		din.top = top;
		LeafOut lo = k3_leaf_search(din, m, 1);
		mo.pp_apos = (uint32_t)rdfirst((int)lo.pp.apos); mo.pp_bpos = (uint32_t)rdfirst((int)lo.pp.bpos); mo.pp_plen = rdfirst64(lo.pp.plen);
	}
	if(go && full) {
		const uint32_t tp_a = (uint32_t)max(1, min((int32_t)mo.pp_apos, (int32_t)rlen)), tp_b = (uint32_t)max(1, min((int32_t)mo.pp_bpos, (int32_t)qlen));
		din.top = top;
		ExtOut e1 = k3_extend_core(din, bw, rsec_r, rlen - tp_a, rev ? qsec_f : qsec_r, qlen - tp_b, 0, rcirc);
		top = (uint32_t)rdfirst((int)e1.d.top); err = rdfirst(e1.d.err);
		o.d.n_vec += (uint32_t)rdfirst((int)e1.d.n_vec); o.d.n_blk += (uint32_t)rdfirst((int)e1.d.n_blk); o.n_fill += (uint32_t)rdfirst((int)e1.n_fill);
		m = (uint32_t)rdfirst((int)e1.m); mmax = (int64_t)rdfirst64((uint64_t)e1.mmax);
		if(err == 0) {
			mo.mmax1 = mmax;
			if(mmax < (int64_t)min_score) { mo.state |= 2; }
			else {
				din.top = top;
				LeafOut l1 = k3_leaf_search(din, m, 0);
				const uint64_t tplen = rdfirst64(l1.plen);
				const uint64_t need_words = (tplen + 31) / 32 + 2;
				unsigned long long po = 0, so_ = 0;
				if(lane == 0) { po = atomicAdd(&stage_top[0], (unsigned long long)need_words); so_ = atomicAdd(&stage_top[1], 8ull); }
				po = rdfirst64(po); so_ = rdfirst64(so_);
				if(po + need_words <= spath_cap && so_ + 8 <= sseg_cap) {
					din.top = top;
					TraceOut to = k3_trace(din, m, l1.lf, tplen, spath + po, sseg + so_);
					gaba::AlnOut ao = to.ao;
					ao.status = rdfirst(ao.status); ao.plen = (uint32_t)rdfirst((int)ao.plen); ao.slen = (uint32_t)rdfirst((int)ao.slen);
					if(rdfirst(to.d.err) == 0) { mo.ao = ao; mo.tplen = tplen; mo.path_off = po; mo.seg_off = (uint32_t)so_; o.d.n_tr += (uint32_t)rdfirst((int)to.d.n_tr); o.n_trace++; mo.state |= 2; }
				}
			}
		}
	}
	const uint32_t bits = mo.state;
	if(flag_in_memo) { mo.state = 0; flag = &mo_out->state; flag_val = bits | 0x80000000u; }
	if(lane == 0) { *mo_out = mo; }
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if(lane == 0) { __hip_atomic_store(flag, flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	o.d.top = top; o.d.err = 0;
	return o;
}

/*
 * mm_seed for iteration >= 1 + mm_chain (minialign.c:3509-3535, 3702-3725) for one read, by the wavefront that holds it: the rescue list is sorted once
 * (key qs | n << 32, the same unstable radix sort), the minimizers whose occurrence count the new threshold admits are expanded behind the seeds, the whole
 * array is sorted and chained again in place in HBM (sort_chain_wave, the form the largest reads take in K2a), chains circularised, roots sorted.  tab:
 * 1536 words of LDS of this wave (bucket tables and range stack of the sort, scratch of the root sort).
 */
__device__ __attribute__((noinline)) uint32_t k3_rescue_round(ReadState *st, uint32_t round, Seed *gs, Root *c, Resc *resc, DevIndex ix, uint32_t twlen, double mcoef, uint32_t min_score, LU32 *tab)
{
	const int lane = lane_id();
	LU32 *cnt = tab, *bb = tab + 256, *be = tab + 512, *stack = tab + 768;
	unsigned long long cs = 0, cc = 0; uint32_t nlid = 0, ncid = 0; uint32_t err = 0;
	K2aArgs ka; ka.twlen = twlen;
	const uint32_t n_resc = (uint32_t)rdfirst((int)st->n_resc);
	if(round == 1 && n_resc > 1) {
		/* n_resc elements, none of them a sentinel: the sort takes "seed_n + 1" elements as they are */
		if(!sort_chain_wave<Seed>((Seed *)resc, n_resc, n_resc - 1, cnt, bb, be, stack, c, ka, lane, nlid, ncid, cs, cc, false, nullptr, nullptr, false, false)) { err |= ERR_STACK; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	uint32_t seed_n = (uint32_t)rdfirst((int)st->n_seed);
	const uint32_t half = (uint32_t)rdfirst((int)st->seed_cap) / 2;
	for(uint32_t i = (uint32_t)lane; i < seed_n; i += 64) { gs[i].lid = 0x7fffffffu; }
	uint32_t p = (uint32_t)rdfirst((int)st->presc);
	const uint32_t occ = ix.occ[round];
	while(p < n_resc) {
		const uint32_t qs = (uint32_t)rdfirst((int)resc[p].qs), mn = (uint32_t)rdfirst((int)resc[p].n); const uint64_t ref = rdfirst64(resc[p].ref);
		if(mn > occ) { break; }
		for(uint32_t j0 = 0; j0 < mn; j0 += 64) {
			const uint32_t j = j0 + (uint32_t)lane;
			if(j < mn) {
				const uint64_t hit = (int64_t)ref >= 0 ? ref : ix.val[((ref & 0x7fffffffffffffffull) >> 24) + j];
				const uint32_t rid = (uint32_t)(hit >> 32), rs = (uint32_t)hit;
				const uint32_t rmask = (uint32_t)-(int32_t)(rid & 1);
				const int32_t _rs = (int32_t)(rs + (ix.k & rmask)), _qs = (int32_t)(qs ^ rmask);
				/* a hit that finds no room is dropped and flagged, the ones behind it move up (minialign.c: the reference reserves; here the host redoes the batch) */
				if(seed_n + j + 2 < half) { gs[seed_n + j] = Seed{ U_(_rs, _qs), rid >> 1, V_(_rs, _qs), 0x7fffffffu }; }
			}
		}
		if(seed_n + mn + 1 < half) { seed_n += mn; } else { err |= ERR_SEED_CAP; seed_n = seed_n + 2 < half ? half - 2 : seed_n; }
		p++;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if(lane == 0) { st->presc = p; st->n_seed = seed_n; st->n_root = 0; st->pred_rid = gaba::NIL; }
	if(seed_n == 0 || (err & ERR_SEED_CAP)) { if(lane == 0) { st->seed_n = 0; } return err; }
	if(lane == 0) { gs[seed_n] = Seed{ 0x80000000u, 0x7fffffffu, 0x80000000u, 0x7fffffffu }; }      /* sentinel, minialign.c:3531 */
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if(!sort_chain_wave<Seed>(gs, 2 * half, seed_n, cnt, bb, be, stack, c, ka, lane, nlid, ncid, cs, cc, false, nullptr, nullptr, false, true)) { err |= ERR_SEED_CAP; return err; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if(lane == 0) {
		st->seed_n = nlid; st->n_root = ncid;
		if(ncid) {
			if(ix.seq_circ) { circularize(gs, c, seed_n, nlid, ncid, ix.seq_len, ix.seq_circ, twlen); }
			if(!radix_sort_64((U64R *)c, ncid, (uint32_t *)tab, 1536)) { err |= ERR_STACK; }
			uint32_t pred = gaba::NIL, n_pass = 0, w_pass = 0;          /* chains that pass the length test of mm_search_load_root and their summed lengths: what the extension will cost */
			for(uint32_t kq = 0; kq < ncid; kq++) {
				uint32_t pl = (uint32_t)OFS((int32_t)c[kq].plen);
				if(pl * mcoef < 2.0 * min_score) { break; }
				pred = gs[gs[c[kq].lid].upos].rid; n_pass++; w_pass += pl;
			}
			st->pred_rid = pred; st->n_pass = n_pass; st->w_pass = w_pass;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	return (uint32_t)rdfirst((int)err);
}

#ifndef MM_K3_WAVES_PER_SIMD
#define MM_K3_WAVES_PER_SIMD 8
#endif
/* per-phase timing of the extension kernel (s_memtime around every fill / search / traceback, per-read ticks): compiled in with -DMM_K3_PROF only
 * (__graft_entry__.build() makes libminialign_amd_prof.so that way; bench.py / tools take it through MM_LIB_OVERRIDE); the production kernel reads the clock
 * twice per wave, for the load-balance figure */
#ifdef MM_K3_PROF
#define MM_TICK() __builtin_amdgcn_s_memtime()
#else
#define MM_TICK() 0ull
#endif
#ifndef MM_K3_LAUNCH_BOUND
#define MM_K3_LAUNCH_BOUND MM_K3_WAVES_PER_SIMD          /* waves per SIMD the register budget of the kernel is set for */
#endif
#define K3_TAB_WORDS 1536u
#define K3_LDS_BYTES ((K3_TAB_WORDS + 16u) * 4u)          /* dynamic LDS of a launch with the rounds in the kernel: the tables of k3_rescue_round + their lock */
/* one thread per heavy read (the first n_heavy entries of the work list): the chains mm_extend will visit -- root order, up to the length test of
 * mm_search_load_root (minialign.c:3849) -- with the positions mm_search_load_pos gives their root seeds; the `apos >= rlen` test sees the length of the
 * reference the chain in front loaded (minialign.c:3864).  Reads with fewer than min_roots such chains are left alone. */
struct SpecJobsArgs { DevIndex idx; const ReadIn *in; ReadState *st; const uint32_t *work; uint32_t n_heavy; const Seed *seed_pool; const Root *root_pool;
	double mcoef; uint32_t min_score, min_roots; SpecJob *jobs; SpecMemo *memo; uint64_t job_cap; unsigned long long *job_top; };
__global__ void __launch_bounds__(64) mm_spec_jobs_kernel(SpecJobsArgs a)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if(t >= a.n_heavy) { return; }
	const uint32_t r = a.work[t];
	ReadState *st = &a.st[r];
	st->spec_n = 0; st->spec_off = 0;
	const uint32_t n_root = st->n_root;
	if(n_root < a.min_roots || n_root >= 0xfffffffeu || st->err) { return; }
	const DevIndex &ix = a.idx;
	const Seed *s = a.seed_pool + st->seed_off; const Root *root = a.root_pool + st->root_off;
	const uint32_t qlen = a.in[r].qlen;
	uint32_t cnt = 0;
	for(uint32_t kq = 0; kq < n_root; kq++) { const uint32_t plen = (uint32_t)OFS((int32_t)root[kq].plen); if(plen * a.mcoef < 2.0 * a.min_score) { break; } cnt++; }
	if(cnt < a.min_roots) { return; }
	const unsigned long long off = atomicAdd(&a.job_top[0], (unsigned long long)cnt);
	if(off + cnt > a.job_cap) {
		/* no room for this read's jobs: it keeps spec_n = 0 and runs its trials itself.  The count stays above the capacity and the extension kernel clamps it, so the
		 * slots this read drew below the capacity are claimed there all the same: they are marked empty (r = ~0) -- left unwritten they would hold whatever an earlier
		 * launch put there */
		for(unsigned long long q = off; q < a.job_cap; q++) { a.jobs[q] = SpecJob{ 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u, 0u }; a.memo[q].state = 0x80000000u; }
		return;
	}
	uint32_t rlen = st->rlen;
	for(uint32_t kq = 0; kq < cnt; kq++) {
		const uint32_t lid = root[kq].lid, rsid = s[lid].upos; const Seed p = s[rsid];
		const int32_t bs = BS(p); const uint32_t rev = bs < 0;
		uint32_t cpa = (uint32_t)AS(p), cpb = (uint32_t)(bs + ((bs >> 31) & (int32_t)qlen));
		if(cpa >= rlen || cpb >= qlen) { cpa -= min(cpa, ix.k); cpb -= min(cpb, ix.k); }
		rlen = ix.seq_len[p.rid];
		a.jobs[off + kq] = SpecJob{ r, p.rid, cpa, cpb, rev, rlen, ix.seq_circ ? (uint32_t)ix.seq_circ[p.rid] : 0u, 0u };
		a.memo[off + kq].state = 0;
	}
	st->spec_off = (uint32_t)off; st->spec_n = cnt;
}

/* Room of a read in the result-bin pool, the alignment pool and the position-hash pool, by the chains the round at hand will walk (st->n_pass: from the chaining of
 * this round -- K2 for the first, k3_rescue_round for the later ones): a chain costs a bin header (two words), every alignment it records a bin word, an alignment
 * record and two position-hash entries.  The typical read walks one or two chains; a read inside a repeat family finds its hundreds of chains only in the later rounds
 * (the repeat's minimizers pass the second or third occurrence threshold), and with one cap for all reads those few made the whole batch run again with 4 x, 16 x, 64 x
 * the pools (the hard-repeat set: 250 - 400 chains, 300 alignments, 770 bin words on reads whose first round had three chains).  A read that needs more than it
 * holds takes a new, larger region from the pool and carries over what it had -- the bins of dropped chains and the alignments they recorded stay readable at their
 * old indices, as in the reference's vectors (a later alignment that ends where one of them did reads them, minialign.c:4018-4067), and the hash table keeps its
 * layout (it grows in place, up to the size of its region).  Called by the whole wave at the start of every round of a read; the state goes through *st.
 * Returns 0, or the error bit of the pool that is used up (the host then runs the batch again with larger pools). */
__device__ __attribute__((noinline)) uint32_t k3_room(ReadState *st, uint32_t round, uint32_t r, uint64_t *bin_pool, uint64_t bin_pool_cap, unsigned long long *bin_top, uint32_t bin_def,
	AlnRec *aln_pool, uint64_t aln_pool_cap, unsigned long long *aln_top, uint32_t aln_def, KhSlot *kh_pool, uint64_t kh_pool_cap, unsigned long long *kh_top, uint64_t kh_base, uint32_t kh_def)
{
	const int lane = lane_id();
	const uint32_t np = (uint32_t)rdfirst((int)st->n_pass);
	const uint64_t bin_off = rdfirst64(st->bin_off), aln_off = rdfirst64(st->aln_off);
	const uint32_t bin_cap = (uint32_t)rdfirst((int)st->bin_cap), aln_cap = (uint32_t)rdfirst((int)st->aln_cap), n_aln = (uint32_t)rdfirst((int)st->n_aln);
	const uint32_t want_bin = min(1u << 24, max(bin_def, 5u * np + 64u)), want_aln = min(1u << 22, max(aln_def, 3u * np + 32u));
	const bool first = bin_off == ~0ull;
	if(first || want_bin > bin_cap || want_aln > aln_cap) {
		/* (a read that moves takes at least twice what it held: the regions it leaves behind are not reclaimed, so the moves of a read are bounded by a logarithm) */
		const uint32_t nb = first ? want_bin : max(want_bin, 2u * bin_cap), na = first ? want_aln : max(want_aln, 2u * aln_cap);
		uint32_t bo_lo = 0, bo_hi = 0, ao_lo = 0, ao_hi = 0;
		if(lane == 0) { const unsigned long long b = atomicAdd(bin_top, (unsigned long long)nb), q = atomicAdd(aln_top, (unsigned long long)na); bo_lo = (uint32_t)b; bo_hi = (uint32_t)(b >> 32); ao_lo = (uint32_t)q; ao_hi = (uint32_t)(q >> 32); }
		const uint64_t bo = (uint64_t)(uint32_t)rdfirst((int)bo_lo) | ((uint64_t)(uint32_t)rdfirst((int)bo_hi) << 32), ao = (uint64_t)(uint32_t)rdfirst((int)ao_lo) | ((uint64_t)(uint32_t)rdfirst((int)ao_hi) << 32);
		/* no room in the pools: the read is given up for this pass; it must not touch another read's region */
		if(bo + nb > bin_pool_cap) { return ERR_BIN_CAP; }
		if(ao + na > aln_pool_cap) { return ERR_ALN_CAP; }
		if(!first) {
			const uint32_t *ob = (const uint32_t *)(bin_pool + bin_off); uint32_t *nbp = (uint32_t *)(bin_pool + bo);
			for(uint32_t i = (uint32_t)lane; i < 2u * bin_cap; i += 64) { nbp[i] = ob[i]; }
			const uint32_t *oa = (const uint32_t *)(aln_pool + aln_off); uint32_t *nap = (uint32_t *)(aln_pool + ao);
			for(uint32_t i = (uint32_t)lane; i < n_aln * (uint32_t)(sizeof(AlnRec) / 4); i += 64) { nap[i] = oa[i]; }
		}
		if(lane == 0) { st->bin_off = bo; st->aln_off = ao; st->bin_cap = nb; st->aln_cap = na; if(first) { st->n_bin = 0; st->n_aln = 0; } }          /* (first round of this read: mm_tbuf_clear, minialign.c:4402) */
	}
	/* the position hash: two entries per recorded alignment at a load of 0.4 */
	uint32_t kcap = (uint32_t)rdfirst((int)st->kh_cap); uint64_t koff = rdfirst64(st->kh_off);
	if(kcap == 0) { kcap = kh_def; koff = (uint64_t)r * kh_def; if(lane == 0) { st->kh_off = koff; st->kh_cap = kh_def; } }          /* the read's ordinary region */
	uint32_t want_kh = kh_def; while(want_kh < 32u * np && want_kh < (1u << 22)) { want_kh <<= 1; }
	if(want_kh > kcap && kh_top != nullptr) {
		uint32_t ko_lo = 0, ko_hi = 0;
		if(lane == 0) { const unsigned long long k = atomicAdd(kh_top, (unsigned long long)want_kh) + kh_base; ko_lo = (uint32_t)k; ko_hi = (uint32_t)(k >> 32); }
		const uint64_t ko = (uint64_t)(uint32_t)rdfirst((int)ko_lo) | ((uint64_t)(uint32_t)rdfirst((int)ko_hi) << 32);
		if(ko + want_kh > kh_pool_cap) { return ERR_KH_CAP; }
		if(round != 0) {
			const uint32_t mask = (uint32_t)rdfirst((int)st->kh_mask);
			const uint32_t *ok_ = (const uint32_t *)(kh_pool + koff); uint32_t *nk_ = (uint32_t *)(kh_pool + ko);
			for(uint32_t i = (uint32_t)lane; i < 4u * (mask + 1u); i += 64) { nk_[i] = ok_[i]; }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if(lane == 0) { st->kh_off = ko; st->kh_cap = want_kh; }
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	return 0;
}
/*
 * A ring of free DP workspace numbers (K3Class: a shared one per class, a private one per class and lane), one per XCD: ring[x * n ..] = numbers (~0 = taken),
 * ctr[4 x + 0] = take tickets drawn, [4 x + 1] = give tickets drawn (the ring starts with its n numbers given), [4 x + 2] = numbers on offer.  Taking never waits for
 * a workspace: a number is promised first (the counter of numbers on offer, a semaphore) and the ticket drawn only then, so that the one wait left is the short one for
 * the number of that ticket to land in its slot (its giver has drawn the give ticket and is about to store).  A wave that finds nothing on offer goes on without, or
 * looks again later (mm_extend_kernel: acquire) -- it holds no ticket and no place in any line, and can leave whenever it likes.  The L2s of different XCDs are not
 * coherent inside a launch, so a workspace never wanders between them: a wave takes from and gives to the rings of the XCD it runs on.
 */
__device__ __forceinline__ uint32_t k3_ring_try(unsigned long long *ctr, uint32_t *ring, uint32_t n, uint32_t xcc, int lane, uint32_t *wdw, uint32_t wave, uint32_t &wst)
{
	uint32_t v = 0xffffffffu;
	if(lane == 0 && n != 0u) {
		unsigned long long *c = ctr + 4u * xcc;
		if((long long)atomicAdd(&c[2], ~0ull) <= 0ll) { atomicAdd(&c[2], 1ull); }          /* nothing on offer (the promise is handed back) */
		else {
			const unsigned long long t = atomicAdd(&c[0], 1ull); uint32_t *slot = &ring[(uint64_t)xcc * n + (uint32_t)(t % n)];
			while((v = atomicExch(slot, 0xffffffffu)) == 0xffffffffu) {          /* (the number is on its way into the slot) */
				__builtin_amdgcn_s_sleep(2);
				if(k3_wd_tick(wdw, wave, wst, K3_WD_TRY, (uint32_t)t & 0xffffffu)) { break; }
			}
			k3_wd_ran(wdw, wave, wst);
		}
	}
	return (uint32_t)rdfirst((int)v);
}
__device__ __forceinline__ void k3_ring_give(unsigned long long *ctr, uint32_t *ring, uint32_t n, uint32_t xcc, uint32_t no, int lane, uint32_t *wdw, uint32_t wave, uint32_t &wst)
{
	if(lane == 0) {
		unsigned long long *c = ctr + 4u * xcc;
		const unsigned long long t = atomicAdd(&c[1], 1ull); uint32_t *slot = &ring[(uint64_t)xcc * n + (uint32_t)(t % n)];
		while(atomicCAS(slot, 0xffffffffu, no) != 0xffffffffu) {          /* (the taker of this slot's previous turn has not picked its number up yet) */
			__builtin_amdgcn_s_sleep(2);
			if(k3_wd_tick(wdw, wave, wst, K3_WD_GIVE, (uint32_t)t & 0xffffffu)) { break; }
		}
		k3_wd_ran(wdw, wave, wst);
		atomicAdd(&c[2], 1ull);
	}
}
#ifdef MM_K3_NUM_VGPR
__attribute__((amdgpu_num_vgpr(MM_K3_NUM_VGPR)))
#endif
__global__ void __launch_bounds__(256, MM_K3_LAUNCH_BOUND) mm_extend_kernel(K3Args a)
{
	extern __shared__ uint32_t k3_tab[];       /* launched with 4 x 1536 words when the rounds run in the kernel (per wave: tables of k3_rescue_round's sort + chain), else with none:
	                                            * a static array would make the compiler trade the 8 waves per SIMD of the launch bounds for registers */
	gaba::SeqArena ar[2] = { a.ar_ref, a.ar_q };
	gaba::Ctx x;
	x.c = a.gc; x.ar = ar; x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
	const int lane = x.lane;
	uint32_t wave = (uint32_t)rdfirst((int)(blockIdx.x * 4 + threadIdx.x / 64));
	uint32_t slab_no = wave; uint32_t xcc = 0; int slab_cls = -1;          /* class of the workspace held: -1 none yet (ring mode) */
	if(a.ring) { xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }          /* HW_REG_XCC_ID, bits 3:0 */
	else {
		x.slab = a.slabs + (uint64_t)slab_no * a.slab_bytes; x.cap = (uint32_t)a.slab_bytes; x.top = gaba::SLAB_HEAD; slab_cls = 0;
		for(uint32_t i = (uint32_t)lane; i < gaba::SLAB_HEAD / 4; i += 64) { ((uint32_t *)x.slab)[i] = ((const uint32_t *)a.roots)[i]; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	/* the watchdog's window (K3Args.wd): where this wave is, and the way out of every wait */
	uint32_t *const wdw = (a.wd != nullptr && wave < a.wd_n) ? a.wd : nullptr; uint32_t wst = 0;
	ReadState *cur_st = nullptr;          /* the read this wave holds (marked ERR_ABORT when the wave leaves a wait because the launch was called off) */
	#define K3_LEAVE() { if(lane == 0) { if(cur_st != nullptr) { cur_st->err |= ERR_ABORT; } k3_wd_mark(wdw, wave, 0, 2); } return; }
	Kh kh; kh.cap = a.kh_cap;
	/* with shared workspaces a wave maps ONE read and ends (grid = reads / 4): wave slots then come free read by read, and the launches of the other lanes --
	 * sketch, sort, chain, copies, the next extension launch -- get theirs within a read's time instead of waiting for a whole persistent launch to drain;
	 * the per-wave scratch is numbered like the workspace.  Without the ring (per-call entries): persistent waves stealing reads from a counter, as before. */
	uint64_t *next = a.next_pool + (uint64_t)wave * MM_NEXT_STRIDE(a.next_cap);      /* [next_cap entries][radix-sort scratch]; one-read-per-wave launches: re-pointed below by workspace number */
	uint32_t *next_scratch = (uint32_t *)(next + a.next_cap);
	const DevIndex &ix = a.idx;
	unsigned long long n_fill = 0, n_trace = 0;
	unsigned long long cy_fill = 0, cy_leaf = 0, cy_trace = 0;        /* wave cycles spent in the three DP phases (s_memtime) */
	unsigned long long cy_next = 0;                                  /* ... and in mm_search_load_next */
	const unsigned long long cy_begin = __builtin_amdgcn_s_memtime();
	if(a.inkernel_rounds) { if(threadIdx.x == 0) { k3_tab[K3_TAB_WORDS] = 0; } __syncthreads(); }          /* the lock of the tables */

	/* workspace `no` of class c is this wave's from here on */
	auto bind_slab = [&](int c, uint32_t no) {
		slab_no = no; slab_cls = c;
		x.slab = a.cls[c].slabs + (uint64_t)no * a.cls[c].bytes; x.cap = (uint32_t)a.cls[c].bytes; x.top = gaba::SLAB_HEAD;
		for(uint32_t i = (uint32_t)lane; i < gaba::SLAB_HEAD / 4; i += 64) { ((uint32_t *)x.slab)[i] = ((const uint32_t *)a.roots)[i]; }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	};
	auto class_of = [&](uint32_t qlen) -> int { int want = 0; while(want + 1 < (int)a.n_cls && qlen > a.cls[want].qmax) { want++; } return want; };
	/* the workspace this wave holds goes back to the ring it came from (numbers below 8 n: the shared one) */
	auto give_slab = [&]() {
		if(slab_cls < 0) { return; }
		const K3Class &k = a.cls[slab_cls];
		if(slab_no < 8u * k.n) { k3_ring_give(k.ctr, k.ring, k.n, xcc, slab_no, lane, wdw, wave, wst); }
		else { k3_ring_give(k.pctr, k.pring, k.pn, xcc, slab_no, lane, wdw, wave, wst); }
		slab_cls = -1;
	};
	/* A workspace of class `want`: one of the launch's own if there is one on offer, else one of the shared ones.  must = false: if there is none right now the wave goes on
	 * with what it holds (a wave without a read, or about to take somebody else's work: it looks again later or does without -- never a line to stand in).  must = true: the
	 * wave holds a read that needs the class; what it holds goes back first and it looks again until there is one -- the launch's own ring has at least one workspace of
	 * every class per XCD, held by waves of this very launch, which are on this hardware queue and give theirs back when they change class or run out of reads.
	 * false with called_off set: the watchdog ended the wait */
	bool called_off = false;
	auto acquire = [&](int want, bool must) -> bool {
		if(want == slab_cls) { return true; }
		if(must) { give_slab(); }
		const K3Class &k = a.cls[want];
		for(;;) {
			uint32_t no = k3_ring_try(k.pctr, k.pring, k.pn, xcc, lane, wdw, wave, wst);
			if(no == 0xffffffffu) { no = k3_ring_try(k.ctr, k.ring, k.n, xcc, lane, wdw, wave, wst); }
			if(no != 0xffffffffu) { give_slab(); bind_slab(want, no); return true; }
			if(!must) { return false; }
			__builtin_amdgcn_s_sleep(32);
			uint32_t off = 0; if(lane == 0) { off = k3_wd_tick(wdw, wave, wst, K3_WD_TAKE, (uint32_t)want) ? 1u : 0u; }
			if(rdfirst((int)off)) { called_off = true; return false; }
		}
	};
	auto need_slab = [&](uint32_t qlen) -> bool { return acquire(class_of(qlen), true); };
	auto try_slab = [&](int want) -> bool { return acquire(want, false); };
	/* a job on the workspace this wave holds (the caller has made sure of its class): counters of the DP work go to this wave */
	auto run_job = [&](const SpecJob &j, SpecMemo *mo_out, uint32_t *flag, uint32_t flag_val, int flag_in_memo) {
		const uint32_t jr = (uint32_t)rdfirst((int)j.r), ja = (uint32_t)rdfirst((int)j.aid);
		gaba::dp_flush(x); x.err = 0;
		if(lane == 0) { k3_wd_mark(wdw, wave, K3_WD_JOB, jr); }
		DpIn din; din.c = x.c; din.ar0 = ar[0]; din.ar1 = ar[1]; din.slab = x.slab; din.top = x.top; din.cap = x.cap;
		const unsigned long long cyj0 = MM_TICK();
		JobOut jo = k3_run_job(din, j, a.in[jr].qlen, a.in[jr].q_off, a.idx.seq_off[ja], a.min_score, mo_out, flag, flag_val, flag_in_memo, a.spath, a.spath_cap, a.sseg, a.sseg_cap, a.stage_top);
		x.n_vec += (uint32_t)rdfirst((int)jo.d.n_vec); x.n_blk += (uint32_t)rdfirst((int)jo.d.n_blk); x.n_tr += (uint32_t)rdfirst((int)jo.d.n_tr);
		n_fill += (uint32_t)rdfirst((int)jo.n_fill); n_trace += (uint32_t)rdfirst((int)jo.n_trace);
		cy_fill += MM_TICK() - cyj0;
		gaba::dp_flush(x); x.err = 0;
		if(lane == 0) { k3_wd_mark(wdw, wave, K3_WD_RAN, 1); }
	};
	/* jobs first: the first trials of the chains of the heaviest reads, one per wave at a time, by every wave of the launch (K3Args.jobs) */
	if(a.jobs && a.ring) {
		const unsigned long long n_jobs = min(rdfirst64(a.job_top[0]), (unsigned long long)a.job_cap);
		__builtin_amdgcn_s_setprio(3);
		while(n_jobs) {
			unsigned long long ji = 0;
			if(lane == 0) { ji = atomicAdd(&a.job_top[1], 1ull); }
			ji = rdfirst64(ji);
			if(ji >= n_jobs) { break; }
			SpecJob j = a.jobs[ji];
			if((uint32_t)rdfirst((int)j.r) == 0xffffffffu) { continue; }          /* a slot of a read whose jobs did not fit (mm_spec_jobs_kernel) */
			const uint32_t qlen = (uint32_t)rdfirst((int)a.in[(uint32_t)rdfirst((int)j.r)].qlen);
			{
				/* the workspace without waiting: the wave holds a claimed job, and the waves that hold the workspaces of a scarce class may soon be waiting for this very job.
				 * None free: the job is handed back undone (the read's own wave runs the trial when it gets there, as without jobs) */
				const int want = class_of(qlen);
				bool have = want == slab_cls;
				/* (a class with a workspace for every wave an XCD can hold never makes anybody wait: the plain ticket, one atomic add -- the compare-and-swap of the other form,
				 * with a thousand waves of an XCD at the same counter when the launch starts, is what a first version with a bounded number of attempts failed on: nearly every
				 * job of an E.coli-size set was handed back, 182 -> 211 ms per step) */
				if(!have) { have = try_slab(want); }
				if(!have) { if(lane == 0) { __hip_atomic_store(&a.memo[ji].state, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } continue; }
			}
			j.pad = JOB_FULL;
			run_job(j, a.memo + ji, nullptr, 0u, 1);
		}
		__builtin_amdgcn_s_setprio(0);
	}
	/* jobs published inside the launch (K3Args.rjobs; SpecJob.pad says which kind): the retry trials behind a recorded alignment (downward pass + max search) and the
	 * first trials of the chains a read finds in a later occurrence-threshold round (the whole trial), into rmemo[ji]; taken by helper waves, by every wave between two
	 * reads, by waves that have run out of reads while a read with published chains is still being walked, or by the read's own wave ahead of its turn */
	enum : uint32_t { RJ_EMPTY = 0, RJ_READY = 1, RJ_CLAIMED = 2, RJ_DONE = 3, RJ_CANCELLED = 4 };
	const bool rq_on = a.rjobs != nullptr && a.ring != nullptr;
	/* the read's own wave, while a job it needs is in another wave's hands: one of its later jobs (slots [q0, q1)), if one is still unclaimed */
	auto own_job = [&](uint32_t q0, uint32_t q1) -> bool {
		uint32_t take = 0xffffffffu;
		if(lane == 0) { for(uint32_t q = q0; q < q1; q++) { if(atomicCAS(&a.rstate[q], (uint32_t)RJ_READY, (uint32_t)RJ_CLAIMED) == RJ_READY) { take = q; break; } } }
		take = (uint32_t)rdfirst((int)take);
		if(take == 0xffffffffu) { return false; }
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		run_job(a.rjobs[take], a.rmemo + take, a.rstate + take, (uint32_t)RJ_DONE, 0);
		return true;
	};

	/* the helpers: the first wave of one workgroup in (mask + 1) / 4, counted within an XCD (workgroup b runs on XCD b % 8: the workspaces a helper can take are its XCD's) */
	const bool rq_helper = rq_on && (threadIdx.x >> 6) == 0 && ((blockIdx.x >> 3) & (max(a.rq_helper_mask, 3u) >> 2)) == 0;
	bool no_reads = rq_helper;          /* this wave takes no (more) reads: the helpers are helpers from the start of the launch (the reads that publish retry jobs are at the front of the work list) */
	uint32_t rq_mine = 0xffffffffu;                   /* a slot number this wave drew that has not been published yet */
	while(true) {
		if(rq_on) {
			/* published jobs come before the next read: a wave with reads left takes what is there and goes on; one without stays -- a helper until the last read is done,
			 * any other wave while a read that has published the chains of a later round is still being walked (rq_ctl[4]) */
			uint32_t idle = 0;
			while(true) {
				uint32_t ji = rq_mine, stt = 0, fin = 0, wide = 0;
				/* two cursors over the one queue: the waves without reads (helpers among them) take whatever is published; a wave with reads left walks the queue on a cursor of
				 * its own and takes the chain jobs only (JOB_FULL) -- the retry trials stay with the helpers as in round 3: taken between reads they cost the ONT-like set 7 %
				 * (a retry trial of a 100 kb read in front of a wave's own next read).  A slot one cursor steps over is still in front of the other; the claim is by state */
				const uint32_t cix = no_reads ? 1u : 5u;
				if(lane == 0) {
					if(ji == 0xffffffffu) {
						const uint32_t cur = __hip_atomic_load(&a.rq_ctl[cix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), top = __hip_atomic_load(&a.rq_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if(cur < top && cur < a.rq_cap) { ji = atomicAdd(&a.rq_ctl[cix], 1u); if(ji >= a.rq_cap) { ji = 0xffffffffu; } }
					}
					if(ji != 0xffffffffu) { stt = __hip_atomic_load(&a.rstate[ji], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					if(no_reads) { fin = __hip_atomic_load(&a.rq_ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.n_work ? 1u : 0u; wide = __hip_atomic_load(&a.rq_ctl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
				}
				ji = (uint32_t)rdfirst((int)ji); stt = (uint32_t)rdfirst((int)stt); fin = (uint32_t)rdfirst((int)fin); wide = (uint32_t)rdfirst((int)wide);
				if(ji != 0xffffffffu && stt == RJ_READY) {
					/* the workspace the job needs comes BEFORE the claim: a claimed job is one that will be finished, whatever the waves that wait for it hold (with several
					 * workspace classes a wave that claimed first and then waited for a workspace of a scarce class could wait for the very waves that wait for it) */
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
					if(!no_reads && ((uint32_t)rdfirst((int)a.rjobs[ji].pad) & JOB_FULL) == 0u) { rq_mine = 0xffffffffu; idle = 0; continue; }          /* (a retry job: the helpers') */
					const uint32_t jr = (uint32_t)rdfirst((int)a.rjobs[ji].r), jq = (uint32_t)rdfirst((int)a.in[jr].qlen);
					const int want = class_of(jq);
					bool have = want == slab_cls;
					/* (a wave with reads left keeps the workspace it holds: on a ladder of classes it would give a scarce one back for a job of another class and wait for it again for
					 * its next read -- it takes the jobs that fit what it holds, the waves without reads take any) */
					if(!have && (no_reads || slab_cls < 0)) { have = try_slab(want); }
					if(have) { if(lane == 0) { stt = atomicCAS(&a.rstate[ji], (uint32_t)RJ_READY, (uint32_t)RJ_CLAIMED) == RJ_READY ? 100u : 99u; } stt = (uint32_t)rdfirst((int)stt); }
					else { stt = RJ_EMPTY; }          /* no workspace of that class free: the job stays with its owner unless one comes back before the owner gets there */
				}
				if(ji != 0xffffffffu && stt == 100u) {
					/* (at the top priority: the wave that waits for this result is the critical path of the launch) */
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __builtin_amdgcn_s_setprio(3);
					run_job(a.rjobs[ji], a.rmemo + ji, a.rstate + ji, (uint32_t)RJ_DONE, 0);
					__builtin_amdgcn_s_setprio(0); rq_mine = 0xffffffffu; idle = 0;
					/* a workspace of a class above the ordinary one goes back at once: the classes are small, and a wave that sat on one between jobs could be what a read is waiting for */
					if(slab_cls >= 1) { give_slab(); }
					continue;
				}
				if(ji != 0xffffffffu && stt != RJ_EMPTY) { rq_mine = 0xffffffffu; idle = 0; continue; }          /* taken by its owner, done or cancelled: the next one */
				rq_mine = ji;                                                                                 /* drawn but not published yet (or nothing drawn) */
				if(!no_reads) { if(ji == 0xffffffffu || ++idle > 4u) { break; } __builtin_amdgcn_s_sleep(8); continue; }          /* (reads are waiting: on with them) */
				if(fin) { break; }
				/* a wave that is not a helper leaves as soon as nothing is on offer: staying for what the reads still being walked MIGHT publish (the first form: while
				 * rq_ctl[4] != 0) held thousands of wave slots through the tail of every launch -- the waves of the other lanes' launches wait for exactly those slots; on the
				 * ONT-like set, where a launch lasts as long as its longest read, 2.1 against 2.8 G bases/s.  a.rq_stay (MM_K3_STAY): the first form */
				if(!rq_helper) { if(ji == 0xffffffffu || ++idle > 16u) { break; } }
				__builtin_amdgcn_s_sleep(64);
				{ uint32_t off = 0; if(lane == 0) { off = k3_wd_tick(wdw, wave, wst, K3_WD_IDLE, ji) ? 1u : 0u; } if(rdfirst((int)off)) { K3_LEAVE(); } }
			}
		}
		if(no_reads) { break; }
		cur_st = nullptr;
		if(wdw != nullptr) { uint32_t off = 0; if(lane == 0) { off = __hip_atomic_load(&wdw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } if(rdfirst((int)off)) { return; } }          /* (called off: no more reads) */
		uint32_t wi = wave;
		if(a.ring) {
			/* The work list by workspace class (K3Args.seg_*; one class: everything in [0]).  A read of the highest class above the ordinary one that has reads left AND a
			 * workspace at hand (held already, or on offer on this XCD right now); else one of the ordinary class -- the workspace FIRST, then the read: a wave that finds no
			 * workspace on offer holds nothing anybody could wait for, looks again while reads of the class are left, and ends when they are gone (the waves of the launch
			 * that hold its own workspaces work the list off whatever the rest of the device does); when the ordinary class is used up, what is left above it: the read
			 * first, then its workspace, waiting for one of the launch's own as need be */
			wi = 0xffffffffu;
			for(int c = (int)a.n_cls - 1; c >= 1 && wi == 0xffffffffu; c--) {
				const uint32_t len = a.seg_len[c];
				if(len == 0) { continue; }
				uint32_t cur = 0; if(lane == 0) { cur = __hip_atomic_load(&a.seg_cnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } cur = (uint32_t)rdfirst((int)cur);
				if(cur >= len) { continue; }
				if(!try_slab(c)) { continue; }
				uint32_t i = 0; if(lane == 0) { i = atomicAdd(&a.seg_cnt[c], 1u); } i = (uint32_t)rdfirst((int)i);
				if(i < len) { wi = a.seg_beg[c] + i; }
			}
			if(wi == 0xffffffffu && a.seg_len[0] != 0u) {
				bool have0 = slab_cls == 0;
				while(!have0) {
					uint32_t cur = 0; if(lane == 0) { cur = __hip_atomic_load(&a.seg_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } cur = (uint32_t)rdfirst((int)cur);
					if(cur >= a.seg_len[0]) { break; }
					have0 = try_slab(0);
					if(have0) { break; }
					__builtin_amdgcn_s_sleep(64);
					uint32_t off = 0; if(lane == 0) { off = k3_wd_tick(wdw, wave, wst, K3_WD_TAKE, 0x1000000u | (cur & 0xffffffu)) ? 1u : 0u; }
					if(rdfirst((int)off)) { K3_LEAVE(); }
				}
				if(lane == 0) { k3_wd_ran(wdw, wave, wst); }
				if(have0) { uint32_t i = 0; if(lane == 0) { i = atomicAdd(&a.seg_cnt[0], 1u); } i = (uint32_t)rdfirst((int)i); if(i < a.seg_len[0]) { wi = a.seg_beg[0] + i; } }
			}
			for(int c = (int)a.n_cls - 1; c >= 1 && wi == 0xffffffffu; c--) {
				if(a.seg_len[c] == 0) { continue; }
				uint32_t i = 0; if(lane == 0) { i = atomicAdd(&a.seg_cnt[c], 1u); } i = (uint32_t)rdfirst((int)i);
				if(i < a.seg_len[c]) { wi = a.seg_beg[c] + i; }
			}
		}
		else { if(lane == 0) { wi = atomicAdd(a.counter, 1u); } wi = (uint32_t)rdfirst((int)wi); }
		if(wi >= a.n_work) {
			/* no read left for this wave: it stays for the published jobs of the reads that are still being walked (above), or ends */
			if(rq_on) { no_reads = true; continue; }
			break;
		}
		const uint32_t r = (uint32_t)rdfirst((int)a.work[wi]);
		ReadState *st = &a.st[r];
		cur_st = st; if(lane == 0) { k3_wd_mark(wdw, wave, K3_WD_READ, wi); }
		if(a.test_hang != 0u && wi + 1u == a.test_hang) {          /* test hook: this wave waits for something that never comes, until the watchdog calls the launch off */
			uint32_t off = 0; if(lane == 0) { while(!k3_wd_tick(wdw, wave, wst, K3_WD_TEST, wi)) { __builtin_amdgcn_s_sleep(32); if(wdw == nullptr && wst > (1u << 16)) { break; } } off = 1; }
			if(rdfirst((int)off) && wdw != nullptr) { K3_LEAVE(); }
		}
		for(uint32_t round = a.round; ; round++) {
		if(round != a.round) {
			/* the next occurrence threshold for this read, here and now */
			/* ONE set of tables per workgroup, taken in turn by its four waves: the rounds are rare (a few per cent of the reads), and 24 KB of LDS per workgroup held
			 * six workgroups' worth of a CU's LDS for the whole launch -- the sort and chain kernels of the other lanes, which live on LDS, ran 2.3 x slower beside it */
			const unsigned long long cy_resc0 = MM_TICK();
			{
				uint32_t off = 0;
				if(lane == 0) { while(atomicCAS((unsigned int *)&k3_tab[K3_TAB_WORDS], 0u, 1u) != 0u) { __builtin_amdgcn_s_sleep(32); if(k3_wd_tick(wdw, wave, wst, K3_WD_LDS, r)) { off = 1; break; } } k3_wd_ran(wdw, wave, wst); }
				if(rdfirst((int)off)) { K3_LEAVE(); }
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			const uint32_t e2 = k3_rescue_round(st, round, a.seed_pool + rdfirst64(st->seed_off), a.root_pool + rdfirst64(st->root_off), a.resc_pool + rdfirst64(st->resc_off),
				a.idx, a.twlen, a.mcoef, a.min_score, (LU32 *)&k3_tab[0]);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			if(lane == 0) { atomicExch((unsigned int *)&k3_tab[K3_TAB_WORDS], 0u); st->k3_ticks += (uint32_t)(MM_TICK() - cy_resc0); st->k3_wait_ticks += (uint32_t)(MM_TICK() - cy_resc0); }          /* (profiling build: the round's sort + chain counts as time of the read; reported with the workspace wait) */
			if(e2) { if(lane == 0) { st->err |= e2; } break; }
		}
		const unsigned long long cy_read0 = MM_TICK(); const uint32_t vec_read0 = x.n_vec; const unsigned long long cyf_read0 = cy_fill, cyt_read0 = cy_trace;
		const uint32_t n_root = (uint32_t)rdfirst((int)st->n_root); uint32_t dg_trials = 0, dg_hits = 0, dg_chains = 0;
		/* reads with many chains run several extension trials and are the critical path of the launch (one of them can cost
		 * three times a wave's fair share): their waves get issue priority so that they move at uncontended speed while the
		 * ordinary reads fill the slots in between.  The top priority goes by place in the work list -- its first 64th is the reads with the
		 * most chains of the batch (run_rounds puts them there) -- and to nobody else: with every read of 8 chains or more at 3 and of 5 at 2
		 * (the earlier rule: a tenth of the reads) the truly heavy ones had company at their level; 2.32 - 2.36 against 2.51 - 2.54 s per step */
		if(wi < (a.n_work >> 6) || a.n_work < 64) { __builtin_amdgcn_s_setprio(3); }          /* (a launch of a few reads is a re-run for the carried value: its lane, and the lanes behind it, wait for it) */ else if(n_root >= 5) { __builtin_amdgcn_s_setprio(1); } else { __builtin_amdgcn_s_setprio(0); }
		const uint32_t qlen = (uint32_t)rdfirst((int)a.in[r].qlen);
		const uint64_t q_off = rdfirst64(a.in[r].q_off);
		if(a.ring) { if(!need_slab(qlen)) { K3_LEAVE(); } }          /* (false only when the launch was called off; the class the read needs: held already unless the read came from what was left above the ordinary class) */
		const unsigned long long cy_slab = MM_TICK();
		Seed *s = a.seed_pool + rdfirst64(st->seed_off);
		Root *root = a.root_pool + rdfirst64(st->root_off);
		uint32_t rlen = (uint32_t)rdfirst((int)st->rlen);
		if(round == a.round) {
			const uint32_t dep = (uint32_t)rdfirst((int)st->dep);
			if(dep != gaba::NIL) {
				/* the value this read starts with is what read `dep` ends with, and that read is one whose later rounds decide it: taken from the read itself (it stands at the
				 * front of the work list, so a wave has it; the wait is bounded all the same -- past it the read runs with the host's prediction and the host's check decides) */
				uint32_t ok = 0;
				if(lane == 0) {
					const unsigned long long t0 = __builtin_amdgcn_s_memtime();
					while((ok = __hip_atomic_load(&a.st[dep].carry_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) { if(__builtin_amdgcn_s_memtime() - t0 > (1ull << 28)) { break; } __builtin_amdgcn_s_sleep(32); if(k3_wd_tick(wdw, wave, wst, K3_WD_CARRY, dep)) { break; } }
					k3_wd_ran(wdw, wave, wst);
				}
				ok = (uint32_t)rdfirst((int)ok);
				if(ok) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); rlen = (uint32_t)rdfirst((int)__hip_atomic_load(&a.st[dep].rlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
			}
			if(lane == 0) { st->rlen_in = rlen; }
		}
		uint32_t err = 0, n_res = (uint32_t)rdfirst((int)st->n_res);
		uint32_t rid_last = (uint32_t)rdfirst((int)st->rid_last);
		uint32_t apos0 = (uint32_t)rdfirst((int)st->apos0), cond0 = (uint32_t)rdfirst((int)st->cond0);

		/* per-read output regions */
		/* room by the chains this round will walk (k3_room: a read that needs more than it holds moves to a larger region of the pools) */
		{
			const uint32_t e3 = k3_room(st, round, r, a.bin_pool, a.bin_pool_cap, a.bin_top, a.bin_cap_per_read, a.aln_pool, a.aln_pool_cap, a.aln_top, a.aln_cap_per_read, a.kh_pool, a.kh_pool_cap, a.kh_top, a.kh_base, a.kh_cap);
			if(e3) { if(lane == 0) { st->err |= e3; } break; }
		}
		uint64_t bin_off = rdfirst64(st->bin_off), aln_off = rdfirst64(st->aln_off);
		uint32_t n_bin = (uint32_t)rdfirst((int)st->n_bin), n_aln = (uint32_t)rdfirst((int)st->n_aln);
		const uint32_t bin_cap_r = (uint32_t)rdfirst((int)st->bin_cap), aln_cap_r = (uint32_t)rdfirst((int)st->aln_cap);
		uint64_t *bin = a.bin_pool + bin_off;
		AlnRec *alns = a.aln_pool + aln_off;
		/* the hash is cleared once per read (mm_tbuf_clear, minialign.c:4402) and shared by the rounds of that read */
		kh.a = a.kh_pool + rdfirst64(st->kh_off); kh.cap = (uint32_t)rdfirst((int)st->kh_cap);
		if(round != 0) { kh.mask = (uint32_t)rdfirst((int)st->kh_mask); kh.cnt = (uint32_t)rdfirst((int)st->kh_cnt); kh.ub = (uint32_t)rdfirst((int)st->kh_ub); }
		if(round == 0) { if(lane == 0) { kh_clear(kh); } }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		x.err = 0;

		Search sr;
		sr.crem = MM_CREM; sr.min_score = a.min_score; sr.narrow = 0; sr.srem = 0; sr.prem = 0; sr.pacc = 0;
		sr.cp_a = sr.cp_b = sr.tp_a = sr.tp_b = 0; sr.aid = sr.bid = sr.iid = sr.eid = sr.sid = sr.rev = 0;
		uint32_t next_n = 0;
		uint32_t rj_base = 0, rj_n = 0, rj_i = 0;          /* retry jobs published for the trials that follow (K3Args.rjobs): first slot, count, next to be used */
		auto cancel_rjobs = [&]() { if(rj_i < rj_n && lane == 0) { for(uint32_t q = rj_i; q < rj_n; q++) { (void)atomicCAS(&a.rstate[rj_base + q], (uint32_t)RJ_READY, (uint32_t)RJ_CANCELLED); } } rj_n = rj_i = 0; };
		const uint32_t spec_n = (a.jobs != nullptr && round == a.round) ? (uint32_t)rdfirst((int)st->spec_n) : 0u, spec_off = (uint32_t)rdfirst((int)st->spec_off);          /* (chain jobs are enumerated for the round a launch starts with) */
		gaba::Sec rsec_f, rsec_r, qsec_f, qsec_r; int rcirc = 0;
		qsec_f = gaba::Sec{ 0, qlen, q_off, 1, 0 }; qsec_r = gaba::Sec{ 1, qlen, q_off, 1, 1 };

		#define LOAD_POS(_p, _cpa, _cpb, _rev) { \
			int32_t _bs = BS(_p); (_rev) = _bs < 0; \
			(_cpa) = (uint32_t)AS(_p); (_cpb) = (uint32_t)(_bs + ((_bs >> 31) & (int32_t)qlen)); \
			if(first_pos) { apos0 = (_cpa); cond0 = (_cpb) >= qlen; first_pos = false; } \
			if((_cpa) >= rlen || (_cpb) >= qlen) { (_cpa) -= min((_cpa), ix.k); (_cpb) -= min((_cpb), ix.k); } }
		bool first_pos = apos0 == gaba::NIL;

		/* The chains of a round that was chained INSIDE the launch (k3_rescue_round above) become jobs here: a read inside a repeat family finds its hundreds of chains only
		 * when the second or third occurrence threshold admits the family's minimizers, nearly every one of them a full-length alignment that is recorded, and walked them one
		 * after the other on this one wave -- seconds, while the rest of the launch had long finished (the hard-repeat set: 6 M DP vectors on one wave, 0.05 G bases/s).  The
		 * first trial of a chain is a pure function of (reference, cp_a, cp_b, strand) -- what mm_search_load_root / load_pos will set up, the carried reference length
		 * included (the `apos >= rlen` test sees the length of the reference the chain in front loaded, minialign.c:3864) -- so all of them are published at once (the hand-off of
		 * the retry jobs: slot states, agent-scope release / acquire), any wave takes them, and the walk below, in order and with the real hash and bins, takes the results.
		 * dyn0_min: the same for the chains of the round the launch starts with, for reads that got no chain jobs before the launch. */
		uint32_t cj_base = 0, cj_n = 0;
		if(rq_on && a.round_jobs && (round != a.round || (a.dyn0_min != 0u && spec_n == 0u))) {
			const uint32_t np = (uint32_t)rdfirst((int)st->n_pass);
			if(np >= (round != a.round ? max(2u, a.round_jobs) : a.dyn0_min) && np <= n_root) {
				uint32_t base = 0, ok = 0;
				if(lane == 0) { base = atomicAdd(&a.rq_ctl[0], np); ok = (base + np <= a.rq_cap) ? 1u : 0u; }          /* (a full queue: the slots stay empty, the waves step over them) */
				base = (uint32_t)rdfirst((int)base); ok = (uint32_t)rdfirst((int)ok);
				if(ok) {
					for(uint32_t kj = (uint32_t)lane; kj < np; kj += 64) {
						const Seed p = s[s[root[kj].lid].upos];
						const uint32_t rl = kj ? ix.seq_len[s[s[root[kj - 1].lid].upos].rid] : rlen;
						const int32_t bs = BS(p); const uint32_t jrev = bs < 0;
						uint32_t cpa = (uint32_t)AS(p), cpb = (uint32_t)(bs + ((bs >> 31) & (int32_t)qlen));
						if(cpa >= rl || cpb >= qlen) { cpa -= min(cpa, ix.k); cpb -= min(cpb, ix.k); }
						a.rjobs[base + kj] = SpecJob{ r, p.rid, cpa, cpb, jrev, ix.seq_len[p.rid], ix.seq_circ ? (uint32_t)ix.seq_circ[p.rid] : 0u, JOB_FULL };
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					for(uint32_t kj = (uint32_t)lane; kj < np; kj += 64) { __hip_atomic_store(&a.rstate[base + kj], (uint32_t)RJ_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
					if(lane == 0) { atomicAdd(&a.rq_ctl[4], 1u); }
					cj_base = base; cj_n = np;
				}
			}
		}

		for(uint32_t kq = 0; kq < n_root; kq++) {
			/* mm_search_load_root (minialign.c:3839-3883) */
			Root rt = root[kq];
			uint32_t lid = (uint32_t)rdfirst((int)rt.lid);
			uint32_t plen = (uint32_t)OFS((int32_t)rdfirst((int)rt.plen));
			if(plen * a.mcoef < 2.0 * a.min_score) { break; }
			next_n = 0; dg_chains++;
			if(n_bin + 2 > bin_cap_r) { err |= ERR_BIN_CAP; break; }
			uint32_t iid = n_bin;
			if(lane == 0) { bin[iid] = 0; bin[iid + 1] = 0; }        /* header {n_aln, plen, lb, ub}: all-zero as in the reference *as built*  */
			n_bin += 2;
			uint32_t eid = n_res++;
			if(lane == 0) { root[eid] = Root{ (uint32_t)OFS(0), iid }; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			uint32_t rsid = (uint32_t)rdfirst((int)s[lid].upos);
			Seed ps = s[rsid];
			ps.upos = (uint32_t)rdfirst((int)ps.upos); ps.vpos = (uint32_t)rdfirst((int)ps.vpos); ps.rid = (uint32_t)rdfirst((int)ps.rid);
			LOAD_POS(ps, sr.cp_a, sr.cp_b, sr.rev);
			sr.tp_a = sr.cp_a; sr.tp_b = sr.cp_b;
			sr.aid = ps.rid; sr.bid = 0; sr.iid = iid; sr.eid = eid; sr.sid = rsid;
			sr.prem = plen; sr.pacc = 0; sr.srem = MM_SREM; sr.narrow = 0;
			/* mm_init_ref */
			rlen = (uint32_t)rdfirst((int)ix.seq_len[sr.aid]); rid_last = sr.aid;
			uint64_t roff = rdfirst64(ix.seq_off[sr.aid]);
			rsec_f = gaba::Sec{ sr.aid << 1, rlen, roff, 0, 0 }; rsec_r = gaba::Sec{ (sr.aid << 1) + 1, rlen, roff, 0, 1 };
			rcirc = ix.seq_circ ? rdfirst((int)ix.seq_circ[sr.aid]) : 0;          /* rtp = circular ? r : t (minialign.c:3753) */

			bool first_iter = true, chain_first = true;
			while(true) {
				const unsigned long long cy_n0 = MM_TICK();
				if(!first_iter) {
					/* mm_search_load_next (minialign.c:3888-3946) */
					if(sr.srem == 0) { /* nothing */ }
					else {
						sr.srem--;
						uint64_t ofs = 2ull * a.tglen;
						int32_t fa = (int32_t)sr.cp_a, fb = (int32_t)(sr.cp_b - (sr.rev ? qlen : 0u));
						V4 fv = V4{ (int32_t)U_(fa, fb), (int32_t)sr.aid, (int32_t)V_(fa, fb), (int32_t)V_(fa, fb) };
						uint32_t ncnt = next_n;
						uint64_t plim = ofs - sr.pacc;
						if(sr.pacc > ofs) { ncnt = 0; }
						/* serial section on lane 0 (short arrays) */
						uint32_t sid_out = sr.sid;
						if(lane == 0) {
							for(uint32_t i = 0; i < ncnt; i++) {
								uint32_t pd = (uint32_t)next[i];
								if(pd >= plim) { ncnt = i; break; }
								next[i] = (next[i] & 0xffffffff00000000ull) | (uint32_t)(pd + sr.pacc);
							}
							uint64_t sid = sr.sid;
							for(uint64_t rcnt = 2ull * sr.srem; sid > 0 && rcnt > 0; sid--) {
								V4 pv = load_pv(s[sid - 1]);
								V4 wv = add_win(pv, (int32_t)a.tglen), zv = add_win(pv, 128);
								if(!inside_uub(wv, fv)) { break; }
								if(!inside_wv(wv, fv) || inside_wv(zv, fv)) { continue; }
								if(ncnt < a.next_cap) { next[ncnt++] = (uint64_t)(uint32_t)pdiff(wv, fv) | ((uint64_t)(sid - 1) << 32); } else { err |= ERR_NEXT_CAP; }
								rcnt--;
							}
							sid_out = (uint32_t)sid;
							/* radix_sort_64x (minialign.c:3932): mostly below the 64-element insertion-sort threshold, the radix passes for the rest */
							if(!radix_sort_64((U64R *)next, ncnt, next_scratch, 2 * MM_NEXT_SCRATCH)) { err |= ERR_STACK; }
						}
						__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
						ncnt = (uint32_t)rdfirst((int)ncnt); sr.sid = (uint32_t)rdfirst((int)sid_out); err = (uint32_t)rdfirst((int)err);
						next_n = ncnt;
						if(ncnt == 0) { sr.pacc = 0; sr.srem = 0; }
						else {
							next_n = ncnt - 1;
							uint64_t e = rdfirst64(next[next_n]);
							uint32_t nsid = (uint32_t)(e >> 32);
							sr.pacc = (uint32_t)(ofs - (uint32_t)e);
							Seed ns = s[nsid];
							ns.upos = (uint32_t)rdfirst((int)ns.upos); ns.vpos = (uint32_t)rdfirst((int)ns.vpos);
							LOAD_POS(ns, sr.cp_a, sr.cp_b, sr.rev);
						}
					}
				}
				cy_next += MM_TICK() - cy_n0;
				first_iter = false;
				if(!(sr.srem > 0 && sr.prem > 0)) { break; }

				/* one extension trial (minialign.c:4134-4166): pass 0 = downward extension + max search + duplicate test,
				 * pass 1 = upward extension from the max + max search for the traceback.  One loop so that the DP code is
				 * instantiated once. */
				gaba::dp_flush(x);
				const int bw = (int)sr.narrow;               /* _dp(x) ignores its argument (minialign.c:4123) */
				uint32_t m = gaba::NIL; int64_t mmax = 0; gaba::Leaf tlf; uint64_t tplen = 0;
				bool skip = false;
				/* the first trial of a chain of a heavy read is a job another wave took (or is still working on): waited for, acquired, and taken if it was computed for
				 * exactly the inputs of this trial (it always is unless the walk stopped differently in front) */
				const SpecMemo *smp = a.memo + (spec_off + kq); bool memo0 = false, memo1 = false, memo_trace = false;
				if(chain_first && bw == 0 && kq < spec_n) {
					uint32_t stt = 0;
					if(lane == 0) { while((stt = __hip_atomic_load(&smp->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) { __builtin_amdgcn_s_sleep(32); if(k3_wd_tick(wdw, wave, wst, K3_WD_MEMO, spec_off + kq)) { break; } } k3_wd_ran(wdw, wave, wst); }
					stt = (uint32_t)rdfirst((int)stt);
					if(stt == 0u) { K3_LEAVE(); }          /* (called off while the job was in another wave's hands) */
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
					if((stt & 1u) && (uint32_t)rdfirst((int)smp->aid) == sr.aid && (uint32_t)rdfirst((int)smp->cp_a) == sr.cp_a && (uint32_t)rdfirst((int)smp->cp_b) == sr.cp_b && (uint32_t)rdfirst((int)smp->rev) == (sr.rev ? 1u : 0u)) { memo0 = true; memo1 = (stt & 2u) != 0; }
					if(memo0 && lane == 0) { atomicAdd(&a.job_top[4], 1ull); }
					dg_hits += memo0 ? 1u : 0u;
				}
				if(chain_first && kq < cj_n) {
					/* the first trial of a chain that was published as a job above: taken where it is done, run here where nobody has claimed it, and while another wave is at
					 * it this one works on a later chain of the read */
					const uint32_t ji = cj_base + kq;
					while(true) {
						uint32_t stt = 0;
						if(lane == 0) { stt = __hip_atomic_load(&a.rstate[ji], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if(stt == RJ_READY) { stt = atomicCAS(&a.rstate[ji], (uint32_t)RJ_READY, (uint32_t)RJ_CANCELLED) == RJ_READY ? (uint32_t)RJ_CANCELLED : (uint32_t)RJ_CLAIMED; } }
						stt = (uint32_t)rdfirst((int)stt);
						if(stt == RJ_CANCELLED) { break; }
						if(stt == RJ_DONE) {
							__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
							smp = a.rmemo + ji;
							const uint32_t bits = (uint32_t)rdfirst((int)smp->state);
							memo0 = (bits & 1u) != 0 && (uint32_t)rdfirst((int)smp->aid) == sr.aid && (uint32_t)rdfirst((int)smp->cp_a) == sr.cp_a && (uint32_t)rdfirst((int)smp->cp_b) == sr.cp_b
								&& (uint32_t)rdfirst((int)smp->rev) == (sr.rev ? 1u : 0u) && (uint32_t)rdfirst((int)smp->bw) == (uint32_t)bw;
							memo1 = memo0 && (bits & 2u) != 0;
							if(memo0) { dg_hits++; if(lane == 0) { atomicAdd(&a.rq_ctl[3], 1u); } }
							break;
						}
						if(!own_job(ji + 1, cj_base + cj_n)) { __builtin_amdgcn_s_sleep(32); uint32_t off = 0; if(lane == 0) { off = k3_wd_tick(wdw, wave, wst, K3_WD_CJOB, ji) ? 1u : 0u; } if(rdfirst((int)off)) { K3_LEAVE(); } }
					}
				}
				if(rq_on && !chain_first) {
					/* a trial mm_search_load_next set up.  If it was published as a job: taken where it is done, run here where nobody has claimed it, and while another wave is at it this
					 * one works on a later job of its own */
					if(rj_i < rj_n) {
						const uint32_t ji = rj_base + rj_i; rj_i++;
						const SpecJob jj = a.rjobs[ji];
						const bool match = (uint32_t)rdfirst((int)jj.aid) == sr.aid && (uint32_t)rdfirst((int)jj.cp_a) == sr.cp_a && (uint32_t)rdfirst((int)jj.cp_b) == sr.cp_b && (uint32_t)rdfirst((int)jj.rev) == (sr.rev ? 1u : 0u) && (uint32_t)rdfirst((int)jj.pad) == (uint32_t)bw;
						if(!match) { rj_i--; cancel_rjobs(); }
						else {
							while(true) {
								uint32_t stt = 0;
								if(lane == 0) { stt = __hip_atomic_load(&a.rstate[ji], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if(stt == RJ_READY) { stt = atomicCAS(&a.rstate[ji], (uint32_t)RJ_READY, (uint32_t)RJ_CANCELLED) == RJ_READY ? (uint32_t)RJ_CANCELLED : (uint32_t)RJ_CLAIMED; } }
								stt = (uint32_t)rdfirst((int)stt);
								if(stt == RJ_CANCELLED) { break; }                                        /* nobody took it: computed below like any trial */
								if(stt == RJ_DONE) {
									__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
									smp = a.rmemo + ji; memo1 = false;
									memo0 = ((uint32_t)rdfirst((int)smp->state) & 1u) != 0 && (uint32_t)rdfirst((int)smp->aid) == sr.aid && (uint32_t)rdfirst((int)smp->cp_a) == sr.cp_a && (uint32_t)rdfirst((int)smp->cp_b) == sr.cp_b
										&& (uint32_t)rdfirst((int)smp->rev) == (sr.rev ? 1u : 0u) && (uint32_t)rdfirst((int)smp->bw) == (uint32_t)bw;          /* (computed for exactly this trial: what the job said when it was run) */
									if(memo0) { dg_hits++; if(lane == 0) { atomicAdd(&a.rq_ctl[3], 1u); } }
									break;
								}
								/* another wave is working on it: one of the later jobs of this read meanwhile */
								if(!own_job(rj_base + rj_i, rj_base + rj_n)) { __builtin_amdgcn_s_sleep(32); uint32_t off = 0; if(lane == 0) { off = k3_wd_tick(wdw, wave, wst, K3_WD_RJOB, ji) ? 1u : 0u; } if(rdfirst((int)off)) { K3_LEAVE(); } }
							}
						}
					}
					if(rj_i >= rj_n && sr.srem > 0) {
						/* nothing published for the trials behind this one: the next-seed list is worked ahead on a copy, as mm_search_load_next would after every duplicate, and the
						 * trials it leads to become jobs (minialign.c:3888-3946; a trial that turns out NOT to be a duplicate cancels what is left of them) */
						rj_n = rj_i = 0;
						uint32_t m_jobs = 0, base = 0;
						if(lane == 0) {
							uint64_t *nx = next + a.next_cap + MM_NEXT_SCRATCH;
							for(uint32_t i = 0; i < next_n; i++) { nx[i] = next[i]; }
							uint32_t c_srem = sr.srem, c_pacc = sr.pacc, c_sid = sr.sid, c_cpa = sr.cp_a, c_cpb = sr.cp_b, c_rev = sr.rev, c_nar = sr.narrow, c_nn = next_n, c_err = 0;
							SpecJob tmp[MM_SREM];
							const uint64_t ofs = 2ull * a.tglen;
							while(m_jobs < MM_SREM && c_srem > 0) {
								c_nar = min(c_nar + 1, 2u);                 /* the trial in front was a duplicate (minialign.c:3977) */
								c_srem--;
								const int32_t fa = (int32_t)c_cpa, fb = (int32_t)(c_cpb - (c_rev ? qlen : 0u));
								const V4 fv = V4{ (int32_t)U_(fa, fb), (int32_t)sr.aid, (int32_t)V_(fa, fb), (int32_t)V_(fa, fb) };
								uint32_t ncnt = c_nn; const uint64_t plim = ofs - c_pacc;
								if(c_pacc > ofs) { ncnt = 0; }
								for(uint32_t i = 0; i < ncnt; i++) { const uint32_t pd = (uint32_t)nx[i]; if(pd >= plim) { ncnt = i; break; } nx[i] = (nx[i] & 0xffffffff00000000ull) | (uint32_t)(pd + c_pacc); }
								uint64_t sid = c_sid;
								for(uint64_t rcnt = 2ull * c_srem; sid > 0 && rcnt > 0; sid--) {
									const V4 pv = load_pv(s[sid - 1]); const V4 wv = add_win(pv, (int32_t)a.tglen), zv = add_win(pv, 128);
									if(!inside_uub(wv, fv)) { break; }
									if(!inside_wv(wv, fv) || inside_wv(zv, fv)) { continue; }
									if(ncnt < a.next_cap) { nx[ncnt++] = (uint64_t)(uint32_t)pdiff(wv, fv) | ((uint64_t)(sid - 1) << 32); } else { c_err = 1; }
									rcnt--;
								}
								c_sid = (uint32_t)sid;
								if(c_err || !radix_sort_64((U64R *)nx, ncnt, next_scratch, 2 * MM_NEXT_SCRATCH)) { break; }          /* (the real walk reports what this one only avoids) */
								if(ncnt == 0) { break; }
								c_nn = ncnt - 1;
								const uint64_t e = nx[c_nn]; const uint32_t nsid = (uint32_t)(e >> 32);
								c_pacc = (uint32_t)(ofs - (uint32_t)e);
								const Seed ns = s[nsid]; const int32_t bs_ = BS(ns);
								c_rev = bs_ < 0; c_cpa = (uint32_t)AS(ns); c_cpb = (uint32_t)(bs_ + ((bs_ >> 31) & (int32_t)qlen));
								if(c_cpa >= rlen || c_cpb >= qlen) { c_cpa -= min(c_cpa, ix.k); c_cpb -= min(c_cpb, ix.k); }
								if(!(c_srem > 0 && sr.prem > 0)) { break; }
								tmp[m_jobs++] = SpecJob{ r, sr.aid, c_cpa, c_cpb, c_rev, rlen, (uint32_t)rcirc, c_nar };
							}
							if(m_jobs) {
								base = atomicAdd(&a.rq_ctl[0], m_jobs);
								if(base + m_jobs > a.rq_cap) { m_jobs = 0; }          /* (the queue is full: the slots stay empty, helpers step over them at the end) */
								for(uint32_t q = 0; q < m_jobs; q++) { a.rjobs[base + q] = tmp[q]; }
							}
						}
						m_jobs = (uint32_t)rdfirst((int)m_jobs); base = (uint32_t)rdfirst((int)base);
						if(m_jobs) {
							__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
							asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
							if(lane == 0) { for(uint32_t q = 0; q < m_jobs; q++) { __hip_atomic_store(&a.rstate[base + q], (uint32_t)RJ_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
							rj_base = base; rj_n = m_jobs; rj_i = 0;
						}
					}
				}
				if(lane == 0) { k3_wd_ran(wdw, wave, wst); }
				chain_first = false; dg_trials++;
				for(int pass = 0; pass < 2 && !skip; pass++) {
					gaba::Sec ca = pass == 0 ? rsec_f : rsec_r;
					gaba::Sec cb = ((sr.rev != 0) == (pass == 0)) ? qsec_r : qsec_f;
					uint32_t sa = pass == 0 ? sr.cp_a : rlen - sr.tp_a, sb = pass == 0 ? sr.cp_b : qlen - sr.tp_b;
					gaba::PosPair pp; pp.aid = pp.bid = 0; pp.apos = pp.bpos = 0; pp.plen = 0;
					if(pass == 0 ? memo0 : memo1) {
						/* a pass another wave ran: its maximum and -- pass 0 -- the position of the maximum, -- pass 1 -- the path length for the pools (its vectors were counted there) */
						mmax = (int64_t)rdfirst64((uint64_t)(pass == 0 ? smp->mmax0 : smp->mmax1)); m = gaba::NIL;
						if(pass == 0 ? (mmax == 0) : (mmax < (int64_t)a.min_score)) { skip = true; break; }
						if(pass == 0) { pp.apos = (uint32_t)rdfirst((int)smp->pp_apos); pp.bpos = (uint32_t)rdfirst((int)smp->pp_bpos); pp.plen = rdfirst64(smp->pp_plen); }
						else { tplen = rdfirst64(smp->tplen); memo_trace = true; }
					} else {
					DpIn din; din.c = x.c; din.ar0 = ar[0]; din.ar1 = ar[1]; din.slab = x.slab; din.top = x.top; din.cap = x.cap;
					const unsigned long long cy0 = MM_TICK();
					/* the downward pass is only searched for its maximum (the walk-back runs on the upward pass): no traceback masks */
					ExtOut eo = k3_extend_core(din, bw, ca, sa, cb, sb, pass == 0, rcirc);
					const unsigned long long cy1 = MM_TICK(); cy_fill += cy1 - cy0;
					x.top = (uint32_t)rdfirst((int)eo.d.top); x.err = rdfirst(eo.d.err); x.n_vec += (uint32_t)rdfirst((int)eo.d.n_vec); x.n_blk += (uint32_t)rdfirst((int)eo.d.n_blk);
					m = (uint32_t)rdfirst((int)eo.m); mmax = (int64_t)rdfirst64((uint64_t)eo.mmax); n_fill += (uint32_t)rdfirst((int)eo.n_fill);
					if(x.err) { skip = true; break; }
					if(pass == 0 ? (mmax == 0) : (mmax < (int64_t)a.min_score)) { skip = true; break; }
					/* leaf_search: for pass 0 this is gaba_dp_search_max, for pass 1 the head of gaba_dp_trace */
					din.top = x.top;
					LeafOut lo = k3_leaf_search(din, m, pass == 0);
					cy_leaf += MM_TICK() - cy1;
					tlf = lo.lf; tplen = rdfirst64(lo.plen);
					if(pass == 0) { pp = lo.pp; pp.apos = (uint32_t)rdfirst((int)pp.apos); pp.bpos = (uint32_t)rdfirst((int)pp.bpos); pp.plen = rdfirst64(pp.plen); }
					}
					if(pass == 0) {
						/* mm_search_test_dup (minialign.c:3953-3982) */
						uint64_t key = mm_key((uint64_t)pp.apos | ((uint64_t)pp.bpos << 32), (uint64_t)sr.aid | ((uint64_t)sr.bid << 32));
						uint64_t prev = 0;
						if(lane == 0) {
							uint64_t ti = kh_put(kh, key, true, &err);
							prev = kh.a[ti].v;
							kh.a[ti].v = (uint64_t)sr.eid | (0xffffffffull << 32);
						}
						prev = rdfirst64(prev); err = (uint32_t)rdfirst((int)err);
						if(err & ERR_KH_CAP) { skip = true; break; }
						int32_t pa = max(1, min((int32_t)pp.apos, (int32_t)rlen)), pb = max(1, min((int32_t)pp.bpos, (int32_t)qlen));
						sr.tp_a = (uint32_t)pa; sr.tp_b = (uint32_t)pb;
						if(prev != ~0ull) {
							/* the reference re-reads the slot it has just overwritten, so the "other chain" test never fires */
							sr.narrow = min(sr.narrow + 1, 2u);
							skip = true;
						}
					}
				}
				if(x.err) { err |= ERR_DP_SLAB; break; }
				if(err & ERR_KH_CAP) { break; }
				if(skip) { continue; }
				/* trace into the output pools */
				if(n_aln >= aln_cap_r) { err |= ERR_ALN_CAP; break; }
				uint64_t need_words = (tplen + 31) / 32 + 2;
				unsigned long long po = 0, so_ = 0;
				if(lane == 0) { po = atomicAdd(a.path_top, (unsigned long long)need_words + 2); so_ = atomicAdd(a.seg_top, 8ull); }
				po = rdfirst64(po); so_ = rdfirst64(so_);
				if(po + need_words + 2 > a.path_pool_cap || so_ + 8 > a.seg_pool_cap) { err |= ERR_PATH_CAP; break; }
				uint32_t *path = a.path_pool + po + 2;
				DpIn din2; din2.c = x.c; din2.ar0 = ar[0]; din2.ar1 = ar[1]; din2.slab = x.slab; din2.top = x.top; din2.cap = x.cap;
				const unsigned long long cy2 = MM_TICK();
				gaba::AlnOut ao;
				if(memo_trace) {
					/* the traceback was done by the job: its path words and segments move from the staging area into the pools */
					const uint32_t *sp = a.spath + rdfirst64(smp->path_off); const gaba::Segment *sg = a.sseg + (uint32_t)rdfirst((int)smp->seg_off);
					for(uint64_t i = (uint64_t)lane; i < need_words; i += 64) { path[i] = sp[i]; }
					ao = smp->ao; ao.status = rdfirst(ao.status); ao.plen = (uint32_t)rdfirst((int)ao.plen); ao.slen = (uint32_t)rdfirst((int)ao.slen);
					for(uint32_t i = (uint32_t)lane; i < ao.slen && i < 8; i += 64) { a.seg_pool[so_ + i] = sg[i]; }
					x.err = 0;
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				} else {
				TraceOut to = k3_trace(din2, m, tlf, tplen, path, a.seg_pool + so_);
				cy_trace += MM_TICK() - cy2;
				ao = to.ao; x.err = rdfirst(to.d.err); x.n_tr += (uint32_t)rdfirst((int)to.d.n_tr);
				ao.status = rdfirst(ao.status); ao.plen = (uint32_t)rdfirst((int)ao.plen); ao.slen = (uint32_t)rdfirst((int)ao.slen);
				n_trace++;
				}
				if(x.err) { err |= (x.err == 1 ? ERR_DP_SLAB : (x.err == 2 ? ERR_PATH_CAP : ERR_SEG_CAP)); break; }
				if(ao.status != 1) { continue; }           /* NULL alignment: path left the band */
				uint32_t ai = n_aln++;
				if(lane == 0) {
					a.path_pool[po] = ao.plen; a.path_pool[po + 1] = 0x40000000u;
					AlnRec *ar_ = &alns[ai];
					ar_->score = ao.score; ar_->identity = ao.identity; ar_->agcnt = ao.agcnt; ar_->bgcnt = ao.bgcnt; ar_->dcnt = ao.dcnt;
					ar_->slen = ao.slen; ar_->plen = ao.plen; ar_->seg_off = (uint32_t)so_; ar_->path_off = po + 2;
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				/* mm_search_record (minialign.c:4018-4067) */
				const gaba::Segment *segs = a.seg_pool + so_;
				gaba::Segment sl = segs[ao.slen - 1], s0 = segs[0];
				uint32_t p0 = rlen - ((uint32_t)rdfirst((int)sl.apos) + (uint32_t)rdfirst((int)sl.alen)), p1 = qlen - ((uint32_t)rdfirst((int)sl.bpos) + (uint32_t)rdfirst((int)sl.blen));
				uint32_t p2 = rlen - (uint32_t)rdfirst((int)s0.apos), p3 = qlen - (uint32_t)rdfirst((int)s0.bpos);
				sr.cp_a = p0; sr.cp_b = p1;
				sr.prem -= ao.plen; sr.pacc = ao.plen;
				uint64_t id = (uint64_t)sr.aid | ((uint64_t)sr.bid << 32);
				uint64_t hk = mm_key((uint64_t)p0 | ((uint64_t)p1 << 32), id), tk = mm_key((uint64_t)p2 | ((uint64_t)p3 << 32), id);
				uint32_t isnew = 0;
				if(lane == 0) {
					/* h is taken before the second insert, which may shift entries under it (minialign.c:4027-4029): indices, literally */
					uint64_t hi = kh_put(kh, hk, true, &err);
					uint64_t ti = kh_put(kh, tk, false, &err);
					isnew = (uint32_t)(kh.a[hi].v >> 32) == 0xffffffffu;
					uint32_t nid;
					if(isnew) { nid = n_bin; if(n_bin < bin_cap_r) { bin[n_bin] = (uint64_t)ai + 1; } else { err |= ERR_BIN_CAP; } }
					else { nid = (uint32_t)(kh.a[hi].v >> 32); }
					uint32_t *hdr = (uint32_t *)&bin[sr.iid];           /* { n_aln, plen, lb, ub } */
					uint32_t lb = hdr[2], ubb = hdr[3];
					uint32_t ovl = max(lb, p1) - min(ubb, p3) - p1 + p3;
					Root *rr = &root[sr.eid];
					rr->plen -= (uint32_t)(ao.score + (int64_t)d2u32((double)(uint32_t)(ovl * 2) * ao.identity));
					hdr[0] += isnew; hdr[1] += ao.plen; hdr[2] = min(lb, p1); hdr[3] = max(ubb, p3);
					uint32_t cur = nid < bin_cap_r ? (uint32_t)bin[nid] - 1 : ai;
					int64_t bscore = alns[cur].score;
					if(bscore > ao.score) {
						kh.a[ti].v = (uint64_t)sr.eid | (0xffffffffull << 32);
					} else {
						if(cur != ai && nid < bin_cap_r) { bin[nid] = (uint64_t)ai + 1; }
						uint64_t nv = (uint64_t)sr.eid | ((uint64_t)nid << 32);
						kh.a[ti].v = nv; kh.a[hi].v = nv;              /* *h = *t = ... (t first, then h, as the chained assignment evaluates) */
					}
				}
				isnew = (uint32_t)rdfirst((int)isnew); err = (uint32_t)rdfirst((int)err);
				n_bin += isnew;
				sr.srem = MM_SREM; sr.narrow = 0;
				if(rq_on) { cancel_rjobs(); }
				{
					float cand = (float)ao.score * a.min_ratio, cur = (float)sr.min_score;
					sr.min_score = f2u32(cur > cand ? cur : cand);
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				if(!(isnew && sr.prem > 0)) { break; }
			}
			if(rq_on) { cancel_rjobs(); }
			if(err & (ERR_DP_SLAB | ERR_PATH_CAP | ERR_ALN_CAP | ERR_SEG_CAP | ERR_KH_CAP)) { break; }
			/* mm_finish_root (minialign.c:3795-3813) */
			{
				uint32_t *hdr = (uint32_t *)&bin[sr.iid];
				uint32_t bn = (uint32_t)rdfirst((int)hdr[0]); uint32_t sc = (uint32_t)rdfirst((int)root[sr.eid].plen);
				if(bn == 0 || sc > (uint32_t)OFS((int32_t)a.min_score)) { n_bin = sr.iid; n_res--; sr.crem--; }
				else { sr.crem = sr.crem != 0 ? MM_CREM : 0; }
				if(sr.crem == 0) { break; }
			}
		}
		#undef LOAD_POS
		if(cj_n) {
			/* the walk is over (or gave up): what is left unclaimed of the read's chain jobs is withdrawn, and the waves that stayed for this read may go */
			for(uint32_t kj = (uint32_t)lane; kj < cj_n; kj += 64) { (void)atomicCAS(&a.rstate[cj_base + kj], (uint32_t)RJ_READY, (uint32_t)RJ_CANCELLED); }
			if(lane == 0) { atomicSub(&a.rq_ctl[4], 1u); }
		}
		if(lane == 0) {
			st->n_res = n_res; st->rlen = rlen; st->rid_last = rid_last; st->apos0 = apos0; st->cond0 = cond0;
			st->n_bin = n_bin; st->bin_off = bin_off; st->n_aln = n_aln; st->aln_off = aln_off;
			st->kh_mask = kh.mask; st->kh_cnt = kh.cnt; st->kh_ub = kh.ub;
			st->err |= err;
			st->k3_trials += dg_trials; st->k3_hits += dg_hits; st->k3_chains += dg_chains;
			st->k3_ticks += (uint32_t)(MM_TICK() - cy_read0); st->k3_vec += x.n_vec - vec_read0;
			st->k3_fill_ticks += (uint32_t)(cy_fill - cyf_read0); st->k3_trace_ticks += (uint32_t)(cy_trace - cyt_read0);
			if(round == a.round) { st->k3_t0 = (uint32_t)(cy_read0 >> 8); st->k3_wait_ticks = (uint32_t)(cy_slab - cy_read0); }
			if(n_res > 0) { st->done = 1; }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if(!a.inkernel_rounds || n_res > 0 || err != 0 || round + 1 >= ix.n_occ) { break; }
		}
		if(((uint32_t)rdfirst((int)st->flags) & RS_CARRY_SRC) != 0u) {
			/* a read whose end decides what the reads behind it start with: its state is out (st->rlen above), then the flag */
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if(lane == 0) { __hip_atomic_store(&st->carry_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		}
		if(rq_on && lane == 0) { atomicAdd(&a.rq_ctl[2], 1u); }          /* (helper waves leave when every read is done) */
	}
	if(lane == 0) {
		atomicAdd(&a.stats[2], n_fill); atomicAdd(&a.stats[3], (unsigned long long)x.n_vec); atomicAdd(&a.stats[4], (unsigned long long)x.n_blk);
		atomicAdd(&a.stats[5], n_trace); atomicAdd(&a.stats[6], (unsigned long long)x.n_tr);
		atomicAdd(&a.stats[12], cy_fill); atomicAdd(&a.stats[13], cy_leaf); atomicAdd(&a.stats[14], cy_trace); atomicAdd(&a.stats[11], cy_next);
		atomicAdd(&a.stats[15], (unsigned long long)(__builtin_amdgcn_s_memtime() - cy_begin));
		atomicMax(&a.stats[9], (unsigned long long)(__builtin_amdgcn_s_memtime() - cy_begin));      /* longest-living wave: load balance */
	}
	if(a.ring) { give_slab(); }
	if(lane == 0) { k3_wd_mark(wdw, wave, 0, 1); }
	#undef K3_LEAVE
}
