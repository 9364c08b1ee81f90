/* the index on the host side: text sources, the index object and its replicas, the device build's driver, the host build (mm_idx_gen, minialign.c:2951-3040), .mai files, mm_idx_get / mm_sketch -- part of mm_host.hip (included from there at the place it stood; split out in round 6 so that it can be read on its own) */
/* =============================================================================================
 * index (host): mm_idx_gen, minialign.c:2951-3040
 * ============================================================================================= */
/* ---- the text of an input file in host memory (a mapping of the file, or what stdin / gzip gave) and its records as the device reader finds them ---- */
struct TextSrc {
	const char *p = nullptr; uint64_t n = 0; char delim = 0; uint64_t first = 0;      /* delim / first: the record delimiter and where the first record starts (minialign.c:1784-1792) */
	void *map = nullptr; uint64_t map_len = 0; std::vector<char> own;
	~TextSrc() { if(map) munmap(map, map_len); }
};
struct RRec { uint64_t start, hdr_end, t_off; uint32_t t_len, n_bases; uint64_t q_off; uint32_t q_len; };      /* absolute offsets in the text: delimiter, end of the header line, sequence extent, quality extent */
struct DevChunk { uint8_t *d = nullptr; uint64_t cap = 0; uint64_t off = 0; uint32_t n = 0; };                  /* a stretch of the text in HBM: text[off, off + n) */
struct ChunkPool {          /* device buffers for stretches of text, reused while a context lives (a hipFree in mid-run would stall every stream of the device) */
	std::mutex mu; std::vector<DevChunk *> idle, all;
	DevChunk *get(uint64_t bytes)
	{
		{ std::lock_guard<std::mutex> lk(mu); for(size_t i = 0; i < idle.size(); i++) { if(idle[i]->cap >= bytes) { DevChunk *c = idle[i]; idle.erase(idle.begin() + i); return c; } } }
		DevChunk *c = new DevChunk(); if(hipMalloc(&c->d, bytes) != hipSuccess) { delete c; return nullptr; } c->cap = bytes;
		std::lock_guard<std::mutex> lk(mu); all.push_back(c); return c;
	}
	void put(DevChunk *c) { std::lock_guard<std::mutex> lk(mu); idle.push_back(c); }
	~ChunkPool() { for(DevChunk *c : all) { (void)hipFree(c->d); delete c; } }
};
struct mm_idx_s {
	uint32_t b, w, k, n_occ; uint32_t occ[8];
	std::vector<HSeq> seq;
	/* flattened table, also what the device gets */
	std::vector<IdxSlot> slot; uint64_t mask;
	std::vector<uint64_t> val;
	uint64_t n_keys = 0;
	/* an index built on the device (idx_gen_device) lives there: the table, the value array (the sorted (pos | rid << 32) of every minimizer; a list is a run inside it)
	 * and the packed reference; mm_align_init adopts them, the host copy above is fetched only when somebody asks (mm_idx_dump, mm_idx_get) */
	std::shared_ptr<TextSrc> rtext; std::vector<RRec> rrec;          /* the reference's text and records when the device reader scanned it (the sequences' base codes are then made on demand, ref_codes) */
	bool on_device = false; int dev = 0; IdxSlot *d_slot = nullptr; uint64_t *d_val = nullptr; uint64_t n_slot = 0, n_val = 0; gaba_arena_t *ref_ar = nullptr;
	mutable std::mutex fetch_mu;
	double build_ms[8] = { 0 };          /* device build: arena, sketch, partition, sort, thresholds, table */
	/* copies of a device-built index on the other devices of the node (one per device, made when a context on that device asks: idx_replica; device to device,
	 * the host never sees the tables) -- "the minimizer index replicated into each GPU's HBM" of north_star */
	/* state: 0 being copied (by the thread that made the entry), 1 ready, -1 failed; serving: copies that read from this holder right now */
	struct Rep { int key, dev; IdxSlot *d_slot; uint64_t *d_val; gaba_arena_t *ref_ar; int state; int serving; };          /* key: the device, or (test hook) 1000 + the context's number */
	std::vector<std::unique_ptr<Rep>> reps; std::mutex rep_mu; std::condition_variable rep_cv; int serving0 = 0;          /* serving0: copies reading from the originals */
	~mm_idx_s()
	{
		if(d_slot) (void)hipFree(d_slot); if(d_val) (void)hipFree(d_val); if(ref_ar) gaba_arena_free(ref_ar);
		for(auto &r : reps) { if(r->d_slot) (void)hipFree(r->d_slot); if(r->d_val) (void)hipFree(r->d_val); if(r->ref_ar) gaba_arena_free(r->ref_ar); }
	}
};
/* the tables and the packed reference of a device-built index on device `dev`: the originals on the device that built them, a copy anywhere else (made once per
 * device, by the first context that asks; the others of that device wait for it).  The contexts of a node ask side by side (mm_align_init: one thread per device), so
 * nothing but the bookkeeping is under the index's lock: a copy reads from whichever holder -- the originals or a replica that is complete -- serves the fewest copies
 * right now, at most MM_REPLICA_FANOUT (default 4) per holder: xGMI is point to point, so copies out of one device to different devices run on different links, and
 * once the first replicas are complete they serve the rest.  Eight devices, 20 GB (a human-size index + packed reference): four copies out of the builder side by side,
 * the other three from three of those four -- two link times, 2 x 20 GB / (what one xGMI link sustains) instead of seven in a row behind one lock as in round 4.
 * forced > 0 (MM_TEST_REPLICA, one-GPU boxes; the number of the context that asks): a context takes the copy path whatever device it is on, and gets a replica
 * of its own (device-to-device on one device), so that allocation, hipMemcpyPeer, the choice among several holders, adoption and release of replicas run where there
 * is no second device.  false: out of memory / copy failed */
static bool idx_replica(const mm_idx_s *cmi, int dev, int forced, IdxSlot **slot, uint64_t **val, gaba_arena_t **ar)
{
	mm_idx_s *mi = const_cast<mm_idx_s *>(cmi);
	if(dev == mi->dev && forced <= 0) { *slot = mi->d_slot; *val = mi->d_val; *ar = mi->ref_ar; return true; }
	const int key = forced > 0 ? 1000 + forced : dev;
	const int fanout = getenv("MM_REPLICA_FANOUT") ? std::max(1, atoi(getenv("MM_REPLICA_FANOUT"))) : 4;
	std::unique_lock<std::mutex> lk(mi->rep_mu);
	for(;;) {
		mm_idx_s::Rep *have = nullptr;
		for(auto &r : mi->reps) if(r->key == key) { have = r.get(); break; }
		if(!have) break;
		if(have->state == 0) { mi->rep_cv.wait(lk); continue; }          /* another context of this device is making it */
		if(have->state < 0) return false;
		*slot = have->d_slot; *val = have->d_val; *ar = have->ref_ar; return true;
	}
	mi->reps.emplace_back(new mm_idx_s::Rep{ key, dev, nullptr, nullptr, nullptr, 0, 0 });
	mm_idx_s::Rep *r = mi->reps.back().get();
	/* the source: the holder that serves the fewest copies, once one is below the fan-out */
	mm_idx_s::Rep *src = nullptr; bool from_orig = false;
	for(;;) {
		int best = mi->serving0; from_orig = true; src = nullptr;
		for(auto &q : mi->reps) if(q->state == 1 && q->serving <= best) { best = q->serving; src = q.get(); from_orig = false; }          /* (a tie goes to a replica: the builder's device has the first contexts' work already) */
		if(best < fanout) break;
		mi->rep_cv.wait(lk);
	}
	if(from_orig) mi->serving0++; else src->serving++;
	const int sdev = from_orig ? mi->dev : src->dev;
	const IdxSlot *s_slot = from_orig ? mi->d_slot : src->d_slot; const uint64_t *s_val = from_orig ? mi->d_val : src->d_val; const gaba_arena_t *s_ar = from_orig ? mi->ref_ar : src->ref_ar;
	lk.unlock();
	const double t0 = now_ms();
	const uint64_t n = mi->ref_ar->n, nw = (n + 15) / 16 + 4, nn = (n + 31) / 32 + 4;
	gaba_arena_t *q = (gaba_arena_t *)calloc(1, sizeof(gaba_arena_t));
	bool ok = q != nullptr && hipSetDevice(dev) == hipSuccess && hipMalloc(&r->d_slot, mi->n_slot * sizeof(IdxSlot)) == hipSuccess && hipMalloc(&r->d_val, (mi->n_val + 64) * 8) == hipSuccess
		&& hipMalloc(&q->pk, nw * 4) == hipSuccess && hipMalloc(&q->nm, nn * 4) == hipSuccess;
	ok = ok && hipMemcpyPeer(r->d_slot, dev, s_slot, sdev, mi->n_slot * sizeof(IdxSlot)) == hipSuccess && hipMemcpyPeer(r->d_val, dev, s_val, sdev, (mi->n_val + 64) * 8) == hipSuccess
		&& hipMemcpyPeer(q->pk, dev, s_ar->pk, sdev, nw * 4) == hipSuccess && hipMemcpyPeer(q->nm, dev, s_ar->nm, sdev, nn * 4) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
	if(!ok) { if(r->d_slot) (void)hipFree(r->d_slot); if(r->d_val) (void)hipFree(r->d_val); if(q) { if(q->pk) (void)hipFree(q->pk); if(q->nm) (void)hipFree(q->nm); free(q); } r->d_slot = nullptr; r->d_val = nullptr; q = nullptr; }
	if(ok) { q->n = n; q->host = NULL; r->ref_ar = q; }
	if(getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd] index replica on device %d from device %d (%s): %.1f MB in %.1f ms%s\n", dev, sdev, from_orig ? "the originals" : "a replica", (mi->n_slot * sizeof(IdxSlot) + (mi->n_val + 64) * 8 + nw * 4 + nn * 4) * 1e-6, now_ms() - t0, ok ? "" : " -- FAILED");
	lk.lock();
	if(from_orig) mi->serving0--; else src->serving--;
	r->state = ok ? 1 : -1;
	mi->rep_cv.notify_all();
	if(!ok) return false;
	*slot = r->d_slot; *val = r->d_val; *ar = r->ref_ar;
	return true;
}
/* host copy of a device-built index (for mm_idx_dump / mm_idx_get) */
static bool idx_fetch_host(const mm_idx_s *cmi)
{
	mm_idx_s *mi = const_cast<mm_idx_s *>(cmi);
	std::lock_guard<std::mutex> lk(mi->fetch_mu);
	if(!mi->on_device || !mi->slot.empty()) return true;
	(void)hipSetDevice(mi->dev);
	mi->slot.resize(mi->n_slot); mi->val.resize(std::max<uint64_t>(mi->n_val, 1));
	return hipMemcpy(mi->slot.data(), mi->d_slot, mi->n_slot * sizeof(IdxSlot), hipMemcpyDeviceToHost) == hipSuccess && (mi->n_val == 0 || hipMemcpy(mi->val.data(), mi->d_val, mi->n_val * 8, hipMemcpyDeviceToHost) == hipSuccess);
}


/* device memory of the library's buffers is recycled, not handed back: a hipFree waits for every stream of the device (a lane's pool that grew used to stall all
 * lanes for 100 - 450 ms), and pages the driver has taken back are wiped before they are handed out again -- the 43 GB of DP workspaces allocated right after the
 * index build's 45 GB of temporaries had been freed took 3.7 s to arrive (1.35 s behind 20 GB), against a few milliseconds on untouched memory.  A released block
 * waits here for the next request it fits (at most four times the size asked for), per device; what is held is bounded by what the library once used. */
#define MM_BATCH_PER_LONGEST 1700ull          /* bases of batch per base of the longest read of the input (batch_spans) */
struct DevCache {
	std::mutex mu; std::multimap<std::pair<int, size_t>, void *> blocks;
	void *take(int dev, size_t want, size_t *got)
	{
		std::lock_guard<std::mutex> lk(mu);
		auto it = blocks.lower_bound(std::make_pair(dev, want));
		if(it == blocks.end() || it->first.first != dev || it->first.second > 4 * want + (64u << 20)) return nullptr;
		void *p = it->second; *got = it->first.second; held[dslot(dev)] -= *got; blocks.erase(it); return p;
	}
	static const int MAX_DEV = 16;
	static int dslot(int dev) { return dev >= 0 && dev < MAX_DEV ? dev : MAX_DEV - 1; }
	size_t held[MAX_DEV] = { 0 };          /* bytes waiting here, per device (read and written under mu) */
	size_t held_on(int dev) { std::lock_guard<std::mutex> lk(mu); return held[dslot(dev)]; }
	/* what is held is bounded (32 GB): a block that would take it beyond that goes back to the driver after all, as do blocks of more than 16 GB (the DP workspaces of a size that is
	 * being replaced: nothing is in flight then) -- an ONT-like run that re-sized its workspace ladder a few times had 150 GB of them waiting here and the runtime ran out of memory */
	void give(int dev, size_t bytes, void *p)
	{
		/* nothing is kept while the device is short of memory (the runtime allocates the kernels' scratch memory on demand and aborts the process when it cannot).
		 * `tight` is what the last fresh allocation found: hipMemGetInfo costs about 2 ms, and a stream gives a dozen buffers back when it ends -- asked here, it made every
		 * stream 20 ms longer, 12 % of one over an E.coli-size set */
		{ std::lock_guard<std::mutex> lk(mu); const int d = dslot(dev); if(!tight[d].load() && bytes <= (16ull << 30) && held[d] + bytes <= (32ull << 30)) { blocks.emplace(std::make_pair(dev, bytes), p); held[d] += bytes; return; } }
		(void)hipFree(p);
	}
	std::atomic<bool> tight[MAX_DEV];          /* per device: one device running short of memory does not stop the others from recycling */
	DevCache() { for(int i = 0; i < MAX_DEV; i++) tight[i].store(false); }
	static size_t reserve() { return 24ull << 30; }
	/* may `bytes` more be taken?  Not when less than 8 GB would be left after giving back what is held here: the runtime allocates the scratch memory of a kernel when it is first
	 * launched on a queue (1.7 GB for the extension kernel) and aborts the process when it cannot -- an allocation that fails cleanly is the better end */
	bool room(int dev, size_t bytes)
	{
		size_t fr = 0, tot = 0; if(hipMemGetInfo(&fr, &tot) != hipSuccess) return true;
		if(fr >= bytes + (8ull << 30)) return true;
		flush(dev); if(hipMemGetInfo(&fr, &tot) != hipSuccess) return true;
		return fr >= bytes + (8ull << 30);
	}
	/* after a fresh allocation: what is held goes back to the driver when less than the reserve is left */
	void relieve(int dev) { size_t fr = 0, tot = 0; const bool t = hipMemGetInfo(&fr, &tot) != hipSuccess || fr < reserve(); tight[dslot(dev)].store(t); if(t && held_on(dev)) flush(dev); }
	void flush(int dev) { std::lock_guard<std::mutex> lk(mu); for(auto it = blocks.begin(); it != blocks.end();) { if(it->first.first == dev) { (void)hipFree(it->second); held[dslot(dev)] -= it->first.second; it = blocks.erase(it); } else ++it; } }          /* out of memory: everything held goes back to the driver */
};
static DevCache &dev_cache() { static DevCache *c = new DevCache(); return *c; }          /* (never destroyed: blocks may come back while the process winds down) */
template<typename T> struct DBuf {
	T *p = nullptr; uint64_t n = 0; size_t bytes = 0; int dev = 0;
	DBuf() {}
	DBuf(const DBuf &) = delete; DBuf &operator=(const DBuf &) = delete;
	~DBuf() { release(); }          /* temporaries (index build, reference reader) go back to the cache on every way out of their function */
	bool ensure(uint64_t want)
	{
		if(want <= n) return true;
		release();
		(void)hipGetDevice(&dev);
		size_t got = 0; void *q = dev_cache().take(dev, want * sizeof(T), &got);
		const bool fresh = q == nullptr;
		if(!q && !dev_cache().room(dev, want * sizeof(T))) { fprintf(stderr, "[minialign_amd] %.1f MB more would leave the device without the memory its runtime needs (kernel scratch): fewer lanes (MM_LANES) or smaller batches (MM_BATCH_BASES) fit\n", want * sizeof(T) / 1048576.0); return false; }
		if(!q) { got = want * sizeof(T); if(hipMalloc(&q, got) != hipSuccess) { q = nullptr; dev_cache().flush(dev); if(hipMalloc(&q, got) != hipSuccess) { fprintf(stderr, "[minialign_amd] hipMalloc of %.1f MB failed\n", got / 1e6); return false; } } }
		p = (T *)q; bytes = got; n = got / sizeof(T); if(fresh) dev_cache().relieve(dev); return true;
	}
	void release() { if(p) dev_cache().give(dev, bytes, p); p = nullptr; n = 0; bytes = 0; }
};

/* base codes (0..3, 4 = N) of reference sequence i.  A reference the device reader scanned keeps its bases in the text; the few consumers on the host (MD:Z, MAF rows,
 * index files, the wrap-around sketch of a circular sequence) have them made here, once */
static const std::vector<uint8_t> &ref_codes(const mm_idx_s *cmi, uint32_t i)
{
	mm_idx_s *mi = const_cast<mm_idx_s *>(cmi); HSeq &q = mi->seq[i];
	if(!mi->rtext || q.len == 0) return q.seq;
	std::lock_guard<std::mutex> lk(mi->fetch_mu);
	if(q.seq.empty()) {
		static const uint8_t enc[16] = { 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 4, 0 };
		const RRec &r = mi->rrec[i]; std::vector<uint8_t> c(r.n_bases);
		const char *p = mi->rtext->p + r.t_off, *e = p + r.t_len; uint32_t k = 0;
		for(; p < e && k < r.n_bases; p++) { if(*p != '\n') c[k++] = enc[*p & 15]; }
		q.seq.swap(c);
	}
	return q.seq;
}
static bool ref_to_device(const mm_opt_s *o, mm_idx_s *mi, const char *fn, std::vector<uint64_t> &off, std::vector<uint32_t> &len);
/* the packed reference in HBM: one arena, every sequence on a multiple of 64 bases */
static gaba_arena_t *upload_reference(const mm_idx_s *mi, std::vector<uint64_t> *off_out, std::vector<uint32_t> *len_out)
{
	uint64_t total = 0; std::vector<uint64_t> off; std::vector<uint32_t> len;
	for(const HSeq &s : mi->seq) { off.push_back(total); len.push_back(s.blen()); total += ((uint64_t)s.blen() + 63) & ~63ull; }
	std::vector<uint8_t> all(total + 64, 4);
	host_parallel((uint32_t)mi->seq.size(), [&](uint32_t t, uint32_t nth) { for(size_t i = t; i < mi->seq.size(); i += nth) { const std::vector<uint8_t> &c = ref_codes(mi, (uint32_t)i); memcpy(all.data() + off[i], c.data(), c.size()); } }, 32);
	gaba_arena_t *ar = gaba_arena_upload(all.data(), total + 64);
	if(ar) gaba_arena_unregister(ar);                 /* `all` is a temporary: keep it out of the per-call API's section lookup */
	if(off_out) off_out->swap(off);
	if(len_out) len_out->swap(len);
	return ar;
}
/* mm_idx_gen on the device (mm_index.hpp): sketch of the reference, stable partition into the 2^b buckets in reference order, the unstable per-bucket sort replayed,
 * occurrence thresholds from the histogram of key counts, table fill.  The sequences are parsed by the host (mi->seq); circular ones (-c) are sketched there too
 * (their wrap-around pass is a serial special case, mm_sketch_cap :2437) and take their place in reference order.  false: no device / out of memory / a bucket beyond
 * what the sort's entries address -- the caller reports it (no silent host build). */
static bool idx_gen_device(const mm_opt_s *o, mm_idx_s *mi, const char *ref_fasta, bool verbose)
{
	double tv = now_ms(); int lapi = 0;
	int ndev = 0; if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "[minialign_amd] mm_idx_gen: no HIP device (MM_HOST_INDEX=1 builds the index on the host)\n"); return false; }
	(void)hipGetDevice(&mi->dev); (void)hipFree(0);
	if(verbose) { fprintf(stderr, "[minialign_amd] index (device): HIP runtime up after %.1f ms\n", now_ms() - tv); } tv = now_ms();
	auto lap = [&](const char *what) { (void)hipDeviceSynchronize(); const double t = now_ms(); if(lapi < 8) mi->build_ms[lapi++] = t - tv; if(verbose) fprintf(stderr, "[minialign_amd] index (device): %s %.1f ms\n", what, t - tv); tv = t; };
#define IK(_e) do { hipError_t _r = (_e); if(_r != hipSuccess) { fprintf(stderr, "[minialign_amd] index (device): HIP error %s at line %d\n", hipGetErrorString(_r), __LINE__); return false; } } while(0)
	std::vector<uint64_t> off; std::vector<uint32_t> len;
	/* the reference goes the way the reads go: its text to HBM, records found (K0r) and bases packed (K0) there, straight into the arena */
	if(!ref_to_device(o, mi, ref_fasta, off, len) || mi->seq.empty()) { fprintf(stderr, "[minialign_amd] cannot read reference `%s'\n", ref_fasta); return false; }
	if(o->circ_set) for(HSeq &q : mi->seq) q.circular = o->circ_names.empty() || std::find(o->circ_names.begin(), o->circ_names.end(), q.name) != o->circ_names.end();
	lap("reference: text to HBM, records, packed arena");
	const uint32_t bbits = mi->b, nb = 1u << bbits, k = mi->k, w = mi->w;
	/* stretches: 2^18 positions each; a circular sequence is one stretch, sketched on the host */
	std::vector<RefStretch> st; std::vector<std::vector<HMin>> hostmin;
	const uint32_t chunk = 1u << 18;
	for(uint32_t i = 0; i < mi->seq.size(); i++) {
		const uint32_t L = len[i];
		if(mi->seq[i].circular) { st.push_back(RefStretch{ off[i], L, 0, L, i, 0, 1, (uint32_t)hostmin.size() }); hostmin.emplace_back(); continue; }
		for(uint32_t bg = 0; bg < L; bg += chunk) st.push_back(RefStretch{ off[i], L, bg, std::min(L, bg + chunk), i, 0, 0, 0 });
	}
	if(!hostmin.empty()) {
		std::vector<uint32_t> which; for(uint32_t t = 0; t < st.size(); t++) if(st[t].host) which.push_back(t);
		host_parallel((uint32_t)which.size(), [&](uint32_t t, uint32_t nth) { for(size_t j = t; j < which.size(); j += nth) { const RefStretch &q = st[which[j]]; const std::vector<uint8_t> &sq = ref_codes(mi, q.seq); sketch_host_circular(sq.data(), (uint32_t)sq.size(), k, w, hostmin[q.pad]); } }, 32);
	}
	const uint32_t n_st = (uint32_t)st.size();
	DBuf<RefStretch> d_st; DBuf<uint32_t> d_cnt, d_ctr; DBuf<IdxMini> d_min, d_flat;
	if(!d_st.ensure(std::max<uint32_t>(n_st, 1)) || !d_cnt.ensure(std::max<uint32_t>(n_st, 1)) || !d_ctr.ensure(16)) return false;
	IK(hipMemcpy(d_st.p, st.data(), (size_t)n_st * sizeof(RefStretch), hipMemcpyHostToDevice)); IK(hipMemset(d_ctr.p, 0, 64)); IK(hipMemset(d_cnt.p, 0, (size_t)std::max<uint32_t>(n_st, 1) * 4));
	hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, mi->dev);
	const uint32_t waves = (uint32_t)prop.multiProcessorCount * 32u;
	I1Args i1; i1.ar = gaba::SeqArena{ mi->ref_ar->pk, mi->ref_ar->nm }; i1.st = d_st.p; i1.n = n_st; i1.k = k; i1.w = w; i1.out = nullptr; i1.count = d_cnt.p; i1.emit = 0; i1.counter = d_ctr.p;
	if(n_st) { hipLaunchKernelGGL(mm_ref_sketch_kernel, dim3(std::min<uint32_t>((n_st + 3) / 4, waves / 4)), dim3(256), 0, 0, i1); IK(hipGetLastError()); }
	std::vector<uint32_t> cnt(n_st);
	if(n_st) IK(hipMemcpy(cnt.data(), d_cnt.p, (size_t)n_st * 4, hipMemcpyDeviceToHost));
	uint64_t N = 0;
	for(uint32_t t = 0; t < n_st; t++) { if(st[t].host) cnt[t] = (uint32_t)hostmin[st[t].pad].size(); st[t].out = N; N += cnt[t]; }
	if(!d_min.ensure(N + 64) || !d_flat.ensure(N + 64)) return false;
	if(n_st) {
		IK(hipMemcpy(d_st.p, st.data(), (size_t)n_st * sizeof(RefStretch), hipMemcpyHostToDevice)); IK(hipMemset(d_ctr.p, 0, 64));
		i1.out = d_min.p; i1.emit = 1;
		hipLaunchKernelGGL(mm_ref_sketch_kernel, dim3(std::min<uint32_t>((n_st + 3) / 4, waves / 4)), dim3(256), 0, 0, i1); IK(hipGetLastError());
		for(uint32_t t = 0; t < n_st; t++) {
			if(!st[t].host || cnt[t] == 0) continue;
			std::vector<IdxMini> tmp(cnt[t]); const std::vector<HMin> &hm = hostmin[st[t].pad];
			for(uint32_t j = 0; j < cnt[t]; j++) tmp[j] = IdxMini{ hm[j].hash, hm[j].pos, (st[t].seq << 1) + hm[j].strand };
			IK(hipMemcpy(d_min.p + st[t].out, tmp.data(), (size_t)cnt[t] * sizeof(IdxMini), hipMemcpyHostToDevice));
		}
	}
	lap("sketch");
	/* stable partition into buckets */
	const uint32_t tile = 1u << 16, n_tiles = (uint32_t)((N + tile - 1) / tile);
	DBuf<uint32_t> d_hist; DBuf<uint64_t> d_bofs;
	if(!d_hist.ensure((uint64_t)std::max<uint32_t>(n_tiles, 1) * nb) || !d_bofs.ensure(nb + 2)) return false;
	I2Args i2; i2.in = d_min.p; i2.n = N; i2.tile = tile; i2.n_tiles = n_tiles; i2.bbits = bbits; i2.hist = d_hist.p; i2.bofs = d_bofs.p; i2.out = d_flat.p;
	std::vector<uint64_t> bofs(nb + 1, 0);
	if(n_tiles) { hipLaunchKernelGGL(mm_idx_hist_kernel, dim3(n_tiles), dim3(64), 0, 0, i2); IK(hipGetLastError()); }
	hipLaunchKernelGGL(mm_idx_colscan_kernel, dim3((nb + 255) / 256), dim3(256), 0, 0, i2); IK(hipGetLastError());
	IK(hipMemcpy(bofs.data(), d_bofs.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost));
	bofs[0] = 0; for(uint32_t bi = 0; bi < nb; bi++) bofs[bi + 1] += bofs[bi];
	IK(hipMemcpy(d_bofs.p, bofs.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice));
	if(n_tiles) { hipLaunchKernelGGL(mm_idx_scatter_kernel, dim3(n_tiles), dim3(64), 0, 0, i2); IK(hipGetLastError()); }
	lap("bucket partition");
	d_hist.release(); d_min.release();
	/* per-bucket sort, records moved into (hrem, val) */
	DBuf<uint32_t> d_ent, d_err; DBuf<uint64_t> d_hrem;
	uint64_t *d_val = nullptr;
	if(!d_ent.ensure(N + 64) || !d_hrem.ensure(N + 64) || !d_err.ensure(4) || hipMalloc(&d_val, (N + 64) * 8) != hipSuccess) return false;
	mi->d_val = d_val; mi->n_val = N;
	IK(hipMemset(d_ctr.p, 0, 64)); IK(hipMemset(d_err.p, 0, 16));
	I3Args i3; i3.in = d_flat.p; i3.bofs = d_bofs.p; i3.n_buckets = nb; i3.key_bits = 64; i3.ent = d_ent.p; i3.hrem = d_hrem.p; i3.val = d_val; i3.counter = d_ctr.p; i3.err = d_err.p;
	hipLaunchKernelGGL(mm_idx_sort_kernel, dim3(std::min<uint32_t>(nb, waves)), dim3(64), 0, 0, i3); IK(hipGetLastError());
	uint32_t serr = 0; IK(hipMemcpy(&serr, d_err.p, 4, hipMemcpyDeviceToHost));
	if(serr) { fprintf(stderr, "[minialign_amd] index (device): a bucket beyond what the sort addresses (flags %u)\n", serr); return false; }
	lap("bucket sort");
	d_flat.release(); d_ent.release();
	/* key counts -> thresholds (minialign.c:2981-2986) */
	DBuf<uint32_t> d_run, d_big; DBuf<unsigned long long> d_h; DBuf<uint64_t> d_cut;
	const uint32_t big_cap = 1u << 20;
	if(!d_run.ensure(N + 64) || !d_big.ensure(big_cap + 4) || !d_h.ensure(IDX_HB + 4) || !d_cut.ensure(nb + 1)) return false;
	IK(hipMemset(d_h.p, 0, (size_t)(IDX_HB + 4) * 8)); IK(hipMemset(d_big.p + big_cap, 0, 16));
	I4Args i4; memset(&i4, 0, sizeof(i4));
	i4.hrem = d_hrem.p; i4.val = d_val; i4.n = N; i4.bofs = d_bofs.p; i4.n_buckets = nb; i4.bbits = bbits; i4.runlen = d_run.p; i4.hist = d_h.p; i4.big = d_big.p; i4.big_cap = big_cap; i4.n_big = d_big.p + big_cap;
	i4.cut = d_cut.p; i4.n_keys = d_h.p + IDX_HB + 2;
	const uint32_t grid_n = (uint32_t)((N + 255) / 256);
	if(N) { hipLaunchKernelGGL(mm_idx_runs_kernel, dim3(grid_n), dim3(256), 0, 0, i4); IK(hipGetLastError()); }
	std::vector<unsigned long long> ch(IDX_HB + 1); uint32_t n_big = 0;
	IK(hipMemcpy(ch.data(), d_h.p, (size_t)(IDX_HB + 1) * 8, hipMemcpyDeviceToHost)); IK(hipMemcpy(&n_big, d_big.p + big_cap, 4, hipMemcpyDeviceToHost));
	if(n_big > big_cap) { fprintf(stderr, "[minialign_amd] index (device): more than %u keys with %u occurrences and more\n", big_cap, IDX_HB); return false; }
	std::vector<uint32_t> big(n_big); if(n_big) IK(hipMemcpy(big.data(), d_big.p, (size_t)n_big * 4, hipMemcpyDeviceToHost));
	uint64_t n_cnt = 0; for(uint32_t c = 0; c <= IDX_HB; c++) n_cnt += ch[c];
	for(uint32_t i = 0; i < o->n_frq; i++) {
		if(o->frq[i] <= 0.0) { mi->occ[i] = UINT32_MAX; continue; }
		if(n_cnt == 0) { mi->occ[i] = 1; continue; }
		const uint32_t kk = (uint32_t)((1.0 - o->frq[i]) * n_cnt);
		const uint64_t kth = std::min<uint64_t>(kk, n_cnt - 1);
		uint64_t acc = 0; uint32_t v = 0; bool found = false;
		for(uint32_t c = 0; c < IDX_HB; c++) { acc += ch[c]; if(acc > kth) { v = c; found = true; break; } }
		if(!found) { auto it = big.begin() + (kth - acc); std::nth_element(big.begin(), it, big.end()); v = *it; }
		mi->occ[i] = v + 1;
	}
	lap("key counts + thresholds");
	/* cut (the reference's fill cursor stops at the first over-frequent key of a bucket), number of keys, table */
	i4.max_cnt = mi->occ[mi->n_occ - 1];
	IK(hipMemcpy(d_cut.p, bofs.data() + 1, (size_t)nb * 8, hipMemcpyHostToDevice));          /* cut[b] = end of the bucket unless a key exceeds the threshold */
	if(N) { hipLaunchKernelGGL(mm_idx_cut_kernel, dim3(grid_n), dim3(256), 0, 0, i4); IK(hipGetLastError()); }
	i4.slot = nullptr;
	if(N) { hipLaunchKernelGGL(mm_idx_fill_kernel, dim3(grid_n), dim3(256), 0, 0, i4); IK(hipGetLastError()); }
	unsigned long long n_keys = 0; IK(hipMemcpy(&n_keys, i4.n_keys, 8, hipMemcpyDeviceToHost));
	uint64_t tsize = 1024; while(tsize < n_keys * 2) tsize <<= 1;
	if(hipMalloc(&mi->d_slot, tsize * sizeof(IdxSlot)) != hipSuccess) return false;
	IK(hipMemset(mi->d_slot, 0, tsize * sizeof(IdxSlot)));
	mi->n_slot = tsize; mi->mask = tsize - 1; mi->n_keys = n_keys;
	i4.slot = mi->d_slot; i4.mask = mi->mask;
	if(N) { hipLaunchKernelGGL(mm_idx_fill_kernel, dim3(grid_n), dim3(256), 0, 0, i4); IK(hipGetLastError()); }
	lap("table");
#undef IK
	mi->on_device = true;
	return true;
}

extern "C" mm_idx_t *mm_idx_gen(mm_opt_t const *o, char const *ref_fasta)
{
	mm_idx_t *mi = new mm_idx_s();
	const bool verbose = getenv("MM_VERBOSE") != NULL; double tv = now_ms();
	auto lap = [&](const char *what) { if(verbose) { double t = now_ms(); fprintf(stderr, "[minialign_amd] index: %s %.1f ms\n", what, t - tv); tv = t; } };
	uint32_t b = std::min(o->k * 2, o->b);
	mi->b = b; mi->w = o->w; mi->k = o->k; mi->n_occ = o->n_frq;
	if(!getenv("MM_HOST_INDEX")) {
		/* the build runs on the device (mm_index.hpp), reference parsing included; MM_HOST_INDEX=1 keeps all of it on the host threads below (machines without a GPU
		 * that only write index files; comparison) */
		if(!idx_gen_device(o, mi, ref_fasta, verbose)) { fprintf(stderr, "[minialign_amd] mm_idx_gen: the device build failed\n"); delete mi; return NULL; }
		return mi;
	}
	if(!read_seq_file(ref_fasta, mi->seq, o->min_len) || mi->seq.empty()) { fprintf(stderr, "[minialign_amd] cannot read reference `%s'\n", ref_fasta); delete mi; return NULL; }
	lap("read + parse");
	if(o->circ_set) for(HSeq &q : mi->seq) q.circular = o->circ_names.empty() || std::find(o->circ_names.begin(), o->circ_names.end(), q.name) != o->circ_names.end();
	const uint64_t nb = 1ull << b, bmask = nb - 1;
	/* sketch every sequence and put (hrem, pos, rid) into its bucket in reference order (minialign.c:2790-2860): stretches of the sequences are
	 * sketched on host threads, a histogram per stretch turns into write positions (stretch order inside a bucket = reference order), and the
	 * stretches scatter their minimizers in parallel into one flat array */
	std::vector<Mini> flat; std::vector<uint64_t> bofs(nb + 1, 0);
	{
		struct Task { uint32_t seq, begin, end; std::vector<HMin> mins; };
		uint64_t total_bases = 0; for(const HSeq &q : mi->seq) total_bases += q.seq.size();
		std::vector<Task> task; const uint32_t chunk = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 18, total_bases / 2048), 1u << 24);
		for(uint32_t i = 0; i < mi->seq.size(); i++) { const uint32_t L = (uint32_t)mi->seq[i].seq.size(); if(mi->seq[i].circular) { task.push_back(Task{ i, 0, L, {} }); continue; } for(uint32_t bgn = 0; bgn < L || bgn == 0; bgn += chunk) { task.push_back(Task{ i, bgn, std::min(L, bgn + chunk), {} }); if(L == 0) break; } }
		std::vector<uint32_t> hist((size_t)task.size() * nb, 0);
		host_parallel((uint32_t)task.size(), [&](uint32_t t, uint32_t nth) {
			for(size_t j = t; j < task.size(); j += nth) {
				Task &q = task[j]; const HSeq &sq = mi->seq[q.seq];
				if(sq.circular) sketch_host_circular(sq.seq.data(), (uint32_t)sq.seq.size(), o->k, o->w, q.mins);
				else sketch_host(sq.seq.data(), (uint32_t)sq.seq.size(), o->k, o->w, q.mins, q.begin, q.end);
				uint32_t *h = &hist[j * nb]; for(const HMin &m : q.mins) h[m.hash & bmask]++;
			}
		}, 64);
		lap("sketch");
		/* hist[j][bi] -> first write position of stretch j in bucket bi (row by row: the rows are contiguous) */
		/* (columns of the stretch x bucket histogram, a range of buckets per thread) */
		host_parallel(64, [&](uint32_t t, uint32_t nth) { for(uint64_t bi = nb * t / nth, be_ = nb * (t + 1) / nth; bi < be_; bi++) { uint64_t acc = 0; for(size_t j = 0; j < task.size(); j++) acc += hist[j * nb + bi]; bofs[bi + 1] = acc; } }, 64);
		for(uint64_t bi = 0; bi < nb; bi++) bofs[bi + 1] += bofs[bi];
		flat.resize(bofs[nb]);
		std::vector<uint64_t> wpos((size_t)task.size() * nb);
		host_parallel(64, [&](uint32_t t, uint32_t nth) { for(uint64_t bi = nb * t / nth, be_ = nb * (t + 1) / nth; bi < be_; bi++) { uint64_t run = bofs[bi]; for(size_t j = 0; j < task.size(); j++) { wpos[j * nb + bi] = run; run += hist[j * nb + bi]; } } }, 64);
		lap("bucket offsets");
		host_parallel((uint32_t)task.size(), [&](uint32_t t, uint32_t nth) {
			for(size_t j = t; j < task.size(); j += nth) {
				Task &q = task[j]; uint64_t *w = &wpos[j * nb];
				for(const HMin &m : q.mins) flat[w[m.hash & bmask]++] = Mini{ m.hash >> b, m.pos, (q.seq << 1) + m.strand };
				std::vector<HMin>().swap(q.mins);
			}
		}, 64);
	}
	lap("bucket scatter");
	/* per-bucket sort on hrem + occurrence histogram (minialign.c:2867-2900); buckets are independent */
	std::vector<uint32_t> cnt;
	{
		std::vector<std::vector<uint32_t>> pc(64);
		host_parallel(64, [&](uint32_t t, uint32_t nth) {
			std::vector<uint32_t> &c = pc[t];
			for(uint64_t bi = t; bi < nb; bi += nth) {
				Mini *v = flat.data() + bofs[bi]; const size_t vn = bofs[bi + 1] - bofs[bi];
				if(vn == 0) continue;
				sort_minis(v, vn);
				uint32_t n = 1;
				for(size_t j = 1; j < vn; j++) { if(v[j - 1].hrem != v[j].hrem) { c.push_back(n); n = 0; } n++; }
				c.push_back(n);
			}
		}, 64);
		for(auto &c : pc) cnt.insert(cnt.end(), c.begin(), c.end());
	}
	lap("bucket sort");
	/* thresholds: (1 - frq)-quantile of the per-key counts, + 1 (minialign.c:2981-2986) */
	{
		/* the k-th smallest count from a histogram of the counts (nearly all are small; the few above the table fall back to selection among themselves) */
		const uint32_t HB = 1u << 16; std::vector<uint64_t> ch(HB + 1, 0); std::vector<uint32_t> big;
		{
			std::vector<std::vector<uint64_t>> ph(32, std::vector<uint64_t>(HB + 1, 0)); std::vector<std::vector<uint32_t>> pb(32);
			host_parallel(32, [&](uint32_t t, uint32_t nth) { for(size_t i = cnt.size() * t / nth, e = cnt.size() * (t + 1) / nth; i < e; i++) { const uint32_t c = cnt[i]; if(c < HB) ph[t][c]++; else { ph[t][HB]++; pb[t].push_back(c); } } }, 32);
			for(auto &h : ph) for(uint32_t c = 0; c <= HB; c++) ch[c] += h[c];
			for(auto &b2 : pb) big.insert(big.end(), b2.begin(), b2.end());
		}
		for(uint32_t i = 0; i < o->n_frq; i++) {
			if(o->frq[i] <= 0.0) { mi->occ[i] = UINT32_MAX; continue; }
			uint32_t kk = (uint32_t)((1.0 - o->frq[i]) * cnt.size());
			if(cnt.empty()) { mi->occ[i] = 1; continue; }
			const size_t kth = std::min<size_t>(kk, cnt.size() - 1);
			uint64_t acc = 0; uint32_t v = 0; bool found = false;
			for(uint32_t c = 0; c < HB; c++) { acc += ch[c]; if(acc > kth) { v = c; found = true; break; } }
			if(!found) { auto it = big.begin() + (kth - acc); std::nth_element(big.begin(), it, big.end()); v = *it; }
			mi->occ[i] = v + 1;
		}
	}
	/* key -> value-list map (minialign.c:2905-2944).  The reference stops advancing its fill cursor at the first key of a
	 * bucket that exceeds the last threshold, which silently drops every later key of that bucket: kept.  Buckets are independent: keys and list
	 * lengths are counted per bucket, a prefix sum gives every bucket its stretch of the value array, and the table is filled by threads that each
	 * own a range of home slots (a key whose probe sequence would leave its owner's range waits for a serial pass), so the layout does not depend
	 * on the number of threads or their timing. */
	const uint64_t max_cnt = mi->occ[mi->n_occ - 1];
	std::vector<uint64_t> bkeys(nb + 1, 0), bvals(nb + 1, 0);
	host_parallel(64, [&](uint32_t t, uint32_t nth) {
		for(uint64_t bi = t; bi < nb; bi += nth) {
			const Mini *v = flat.data() + bofs[bi]; const size_t vn = bofs[bi + 1] - bofs[bi]; uint64_t nk = 0, nv = 0;
			for(size_t j = 0; j < vn;) { size_t e = j + 1; while(e < vn && v[e].hrem == v[j].hrem) e++; if(e - j > max_cnt) break; nk++; if(e - j > 1) nv += e - j; j = e; }
			bkeys[bi + 1] = nk; bvals[bi + 1] = nv;
		}
	}, 64);
	for(uint64_t bi = 0; bi < nb; bi++) { bkeys[bi + 1] += bkeys[bi]; bvals[bi + 1] += bvals[bi]; }
	const uint64_t n_keys = bkeys[nb], n_multi_vals = bvals[nb];
	uint64_t tsize = 1024; while(tsize < n_keys * 2) tsize <<= 1;
	mi->slot.assign(tsize, IdxSlot{ 0, 0 }); mi->mask = tsize - 1; mi->val.assign(std::max<uint64_t>(n_multi_vals, 1), 0); mi->n_keys = n_keys;
	auto hash = [](uint64_t x) { x ^= x >> 31; x *= 0x9e3779b97f4a7c15ull; x ^= x >> 29; return x; };
	/* (key, value) records in bucket order, value lists written in place */
	std::vector<IdxSlot> kv(n_keys);
	host_parallel(64, [&](uint32_t t, uint32_t nth) {
		for(uint64_t bi = t; bi < nb; bi += nth) {
			const Mini *v = flat.data() + bofs[bi]; const size_t vn = bofs[bi + 1] - bofs[bi]; uint64_t ko = bkeys[bi], vo = bvals[bi];
			for(size_t j = 0; j < vn;) {
				size_t e = j + 1; while(e < vn && v[e].hrem == v[j].hrem) e++;
				if(e - j > max_cnt) break;
				uint64_t minier = (v[j].hrem << b) | bi, value;
				if(e - j == 1) value = (uint64_t)v[j].pos | ((uint64_t)v[j].rid << 32);
				else { value = (1ull << 63) | (vo << 24) | (uint64_t)(e - j); for(size_t x = j; x < e; x++) mi->val[vo++] = (uint64_t)v[x].pos | ((uint64_t)v[x].rid << 32); }
				kv[ko++] = IdxSlot{ minier + 1, value };
				j = e;
			}
		}
	}, 64);
	std::vector<Mini>().swap(flat);
	{
		const uint32_t parts = 64; const uint64_t span = tsize / parts;            /* tsize >= 1024: a power of two, divisible */
		std::vector<std::vector<uint64_t>> late(parts);
		/* home slot of every key once (parallel over the keys); the owner of a slot range then walks the one-byte owner column */
		std::vector<uint64_t> home(n_keys); std::vector<uint8_t> owner(n_keys);
		host_parallel(64, [&](uint32_t t, uint32_t nth) { for(uint64_t i = (uint64_t)n_keys * t / nth, e = (uint64_t)n_keys * (t + 1) / nth; i < e; i++) { home[i] = hash(kv[i].key - 1) & mi->mask; owner[i] = (uint8_t)(home[i] / span); } }, 64);
		host_parallel(parts, [&](uint32_t t, uint32_t nth) {
			for(uint32_t pt = t; pt < parts; pt += nth) {
				const uint64_t hi = ((uint64_t)pt + 1) * span;
				for(uint64_t i = 0; i < n_keys; i++) {
					if(owner[i] != pt) continue;
					uint64_t sl = home[i]; while(sl < hi && mi->slot[sl].key != 0) sl++;
					if(sl < hi) mi->slot[sl] = kv[i]; else late[pt].push_back(i);
				}
			}
		}, 64);
		for(auto &l : late) for(uint64_t i : l) { uint64_t sl = hash(kv[i].key - 1) & mi->mask; while(mi->slot[sl].key != 0) sl = (sl + 1) & mi->mask; mi->slot[sl] = kv[i]; }
	}
	if(mi->val.empty()) mi->val.push_back(0);
	lap("thresholds + table");
	return mi;
}
extern "C" void mm_idx_destroy(mm_idx_t *mi) { delete mi; }

/* index files (-d idx.mai, then `minialign idx.mai reads.fa`; mm_idx_dump / mm_idx_load, minialign.c:3070-3167).  The reference's file is a
 * memory image of its own tables with pointers turned into offsets, declared unstable across its releases (README.md:198); this one holds the
 * flattened table the device uses: magic, the parameters, the sequences (name + one byte per base), the slots and the value array.  A file
 * may hold several such blocks back to back (one per reference file given to -d), as the reference's does. */
namespace {
const uint32_t MAI_MAGIC = 0x0341414du;        /* "MAA\x03" */
struct MaiHead { uint32_t b, w, k, n_occ, occ[8]; uint64_t n_seq, n_slot, n_val, n_keys; };
}
extern "C" int mm_idx_dump(mm_idx_t const *mi, FILE *fp)
{
	if(!idx_fetch_host(mi)) return 1;
	bool ok = true;
	auto put = [&](const void *p, size_t n) { ok = ok && (n == 0 || fwrite(p, 1, n, fp) == n); };
	MaiHead h; memset(&h, 0, sizeof(h));
	h.b = mi->b; h.w = mi->w; h.k = mi->k; h.n_occ = mi->n_occ; memcpy(h.occ, mi->occ, sizeof(h.occ)); h.n_seq = mi->seq.size(); h.n_slot = mi->slot.size(); h.n_val = mi->val.size(); h.n_keys = mi->n_keys;
	put(&MAI_MAGIC, 4); put(&h, sizeof(h));
	for(uint32_t i = 0; i < mi->seq.size(); i++) {
		const HSeq &q = mi->seq[i]; const std::vector<uint8_t> &codes = ref_codes(mi, i);
		uint64_t l[3] = { q.name.size(), codes.size(), q.circular ? 1u : 0u };
		put(l, sizeof(l)); put(q.name.data(), l[0]); put(codes.data(), l[1]);
	}
	put(mi->slot.data(), mi->slot.size() * sizeof(IdxSlot)); put(mi->val.data(), mi->val.size() * sizeof(uint64_t));
	return ok && fflush(fp) == 0 ? 0 : 1;
}
/* next block of an index file; NULL at the end of the file (*at_eof = 1) or when the block is damaged / of another version (*at_eof = 0) */
extern "C" mm_idx_t *mm_idx_load(FILE *fp, int *at_eof)
{
	if(at_eof) *at_eof = 0;
	uint32_t magic = 0; size_t got = fread(&magic, 1, 4, fp);
	if(got == 0) { if(at_eof) *at_eof = 1; return NULL; }
	MaiHead h;
	if(got != 4 || magic != MAI_MAGIC || fread(&h, 1, sizeof(h), fp) != sizeof(h)) return NULL;
	if(h.n_occ == 0 || h.n_occ > 7 || h.k < 2 || h.k > 31 || h.w < 1 || h.w > 31 || h.n_slot == 0 || (h.n_slot & (h.n_slot - 1)) || h.n_val == 0) return NULL;
	mm_idx_t *mi = new mm_idx_s();
	mi->b = h.b; mi->w = h.w; mi->k = h.k; mi->n_occ = h.n_occ; memcpy(mi->occ, h.occ, sizeof(h.occ)); mi->n_keys = h.n_keys; mi->mask = h.n_slot - 1;
	bool ok = true;
	auto get = [&](void *p, size_t n) { ok = ok && (n == 0 || fread(p, 1, n, fp) == n); };
	try {
		for(uint64_t i = 0; ok && i < h.n_seq; i++) {
			uint64_t l[3] = { 0, 0, 0 }; get(l, sizeof(l));
			if(!ok || l[0] > (1u << 20) || l[1] > 0xffffffffull) { ok = false; break; }
			mi->seq.emplace_back(); HSeq &q = mi->seq.back();
			q.name.resize(l[0]); q.seq.resize(l[1]); q.circular = l[2] != 0; get(&q.name[0], l[0]); get(q.seq.data(), l[1]);
		}
		if(ok) { mi->slot.resize(h.n_slot); get(mi->slot.data(), h.n_slot * sizeof(IdxSlot)); }
		if(ok) { mi->val.resize(h.n_val); get(mi->val.data(), h.n_val * sizeof(uint64_t)); }
	} catch(std::bad_alloc &) { ok = false; }
	if(!ok || mi->seq.empty()) { delete mi; return NULL; }
	return mi;
}
/* test entry: the packed reference of a device-built index (K0 over the text in HBM: mm_text_codes_tiled_kernel + mm_codes_pack_kernel) brought back and compared base by
 * base with the host's conversion of the same text (ref_codes: the table of minialign.c:223-229).  Returns the number of differing bases (the first few go to stderr),
 * -1 when there is nothing on the device to compare */
extern "C" int64_t mm_idx_ref_check(mm_idx_t const *mi)
{
	if(!mi->on_device || !mi->ref_ar) return -1;
	(void)hipSetDevice(mi->dev);
	const uint64_t n = mi->ref_ar->n, nw = (n + 15) / 16 + 4, nn = (n + 31) / 32 + 4;
	std::vector<uint32_t> pk(nw), nm(nn);
	if(hipMemcpy(pk.data(), mi->ref_ar->pk, nw * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(nm.data(), mi->ref_ar->nm, nn * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	uint64_t off = 0; std::atomic<int64_t> bad{0}; std::mutex pm;
	std::vector<uint64_t> offs; for(const HSeq &q : mi->seq) { offs.push_back(off); off += ((uint64_t)q.blen() + 63) & ~63ull; }
	host_parallel((uint32_t)mi->seq.size(), [&](uint32_t t, uint32_t nth) {
		for(size_t i = t; i < mi->seq.size(); i += nth) {
			const std::vector<uint8_t> &c = ref_codes(mi, (uint32_t)i);
			for(uint64_t j = 0; j < c.size(); j++) {
				const uint64_t p = offs[i] + j; const uint8_t d = ((nm[p >> 5] >> (p & 31)) & 1) ? 4 : (uint8_t)((pk[p >> 4] >> (2 * (p & 15))) & 3);
				if(d != c[j]) { if(bad.fetch_add(1) < 8) { std::lock_guard<std::mutex> lk(pm); fprintf(stderr, "[minialign_amd] reference check: sequence %zu (`%s') base %lu: %u on the device, %u from the text\n", i, mi->seq[i].name.c_str(), (unsigned long)j, d, c[j]); } }
			}
		}
	}, 32);
	return bad.load();
}
extern "C" uint32_t mm_idx_n_seq(mm_idx_t const *mi) { return (uint32_t)mi->seq.size(); }
extern "C" uint32_t mm_idx_occ(mm_idx_t const *mi, uint32_t i) { return mi->occ[i]; }
extern "C" uint32_t mm_idx_max_len(mm_idx_t const *mi) { uint32_t m = 0; for(const HSeq &q : mi->seq) m = std::max<uint32_t>(m, q.blen()); return m; }
extern "C" uint32_t mm_idx_get(mm_idx_t const *mi, uint64_t minier, uint64_t *out, uint32_t max)
{
	auto hash = [](uint64_t x) { x ^= x >> 31; x *= 0x9e3779b97f4a7c15ull; x ^= x >> 29; return x; };
	if(!idx_fetch_host(mi)) return 0;
	uint64_t s = hash(minier) & mi->mask;
	while(mi->slot[s].key != 0) {
		if(mi->slot[s].key == minier + 1) {
			uint64_t v = mi->slot[s].val;
			if((int64_t)v >= 0) { if(max) out[0] = v; return 1; }
			uint32_t n = (uint32_t)(v & 0xffffff); uint64_t off = (v & 0x7fffffffffffffffull) >> 24;
			for(uint32_t i = 0; i < n && i < max; i++) out[i] = mi->val[off + i];
			return n;
		}
		s = (s + 1) & mi->mask;
	}
	return 0;
}

/* mm_sketch (minialign.c:2410-2435) on the host: the (w,k)-minimizer stream of a sequence given one byte per base (0..3, 4 = N), as the words the
 * reference's stream holds -- hash << 8 | strand << 7 | index inside its block of w (minialign.c:2402) -- plus, when pos != NULL, the k-mer start position
 * the stream decoder (minialign.c:2831-2835) gives each word.  Returns the count (at most max are written).  Reads are sketched on the device by K1;
 * this is the entry the index construction uses. */
extern "C" uint32_t mm_sketch(uint8_t const *seq, uint32_t len, uint32_t w, uint32_t k, uint64_t *words, uint32_t *pos, uint32_t max)
{
	if(!seq || k < 2 || k > 31 || w < 1 || w > 31) return 0;
	std::vector<HMin> m; sketch_host(seq, len, k, w, m);
	for(size_t i = 0; i < m.size() && i < max; i++) { if(words) words[i] = m[i].hash << 8 | (uint64_t)m[i].strand << 7 | (m[i].pos % w); if(pos) pos[i] = m[i].pos; }
	return (uint32_t)m.size();
}
