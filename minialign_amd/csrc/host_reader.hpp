/* the text reader: a query file's text to HBM in fixed pieces, records found on the device (K0r), batches cut from them (TextReader) -- part of mm_host.hip (included from there at the place it stood; split out in round 6 so that it can be read on its own) */
/* =============================================================================================
 * the reader: the text of a query file -> records -> batches, with the record scanning on the device (K0r, mm_device.hpp).
 * The host maps the file (or holds what stdin / gzip gave) and brings its bytes to HBM in stretches of 256 MB through pinned staging buffers; a stretch starts
 * where a record starts and is scanned by the kernels, which leave a table of records (delimiter, end of the header line, sequence extent, number of bases);
 * the last, possibly incomplete record of a stretch opens the next one.  The stretches stay in HBM until the batches cut from them have been packed (K0 reads
 * the bases from there), and no base of a read is touched by the host until its record is printed.  FASTQ in any shape other than four lines per record is
 * scanned by the host's sequential reader (host_find_fastq): that grammar -- the number of quality lines depends on the number of bases -- is sequential.
 * ============================================================================================= */
static void free_chunk_pool(struct ChunkPool *p) { delete p; }
namespace {
/* text of a file: a read-only mapping of a plain file (page cache, nothing copied), or memory for stdin and gzip input */
std::shared_ptr<TextSrc> open_text(const char *fn)
{
	auto t = std::make_shared<TextSrc>();
	bool mapped = false;
	if(strcmp(fn, "-") != 0) {
		const int fd = open(fn, O_RDONLY);
		if(fd < 0) return nullptr;
		struct stat sb; uint8_t mg[2] = { 0, 0 };
		if(fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && pread(fd, mg, 2, 0) == 2 && !(mg[0] == 0x1f && mg[1] == 0x8b)) {
			void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
			if(m != MAP_FAILED) { (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL); t->map = m; t->map_len = (uint64_t)sb.st_size; t->p = (const char *)m; t->n = (uint64_t)sb.st_size; mapped = true; }
		}
		close(fd);
	}
	if(!mapped) {
		FILE *fp = strcmp(fn, "-") == 0 ? stdin : fopen(fn, "rb");
		if(!fp) return nullptr;
		std::vector<char> data(1 << 22); size_t len = 0, got;
		while((got = fread(data.data() + len, 1, data.size() - len, fp)) > 0) { len += got; if(len == data.size()) data.resize(data.size() * 2); }
		data.resize(len);
		if(fp != stdin) fclose(fp);
		if(data.size() >= 2 && (uint8_t)data[0] == 0x1f && (uint8_t)data[1] == 0x8b) {          /* gzip members back to back (the reference reads through gzread) */
			std::vector<char> raw(std::max<size_t>(data.size() * 4, 1 << 16));
			z_stream zs; memset(&zs, 0, sizeof(zs));
			if(inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return nullptr;
			zs.next_in = (Bytef *)data.data(); size_t in_left = data.size(), out_len = 0; bool ok = true;
			while(ok) {
				zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30); const size_t in_before = zs.avail_in;
				if(raw.size() - out_len < (1u << 16)) raw.resize(raw.size() * 2);
				zs.next_out = (Bytef *)raw.data() + out_len; zs.avail_out = (uInt)std::min<size_t>(raw.size() - out_len, 1u << 30); const size_t out_before = zs.avail_out;
				const int rc = inflate(&zs, Z_NO_FLUSH);
				in_left -= in_before - zs.avail_in; out_len += out_before - zs.avail_out;
				if(rc == Z_STREAM_END) { if(in_left < 2 || (uint8_t)zs.next_in[0] != 0x1f || (uint8_t)zs.next_in[1] != 0x8b) break; if(inflateReset(&zs) != Z_OK) ok = false; }
				else if(rc != Z_OK && !(rc == Z_BUF_ERROR && zs.avail_out == 0)) ok = false;
				else if(in_left == 0 && zs.avail_out != 0) ok = false;
			}
			inflateEnd(&zs);
			if(!ok) { fprintf(stderr, "[minialign_amd] broken gzip stream in `%s'\n", fn); return nullptr; }
			raw.resize(out_len); data.swap(raw);
		}
		t->own.swap(data); t->p = t->own.data(); t->n = t->own.size();
	}
	/* the file type is the first '>' or '@' among the first four bytes; what stands in front of it is dropped (minialign.c:1784-1792) */
	for(int i = 0; i < 4 && t->first < t->n; i++) { if(t->p[t->first] == '>' || t->p[t->first] == '@') { t->delim = t->p[t->first]; break; } t->first++; }
	if(!t->delim) { fprintf(stderr, "[minialign_amd] `%s' is neither FASTA nor FASTQ\n", fn); return nullptr; }
	return t;
}
/* FASTQ records of text[0, n) one after the other, as parse_fastq reads them, offsets only (relative to t).  A record the text ends in (last == false: the next
 * stretch brings the rest) is left out and *consumed stops in front of it.  false when a record does not start with '@' where one must (the reference gives up). */
bool host_find_fastq(const char *t, uint64_t n, bool last, bool keep_qual, std::vector<RRec> &out, uint64_t &consumed)
{
	const char *p = t, *end = t + n;
	consumed = 0;
	while(p < end) {
		const char *rs = p;
		if(*p++ != '@') return false;
		RRec r; memset(&r, 0, sizeof(r)); r.start = (uint64_t)(rs - t);
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		if(!nl) { if(!last) break; r.hdr_end = n; r.t_off = n; out.push_back(r); p = end; consumed = n; break; }
		r.hdr_end = (uint64_t)(nl - t); p = nl + 1;
		r.t_off = (uint64_t)(p - t); const char *t_end = p; uint64_t nb = 0; bool at = false;
		while(p < end) {
			nl = (const char *)memchr(p, '\n', (size_t)(end - p)); const char *le = nl ? nl : end;
			const char *dl = (const char *)memchr(p, '+', (size_t)(le - p)); const char *stop = dl ? dl : le;
			if(stop > p) { nb += (uint64_t)(stop - p); t_end = stop; }
			if(dl) { at = true; p = dl; break; }
			p = nl ? nl + 1 : end;
		}
		r.t_len = (uint32_t)((uint64_t)(t_end - t) - r.t_off); r.n_bases = (uint32_t)nb;
		if(!at) { if(!last) break; out.push_back(r); consumed = n; p = end; break; }
		nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		if(!nl && !last) break;
		p = nl ? nl + 1 : end;
		r.q_off = (uint64_t)(p - t); uint64_t acc = 0;
		while(p < end) {
			nl = (const char *)memchr(p, '\n', (size_t)(end - p)); const char *le = nl ? nl : end; size_t ll = (size_t)(le - p);
			if(keep_qual) { if(ll > 0 && p[ll - 1] == '\r') ll--; }
			acc += ll; p = le;
			if(p >= end || acc >= nb) break;
			p++;
		}
		r.q_len = (uint32_t)((uint64_t)(p - t) - r.q_off);
		if(!last && p >= end) break;          /* the quality line may go on in the next stretch */
		out.push_back(r);
		while(p < end && *p == '\n') p++;
		consumed = (uint64_t)(p - t);
	}
	return true;
}
/* host threads that stay for the life of a reader: a piece of text is copied into a pinned staging buffer by all of them, a slice each (threads made per 32 MB piece
 * cost more than the copy) */
struct CopyPool {
	std::vector<std::thread> th; std::mutex mu; std::condition_variable cv, dcv;
	const char *src = nullptr; char *dst = nullptr; size_t n = 0; uint64_t gen = 0; uint32_t left = 0, nth = 0; bool stop = false;
	void start(uint32_t want) { nth = want; for(uint32_t t = 0; t < nth; t++) th.emplace_back([this, t]() { run(t); }); }
	void run(uint32_t t)
	{
		uint64_t seen = 0;
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
		if(nth == 0 || bytes < (1u << 20)) { memcpy(d, sp, bytes); return; }
		std::unique_lock<std::mutex> lk(mu); src = sp; dst = d; n = bytes; left = nth; gen++; cv.notify_all();
		dcv.wait(lk, [&]() { return left == 0; });
	}
	~CopyPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for(auto &t : th) t.join(); }
};
/* the reader's side of ONE device: the way of the text into its HBM (pinned staging ring + copy threads + an upload stream) and the record scan of a stretch there */
struct ReaderDev {
	int dev = 0; ChunkPool *pool = nullptr; bool fastq = false, keep_qual = false;
	hipStream_t st = nullptr, up = nullptr;          /* scan stream; upload stream */
	static const int RING = 4; void *pin[RING] = { nullptr, nullptr, nullptr, nullptr }; hipEvent_t pev[RING] = { nullptr, nullptr, nullptr, nullptr }; size_t pin_cap = 32u << 20; uint64_t pin_k = 0;
	CopyPool cp;
	DBuf<uint64_t> d_ma, d_mb; DBuf<uint32_t> d_blk, d_pos, d_cum, d_flag; DBuf<TextRec> d_rec;
	uint64_t chunk_bytes = 256ull << 20;          /* what the scan's scratch arrays are sized for from the start */
	uint64_t n_host_scanned = 0, bytes_up = 0; double t_io = 0, t_scan = 0;
	~ReaderDev()
	{
		if(st || up) (void)hipSetDevice(dev);
		if(st) (void)hipStreamDestroy(st); if(up) (void)hipStreamDestroy(up);
		for(int i = 0; i < RING; i++) { if(pin[i]) (void)hipHostFree(pin[i]); if(pev[i]) (void)hipEventDestroy(pev[i]); }
		d_ma.release(); d_mb.release(); d_blk.release(); d_pos.release(); d_cum.release(); d_flag.release(); d_rec.release();
	}
	/* streams, staging ring, copy threads; the calling thread is on the device */
	bool init(uint32_t copy_threads)
	{
		if(hipGetDevice(&dev) != hipSuccess) return false;
		if(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) return false;
		for(int i = 0; i < RING; i++) { if(hipHostMalloc(&pin[i], pin_cap, hipHostMallocPortable) != hipSuccess || hipEventCreateWithFlags(&pev[i], hipEventDisableTiming) != hipSuccess) return false; }
		cp.start(copy_threads);
		return true;
	}
	/* host text -> pinned staging (the copy threads) -> HBM, 32 MB at a time on stream q; returns when the last piece has been queued (the caller waits for q) */
	std::mutex up_mu;          /* (the uploader thread and the scan's slow way share the ring) */
	bool upload(uint8_t *dst, const char *sp, uint64_t len, hipStream_t q)
	{
		const double t0 = now_ms();
		for(uint64_t o = 0; o < len; o += pin_cap) {
			std::lock_guard<std::mutex> lk(up_mu);
			const size_t nb = (size_t)std::min<uint64_t>(pin_cap, len - o); const int pi = (int)(pin_k++ % RING);
			CK(hipEventSynchronize(pev[pi]));
			cp.copy((char *)pin[pi], sp + o, nb);
			CK(hipMemcpyAsync(dst + o, pin[pi], nb, hipMemcpyHostToDevice, q));
			CK(hipEventRecord(pev[pi], q));
		}
		bytes_up += len; t_io += now_ms() - t0;
		return true;
	}
	/* records of the stretch text[at, at + len) (host copy: tx), which stands in HBM at base + skip (base 64-byte aligned, skip < 64).  Offsets come back absolute;
	 * consumed: bytes of the stretch up to where the next one starts (all of them at the end of the text); grow: not one complete record in it */
	bool scan(const char *tx, const uint8_t *base, uint32_t skip, uint64_t at, uint64_t len, bool last, std::vector<RRec> &recs, uint64_t &consumed, bool &grow)
	{
		grow = false; consumed = 0; recs.clear();
		const double t1 = now_ms();
		const uint32_t n = (uint32_t)(len + skip), n_words = (n + 63) / 64, n_blk = (n_words + 255) / 256;
		/* the scratch arrays are sized for a whole stretch from the start: a buffer that grows in mid-run costs a hipFree, which waits for every stream of the device */
		const uint64_t cap_n = std::max<uint64_t>(len + 64, chunk_bytes + 64), cap_words = (cap_n + 63) / 64, cap_blk = (cap_words + 255) / 256;
		const uint32_t pos_cap = (uint32_t)(cap_n / 8 + 1024);
		if(!d_ma.ensure(cap_words) || !d_mb.ensure(cap_words) || !d_cum.ensure(cap_words) || !d_blk.ensure(2 * cap_blk + 2) || !d_pos.ensure(pos_cap) || !d_flag.ensure(4)) return false;
		ScanArgs sa; memset(&sa, 0, sizeof(sa));
		sa.text = base; sa.n = n; sa.skip = skip; sa.fastq = fastq ? 1u : 0u; sa.ma = d_ma.p; sa.mb = d_mb.p; sa.blk = d_blk.p; sa.n_blk = n_blk; sa.pos = d_pos.p; sa.pos_cap = pos_cap; sa.cum = d_cum.p;
		sa.last = last ? 1u : 0u; sa.keep_qual = keep_qual ? 1u : 0u; sa.flag = d_flag.p;
		CK(hipMemsetAsync(d_flag.p, 0, 16, st));
		if(fastq) { CK(hipMemcpyAsync(d_pos.p, &skip, 4, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }          /* the first line starts where the stretch starts */
		hipLaunchKernelGGL(mm_text_marks_kernel, dim3(n_blk), dim3(256), 0, st, sa); CK(hipGetLastError());
		hipLaunchKernelGGL(mm_text_blocks_kernel, dim3(1), dim3(256), 0, st, sa); CK(hipGetLastError());
		hipLaunchKernelGGL(mm_text_emit_kernel, dim3(n_blk), dim3(256), 0, st, sa); CK(hipGetLastError());
		uint32_t tot[2], flag[4];
		CK(hipMemcpyAsync(tot, d_blk.p + 2 * (uint64_t)n_blk, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(flag, d_flag.p, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
		bool on_host = fastq && flag[1] != 0;
		uint32_t n_rec = 0;
		if(!fastq) {
			if(flag[1]) { fprintf(stderr, "[minialign_amd] reader: more record starts than one per 8 bytes in a stretch of `>' records\n"); return false; }
			const uint32_t n_starts = tot[1];
			n_rec = last ? n_starts : (n_starts ? n_starts - 1 : 0);
			if(!last && n_rec == 0) { grow = true; t_scan += now_ms() - t1; return true; }
			if(!last) { uint32_t q = 0; CK(hipMemcpyAsync(&q, d_pos.p + (n_starts - 1), 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); consumed = q - skip; } else consumed = len;
		} else if(!on_host) {
			/* complete lines; at the end of the text a last line without '\n' counts, and lines left over behind the last record must be empty (the reference skips them) */
			uint64_t lines = tot[0] + ((last && len > 0 && tx[len - 1] != '\n') ? 1u : 0u);
			n_rec = (uint32_t)(lines / 4);
			if(!last && n_rec == 0) { grow = true; t_scan += now_ms() - t1; return true; }
			if(!last) { uint32_t q = 0; CK(hipMemcpyAsync(&q, d_pos.p + 4 * (uint64_t)n_rec, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); consumed = q - skip; }
			else {
				consumed = len;
				if(lines % 4) { std::vector<uint32_t> ls(lines % 4 + 1, (uint32_t)(len + skip)); CK(hipMemcpyAsync(ls.data(), d_pos.p + 4 * (uint64_t)n_rec, (lines % 4) * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
					for(uint64_t q = ls[0] - skip; q < len; q++) if(tx[q] != '\n') { on_host = true; break; } }
			}
		}
		if(!on_host && n_rec) {
			if(!d_rec.ensure(std::max<uint64_t>(n_rec, d_rec.n ? 0 : chunk_bytes / 4096))) return false;          /* (room for reads of 4 kb and more from the start) */
			sa.rec = d_rec.p; sa.n_rec = n_rec;
			if(fastq) hipLaunchKernelGGL(mm_text_fastq_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, st, sa); else hipLaunchKernelGGL(mm_text_fasta_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, st, sa);
			CK(hipGetLastError());
			std::vector<TextRec> tr(n_rec);
			CK(hipMemcpyAsync(tr.data(), d_rec.p, (size_t)n_rec * sizeof(TextRec), hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(flag, d_flag.p, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			if(fastq && flag[0]) on_host = true;
			else { const uint64_t o = at - skip; recs.resize(n_rec); for(uint32_t i = 0; i < n_rec; i++) { const TextRec &q = tr[i]; recs[i] = RRec{ o + q.start, o + q.hdr_end, o + q.t_off, q.t_len, q.n_bases, o + q.q_off, q.q_len }; } }
		}
		if(on_host) {
			/* a FASTQ stretch in another shape than four lines per record: the sequential reader, over the host's copy of the same bytes */
			recs.clear(); uint64_t used = 0;
			if(!host_find_fastq(tx, len, last, keep_qual, recs, used)) { fprintf(stderr, "[minialign_amd] broken FASTQ record\n"); return false; }          /* the reference gives up on the run (exit 1) */
			if(!last && recs.empty()) { grow = true; t_scan += now_ms() - t1; return true; }
			for(RRec &q : recs) { q.start += at; q.hdr_end += at; q.t_off += at; q.q_off += at; }
			consumed = last ? len : used; n_host_scanned += recs.size();
		}
		t_scan += now_ms() - t1;
		return true;
	}
};
/* The reader of a query text over the devices of a context.  The text is cut into PIECES at fixed byte offsets (the first ones 64 MB, then 256 MB; piece s goes to device
 * s mod N), so every device's uploader thread brings its pieces to HBM without waiting for anybody -- record boundaries are found afterwards, by the scan, in order:
 * the stretch that piece s closes starts where the scan of the stretch before it stopped (the record that was cut by the boundary), and those few bytes -- the tail
 * of piece s - 1 -- are put in FRONT of piece s in its buffer (every buffer keeps `prefix` bytes of room there), so that the stretch is contiguous in the HBM of the
 * device that scans and packs it.  The sequential part of the reader is thereby the scan alone (four short launches and a few words of D2H per stretch); the uploads
 * run side by side, one PCIe link each.  A record longer than the room in front (or than a piece) takes the slow way: its stretch goes up again as a whole.
 * Batches are cut from the records as before; with several devices a batch never holds reads of two stretches on different devices. */
struct TextReader {
	std::shared_ptr<TextSrc> src; uint32_t min_len = 1; bool keep_qual = false; int lanes = 4;
	std::vector<mm_align_t *> dctx;          /* the device slots of the engine: one primary context each (set by the caller) */
	std::vector<ReaderDev *> rdev;
	uint64_t chunk_bytes = 256ull << 20, prefix = 1ull << 20;
	std::vector<uint64_t> B;          /* piece s = text[B[s], B[s + 1]) */
	struct PieceSt { DevChunk *c = nullptr; int state = 0; };          /* 0 untouched, 1 in HBM, 2 failed, 3 not wanted (its stretch went the slow way), 4 on its way */
	std::vector<PieceSt> pieces;
	std::mutex mu; std::condition_variable cv;
	uint32_t scanned = 0;          /* pieces the scan is done with: the uploaders stay at most `ahead` pieces per device in front of it */
	static const uint32_t ahead = 2;
	std::vector<std::deque<mm_batch_t *>> ready; uint32_t n_cut = 0; bool done = false, failed = false, stop = false;
	uint64_t max_bases = 300000000ull, cap_bases = 1000000000ull; uint32_t longest = 0; bool repeat_rich = false;          /* cap_bases: what the device memory allows (batch_cap_bases); repeat_rich: see start() */
	std::vector<std::thread> th;
	mm_batch_t *cur = nullptr; uint64_t cur_bases = 0; int cur_slot = 0;
	uint64_t n_records = 0, n_host_scanned = 0, n_stretches = 0, n_slow = 0; double t_start = 0;

	~TextReader()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all();
		for(auto &t : th) if(t.joinable()) t.join();
		for(auto &q : ready) for(mm_batch_t *h : q) delete h;
		delete cur;
		for(size_t i = 0; i < pieces.size(); i++) if(pieces[i].c) { rdev[i % rdev.size()]->pool->put(pieces[i].c); pieces[i].c = nullptr; }
		for(ReaderDev *r : rdev) delete r;
	}
	void push_batch()
	{
		if(!cur) return;
		mm_batch_t *h = cur; cur = nullptr; cur_bases = 0;
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&]() { return stop || ready[cur_slot].size() < (size_t)lanes + 2; });          /* not further ahead of the lanes of that device than this */
		if(stop) { delete h; return; }
		h->k = n_cut++; ready[cur_slot].push_back(h); lk.unlock(); cv.notify_all();
	}
	void add(const RRec &r, const std::shared_ptr<DevChunk> &ch, int slot)
	{
		if(r.n_bases < min_len) return;          /* -L (minialign.c:2077) */
		if(r.n_bases > longest) { longest = r.n_bases; if(!getenv("MM_BATCH_BASES")) max_bases = std::max<uint64_t>(max_bases, std::min<uint64_t>(cap_bases, (uint64_t)longest * MM_BATCH_PER_LONGEST)); }
		if(cur && (slot != cur_slot || cur->b.lens.size() >= (1u << 17) || (cur_bases && cur_bases + r.n_bases > max_bases))) push_batch();
		if(!cur) { cur = new mm_batch_s(); cur->b.tsrc = src; cur_slot = slot; }
		Batch &b = cur->b;
		if(b.dch.empty() || b.dch.back().ch != ch) b.dch.push_back(Batch::Piece{ ch, (uint32_t)b.lens.size(), 0 });
		b.dch.back().n++; b.lens.push_back(r.n_bases); b.trec.push_back(r); cur_bases += r.n_bases;
	}
	/* the uploader of device slot di: its pieces in order, each into a buffer with `prefix` bytes of room in front */
	void upload_main(int di)
	{
		ReaderDev *R = rdev[di]; const uint32_t nd = (uint32_t)rdev.size();
		bool ok = hipSetDevice(R->dev) == hipSuccess;
		for(uint32_t s = (uint32_t)di; s + 1 < B.size(); s += nd) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return stop || s < scanned + ahead * nd; });
				if(stop) return;
				if(pieces[s].state == 3) continue;
				pieces[s].state = 4;
			}
			const uint64_t len = B[s + 1] - B[s];
			DevChunk *c = ok ? R->pool->get(prefix + std::max<uint64_t>(chunk_bytes, (len + 63) & ~63ull) + 128) : nullptr;
			bool up = c != nullptr;
			if(up) { c->off = B[s] - prefix; c->n = 0; up = R->upload(c->d + prefix, src->p + B[s], len, R->up) && hipStreamSynchronize(R->up) == hipSuccess; }
			{ std::lock_guard<std::mutex> lk(mu); pieces[s].c = c; pieces[s].state = up ? 1 : 2; }
			cv.notify_all();
		}
	}
	/* waits for piece s to be in HBM (or settled otherwise); its state */
	int piece_wait(uint32_t s) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return stop || (pieces[s].state != 0 && pieces[s].state != 4); }); return stop ? 2 : pieces[s].state; }
	void run()
	{
		const uint32_t nd = (uint32_t)rdev.size(), n_pieces = (uint32_t)B.size() - 1;
		bool ok = true; uint64_t at = src->first; uint32_t s = 0;
		while(ok && at < src->n && s < n_pieces) {
			{ std::lock_guard<std::mutex> lk(mu); if(stop) break; }
			uint32_t e = s + 1, span = 1; int slot = 0;
			std::vector<RRec> recs; uint64_t consumed = 0; std::shared_ptr<DevChunk> ch;
			while(ok) {
				const uint64_t end = B[e], len = end - at; const bool last = end == src->n; bool grow = false;
				if(len + 64 > 0x7fff0000ull) { fprintf(stderr, "[minialign_amd] reader: a record of more than 2 GB\n"); ok = false; break; }
				if(e == s + 1 && B[s] - at <= prefix) {
					/* the common way: piece s is (being) brought up by its device's uploader; the bytes of the record its boundary cut go in front of it */
					slot = (int)(s % nd); ReaderDev *R = rdev[slot];
					if(piece_wait(s) != 1 || hipSetDevice(R->dev) != hipSuccess) { ok = false; break; }
					DevChunk *c = pieces[s].c; const uint64_t tail = B[s] - at;
					if(tail && (hipMemcpyAsync(c->d + prefix - tail, src->p + at, tail, hipMemcpyHostToDevice, R->st) != hipSuccess || hipStreamSynchronize(R->st) != hipSuccess)) { ok = false; break; }
					const uint64_t o = prefix - tail;
					ok = R->scan(src->p + at, c->d + (o & ~63ull), (uint32_t)(o & 63), at, len, last, recs, consumed, grow);
					if(ok && !grow) { pieces[s].c = nullptr; ChunkPool *pl = R->pool; ch = std::shared_ptr<DevChunk>(c, [pl](DevChunk *q) { pl->put(q); }); }
				} else {
					/* the slow way (a record longer than the room in front of a piece, or than a piece): what the uploaders brought or will bring of pieces s .. e - 1 is not
					 * wanted, the stretch goes up as a whole on the device of its last piece */
					for(uint32_t j = s; j < e; j++) {
						{ std::lock_guard<std::mutex> lk(mu); if(pieces[j].state == 0) { pieces[j].state = 3; continue; } if(pieces[j].state == 3) continue; }
						(void)piece_wait(j);
						if(pieces[j].c) { rdev[j % nd]->pool->put(pieces[j].c); pieces[j].c = nullptr; }
					}
					slot = (int)((e - 1) % nd); ReaderDev *R = rdev[slot]; n_slow++;
					if(hipSetDevice(R->dev) != hipSuccess) { ok = false; break; }
					DevChunk *c = R->pool->get(((len + 63) & ~63ull) + 128);
					if(!c) { ok = false; break; }
					c->off = at; c->n = 0;
					ok = R->upload(c->d, src->p + at, len, R->st) && hipStreamSynchronize(R->st) == hipSuccess && R->scan(src->p + at, c->d, 0, at, len, last, recs, consumed, grow);
					if(ok && !grow) { ChunkPool *pl = R->pool; ch = std::shared_ptr<DevChunk>(c, [pl](DevChunk *q) { pl->put(q); }); } else { R->pool->put(c); }
				}
				if(!ok || !grow) break;
				if(e == n_pieces) { ok = false; break; }          /* (cannot happen: the stretch that ends the text is complete by definition) */
				e = std::min<uint32_t>(n_pieces, e + span); span *= 2;          /* a record longer than the stretch: more pieces */
			}
			if(!ok) break;
			n_stretches++; n_records += recs.size();
			for(const RRec &r : recs) add(r, ch, slot);
			if(consumed == 0) { ok = false; break; }
			at += consumed; s = e;
			{ std::lock_guard<std::mutex> lk(mu); scanned = s; } cv.notify_all();
			if(nd > 1) push_batch();          /* several devices: the next stretch is another device's */
		}
		if(ok) push_batch();
		{ std::lock_guard<std::mutex> lk(mu); done = true; failed = !ok; }
		cv.notify_all();
		for(ReaderDev *R : rdev) n_host_scanned += R->n_host_scanned;
		if(getenv("MM_VERBOSE")) {
			double io = 0, sc = 0; uint64_t up = 0; for(ReaderDev *R : rdev) { io += R->t_io; sc += R->t_scan; up += R->bytes_up; }
			fprintf(stderr, "[minialign_amd] reader: %lu records in %lu stretches on %u device(s) (%lu the slow way; %lu records through the host's sequential FASTQ reader), %.2f GB of text to HBM in %.1f ms of uploader time (%.1f GB/s per uploader), scans %.1f ms, done %.1f ms after the start\n",
				(unsigned long)n_records, (unsigned long)n_stretches, nd, (unsigned long)n_slow, (unsigned long)n_host_scanned, up * 1e-9, io, io > 0 ? up * 1e-6 / io : 0.0, sc, now_ms() - t_start);
		}
	}
	bool start()
	{
		if(dctx.empty()) return false;
		if(const char *e = getenv("MM_CHUNK_BYTES")) chunk_bytes = std::max<uint64_t>(64, (uint64_t)atoll(e)) & ~63ull;          /* test hook: small stretches */
		prefix = std::min<uint64_t>(1ull << 20, chunk_bytes);
		const uint32_t nd = (uint32_t)dctx.size();
		/* copy threads per device: the staging copy wants a handful of cores (a core copies 5 - 10 GB/s; PCIe takes 50) */
		const uint32_t hw = std::max<uint32_t>(1, std::thread::hardware_concurrency());
		const uint32_t cpt = std::max<uint32_t>(2, std::min<uint32_t>(12, hw / (4 * nd)));
		int cur_dev = 0; (void)hipGetDevice(&cur_dev);
		for(uint32_t d = 0; d < nd; d++) {
			mm_align_t *P = dctx[d];
			if(!P->chunk_pool) P->chunk_pool = new ChunkPool();
			ReaderDev *R = new ReaderDev(); rdev.push_back(R);
			R->pool = P->chunk_pool; R->fastq = src->delim == '@'; R->keep_qual = keep_qual; R->chunk_bytes = chunk_bytes + prefix;
			if(hipSetDevice(P->dev) != hipSuccess || !R->init(cpt)) { (void)hipSetDevice(cur_dev); return false; }
		}
		(void)hipSetDevice(cur_dev);
		ready.resize(nd);
		/* batch size as batch_spans: 300 Mb; a text smaller than lanes x that (per device) is cut into one and a half batches per lane (its bases are a little fewer than its bytes) */
		const uint64_t all_lanes = (uint64_t)lanes * nd;
		if(getenv("MM_BATCH_BASES")) max_bases = (uint64_t)atoll(getenv("MM_BATCH_BASES"));
		else if(src->n < all_lanes * max_bases) max_bases = std::max<uint64_t>(64ull << 20, src->n / (all_lanes + all_lanes / 2) + (1ull << 20));          /* (an eighth of the headline set maps in 313 ms in six batches, 366 in four) */
		/* ... and a text of more than five batches per lane gets larger ones, up to 500 Mb: every batch ends in the tail of its extension launch, and since the first read of a batch
		 * starts from the value the batch in front predicts (PredBoard) nothing is gained from many small ones -- the headline set: 4.44 / 4.54 G bases/s at 300 Mb, 4.59 / 4.60 at
		 * 400, 4.65 / 4.68 at 500, 4.59 at 600 (profiles/round6_batch_size.txt) */
		else {
			/* ... sooner on a repeat-rich reference (`repeat_rich`: the last occurrence threshold of the index, what -f 0.001 makes of the minimizer counts, at 64 and more -- 113 on the
			 * hard-repeat human-size reference, 24 on the headline one with its 5 % of repeats, 12 and 10 on the dm6- and E.coli-size ones): there a launch lasts as long as the
			 * read with the most chains, every batch pays that tail once, and seven batches of 443 Mb map the 3.1 Gb of the hard-repeat set at 2.03 - 2.10 G bases/s where eleven
			 * of 300 Mb make 1.72 (600 Mb: 1.91, 150 Mb: 1.26); the dm6-size set, whose launches have no such tail, loses 4 % with seven batches instead of ten */
			const uint64_t est_bases = src->delim == '@' ? src->n / 2 : src->n, per = repeat_rich ? 2 * all_lanes - 1 : all_lanes * 5;
			max_bases = std::max<uint64_t>(max_bases, std::min<uint64_t>(std::min<uint64_t>(500000000ull, std::max<uint64_t>(max_bases, cap_bases)), est_bases / per));
		}
		/* pieces: the first three per device short (64 MB), so that the first lanes have a batch to work on early; several devices: no longer than a batch */
		uint64_t first_len = std::min<uint64_t>(chunk_bytes, 64ull << 20), later_len = chunk_bytes;
		if(nd > 1 && !getenv("MM_CHUNK_BYTES")) { later_len = std::min<uint64_t>(later_len, std::max<uint64_t>(1ull << 20, max_bases)); first_len = std::min(first_len, later_len); }
		B.push_back(src->first);
		while(B.back() < src->n) { const uint64_t ln = (B.size() - 1 < 3ull * nd) ? first_len : later_len; B.push_back(std::min<uint64_t>(src->n, B.back() + ln)); }
		pieces.assign(B.size() - 1, PieceSt());
		t_start = now_ms();
		for(uint32_t d = 0; d < nd; d++) th.emplace_back([this, d]() { upload_main((int)d); });
		th.emplace_back([this]() { run(); });
		return true;
	}
	/* the next batch of device slot di (lanes of a device ask in order); its number in the order of the text in h->k; NULL when the text has no more for this device,
	 * *err when the reader failed */
	mm_batch_t *take(int di, bool *err)
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&]() { return done || !ready[di].empty(); });
		if(ready[di].empty()) { if(failed && err) *err = true; return nullptr; }
		mm_batch_t *h = ready[di].front(); ready[di].pop_front();
		lk.unlock(); cv.notify_all();
		batch_pack(h->b, false);
		return h;
	}
};
} /* anonymous */
