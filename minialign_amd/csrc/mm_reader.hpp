/* K0r / K0: records from the raw text of a read file in HBM, base conversion and 2-bit packing (bseq_read_fasta, minialign.c:1996-2090) -- part of mm_device.hpp (included from there, inside namespace mm; split out in round 6 so that each stage can be read on its own) */
/* =====================================================================================================
 * K0: reads from their text.  bseq_read_fasta's base conversion (minialign.c:1996-2090 with the table encaf, :223-229: the low nibble of the byte picks
 * A / a -> 0, C / c -> 1, G / g -> 2, T / t / U / u -> 3, N / n -> 4 and EVERY other byte -> 0) and the 2-bit packing, from the raw text of the file in HBM:
 * the host parser only finds where each record's sequence lines begin and end; every byte of that extent except '\n' is a base (a CR too).
 *   mm_text_codes_kernel   wave per read: 64 text bytes at a time, newlines squeezed out by ballot + popcount, one code byte per base into the arena image
 *   mm_codes_pack_kernel   thread per 32 bases of the arena: two 2-bit words and one N-mask word (what pack_bases builds on the host)
 * ===================================================================================================== */
struct TextRead { uint64_t t_off; uint32_t t_len; uint32_t pad; uint64_t q_off; };      /* extent in the uploaded text, first base in the arena */
/* register budget of the short kernels (sketch, sort, chain sweep): 64 VGPRs, the extension kernel's own -- they start in the wave slots that extension waves of
 * the other lanes leave, and a slot left by a 64-VGPR wave holds nothing larger (at 74 / 75 VGPRs the sketch and the sweep had to wait for two to come free on one SIMD) */
#ifndef MM_SHORT_KERNEL_WAVES
#define MM_SHORT_KERNEL_WAVES 8
#endif
__global__ void __launch_bounds__(256) mm_text_codes_kernel(const uint8_t *text, const TextRead *tr, uint32_t n_reads, uint8_t *codes, uint32_t *n_bases)
{
	const int lane = lane_id();
	const uint32_t r = (uint32_t)rdfirst((int)(blockIdx.x * 4 + threadIdx.x / 64));
	if(r >= n_reads) { return; }
	const uint64_t t0 = rdfirst64(tr[r].t_off), q0 = rdfirst64(tr[r].q_off); const uint32_t tl = (uint32_t)rdfirst((int)tr[r].t_len);
	const uint64_t lut = 0x0400000020331000ull;          /* 4 bits per low nibble: 'A' & 15 = 1 -> 0, 'C' = 3 -> 1, 'T' = 4 -> 3, 'U' = 5 -> 3, 'G' = 7 -> 2, 'N' = 14 -> 4, everything else 0 */
	uint32_t out = 0;
	for(uint32_t i0 = 0; i0 < tl; i0 += 64) {
		const uint32_t i = i0 + (uint32_t)lane;
		const uint8_t c = i < tl ? text[t0 + i] : (uint8_t)'\n';
		const bool keep = c != (uint8_t)'\n';
		const uint64_t m = __ballot(keep);
		if(keep) { codes[q0 + out + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint8_t)((lut >> (4 * (c & 15))) & 15); }
		out += (uint32_t)__popcll(m);
	}
	if(lane == 0) { n_bases[r] = out; }
}
__global__ void __launch_bounds__(256) mm_codes_pack_kernel(const uint8_t *codes, uint64_t n_words32, uint32_t *pk, uint32_t *nm)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;          /* one N-mask word = 32 bases = two 2-bit words */
	if(w >= n_words32) { return; }
	const uint4 c0 = ((const uint4 *)codes)[2 * w], c1 = ((const uint4 *)codes)[2 * w + 1];
	const uint32_t cw[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
	uint32_t w0 = 0, w1 = 0, m = 0;
	for(int j = 0; j < 32; j++) {
		const uint32_t c = (cw[j >> 2] >> (8 * (j & 3))) & 0xffu;
		const uint32_t two = c <= 3 ? c : 0u;
		if(j < 16) { w0 |= two << (2 * j); } else { w1 |= two << (2 * (j - 16)); }
		m |= (uint32_t)(c > 3) << j;
	}
	pk[2 * w] = w0; pk[2 * w + 1] = w1; nm[w] = m;
}

/* =====================================================================================================
 * K0r: records from text.  bseq_read_fasta's record scanning (minialign.c:1996-2090) over a stretch of the raw text of the file in HBM -- the host only brings
 * the bytes (mmap / gunzip).  A stretch starts where a record starts and is scanned in four launches:
 *   mm_text_marks_kernel    thread per 64 bytes: the '\n' mask and, FASTA, the mask of record starts -- a '>' starts a record iff it is the first '>' of its line
 *                           (the reference's reader looks for its delimiter ANYWHERE in a sequence line, and the rest of that line is the header, whatever it
 *                           holds) -- or, FASTQ, the mask of '+' bytes; per-block totals
 *   mm_text_blocks_kernel   one block: exclusive scan of the per-block totals
 *   mm_text_emit_kernel     thread per 64 bytes: FASTA the positions of the record starts in order and the number of newlines in front of every 64-byte word; FASTQ the
 *                           line starts in order and the number of '+' bytes in front of every word
 *   mm_text_fasta_kernel    thread per record: where its header line ends, the extent of its sequence lines (up to the next record start; the last record of a
 *                           stretch that is not the end of the file is left to the next stretch) and its number of bases = bytes of the extent that are not '\n'
 *   mm_text_fastq_kernel    thread per record, for the common shape of four lines per record -- '@' header, one sequence line without a '+', a line that starts with
 *                           '+', one quality line at least as long as the sequence -- checked line by line; anything else (wrapped records, a '+' inside the
 *                           bases, qualities that run short or long, blank lines) raises a flag and the stretch goes through the host's sequential reader, which is
 *                           what that grammar is
 * ===================================================================================================== */
struct TextRec { uint32_t start, hdr_end, t_off, t_len, n_bases, q_off, q_len; };      /* offsets inside the stretch: delimiter, the '\n' that ends the header (or the end), sequence extent, quality extent */
struct ScanArgs {
	const uint8_t *text; uint32_t n;       /* the stretch: text[skip, n) -- `text` is 64-byte aligned, the first `skip` (< 64) bytes in front of the stretch are not part of it */
	uint32_t skip;
	uint32_t fastq;                        /* 0: FASTA ('>'), 1: FASTQ */
	uint64_t *ma, *mb;                     /* per 64-byte word: '\n' mask; record-start mask (FASTA) / '+' mask (FASTQ) */
	uint32_t *blk;                         /* per block of 256 words: [2 b] = bits of ma, [2 b + 1] = bits of mb; after the block scan: exclusive prefixes, totals at [2 n_blk], [2 n_blk + 1] */
	uint32_t n_blk;
	uint32_t *pos; uint32_t pos_cap;       /* FASTA: record starts in order; FASTQ: line starts in order (pos[0] = 0) */
	uint32_t *cum;                         /* per word: bits of the other mask in front of it */
	TextRec *rec; uint32_t n_rec; uint32_t last;        /* records to describe; last = the stretch ends the text (its last record ends there) */
	uint32_t keep_qual;
	uint32_t *flag;                        /* [0] nonzero: the stretch is not in the shape the kernels handle (FASTQ), [1] positions did not fit pos_cap */
};
/* bit i set iff byte i of the 8 bytes equals c */
__device__ __forceinline__ uint32_t bytes_eq4(uint32_t x, uint32_t pat)
{
	const uint32_t v = x ^ pat, t = ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v;          /* top bit of each byte clear iff the byte is zero */
	return ((~t & 0x80808080u) >> 7) * 0x10204080u >> 28;                          /* the four flags (bits 0, 8, 16, 24) gathered into bits 0..3 */
}
__global__ void __launch_bounds__(256) mm_text_marks_kernel(ScanArgs a)
{
	const uint32_t w = blockIdx.x * 256u + threadIdx.x, n_words = (a.n + 63u) / 64u;
	uint64_t nl = 0, dm = 0;
	if(w < n_words) {
		const uint4 *p = (const uint4 *)(a.text + (uint64_t)w * 64);
		const uint32_t pat = a.fastq ? 0x2b2b2b2bu : 0x3e3e3e3eu;
		for(int q = 0; q < 4; q++) {
			const uint4 v = p[q]; const uint32_t x[4] = { v.x, v.y, v.z, v.w };
			for(int j = 0; j < 4; j++) { nl |= (uint64_t)bytes_eq4(x[j], 0x0a0a0a0au) << (16 * q + 4 * j); dm |= (uint64_t)bytes_eq4(x[j], pat) << (16 * q + 4 * j); }
		}
		const uint32_t live = a.n - w * 64u;          /* bytes of the word inside the stretch */
		if(live < 64u) { const uint64_t m = (1ull << live) - 1; nl &= m; dm &= m; }
		if(w == 0 && a.skip) { const uint64_t m = ~((1ull << a.skip) - 1); nl &= m; dm &= m; }          /* (the bytes in front of the stretch: whatever the buffer held) */
		if(!a.fastq) {
			/* record starts: the first '>' of a line.  What came last in front of a '>' -- a '\n' (or the beginning of the stretch, which is a record start by
			 * construction) makes it one, another '>' does not; found inside the word when it holds either, else by walking the text backwards (one step for a '>'
			 * at the beginning of a line) */
			uint64_t st = 0, g = dm; const uint64_t ev = nl | dm;
			while(g) {
				const int b = __ffsll((long long)g) - 1; g &= g - 1;
				const uint64_t below = ev & ((1ull << b) - 1);
				bool start;
				if(below) { start = (nl >> (63 - __clzll((long long)below))) & 1; }
				else {
					start = true;
					for(int64_t q = (int64_t)w * 64 + b - 1; q >= (int64_t)a.skip; q--) { const uint8_t c = a.text[q]; if(c == (uint8_t)'\n') { break; } if(c == (uint8_t)'>') { start = false; break; } }
				}
				if(start) { st |= 1ull << b; }
			}
			dm = st;
		}
		a.ma[w] = nl; a.mb[w] = dm;
	}
	/* block totals */
	uint32_t ca = (uint32_t)__popcll(nl), cb = (uint32_t)__popcll(dm);
	for(int o = 32; o > 0; o >>= 1) { ca += (uint32_t)__shfl_xor((int)ca, o); cb += (uint32_t)__shfl_xor((int)cb, o); }
	__shared__ uint32_t sa[4], sb[4];
	if((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = ca; sb[threadIdx.x >> 6] = cb; }
	__syncthreads();
	if(threadIdx.x == 0) { a.blk[2 * blockIdx.x] = sa[0] + sa[1] + sa[2] + sa[3]; a.blk[2 * blockIdx.x + 1] = sb[0] + sb[1] + sb[2] + sb[3]; }
}
__global__ void __launch_bounds__(256) mm_text_blocks_kernel(ScanArgs a)
{
	/* exclusive scan of the two interleaved columns of blk over n_blk blocks: every thread sums a contiguous slice, the 256 partial sums are scanned in LDS.  (One
	 * workgroup of four waves: a block of sixteen waves waited up to 180 ms for a CU with that many free slots beside the extension waves.) */
	__shared__ uint32_t pa[256], pb[256];
	const uint32_t t = threadIdx.x, per = (a.n_blk + 255u) / 256u, lo = min(a.n_blk, t * per), hi = min(a.n_blk, lo + per);
	uint32_t xa = 0, xb = 0;
	for(uint32_t i = lo; i < hi; i++) { xa += a.blk[2 * i]; xb += a.blk[2 * i + 1]; }
	pa[t] = xa; pb[t] = xb; __syncthreads();
	for(uint32_t o = 1; o < 256; o <<= 1) { uint32_t ya = t >= o ? pa[t - o] : 0, yb = t >= o ? pb[t - o] : 0; __syncthreads(); pa[t] += ya; pb[t] += yb; __syncthreads(); }
	uint32_t ra = pa[t] - xa, rb = pb[t] - xb;          /* exclusive */
	for(uint32_t i = lo; i < hi; i++) { const uint32_t va = a.blk[2 * i], vb = a.blk[2 * i + 1]; a.blk[2 * i] = ra; a.blk[2 * i + 1] = rb; ra += va; rb += vb; }
	if(t == 255) { a.blk[2 * a.n_blk] = pa[255]; a.blk[2 * a.n_blk + 1] = pb[255]; }
}
__global__ void __launch_bounds__(256) mm_text_emit_kernel(ScanArgs a)
{
	const uint32_t w = blockIdx.x * 256u + threadIdx.x, n_words = (a.n + 63u) / 64u;
	const uint64_t nl = w < n_words ? a.ma[w] : 0, dm = w < n_words ? a.mb[w] : 0;
	/* FASTA: positions of the starts (mb), newlines counted (ma); FASTQ: positions behind the newlines (ma), '+' counted (mb) */
	const uint64_t em = a.fastq ? nl : dm, cm = a.fastq ? dm : nl;
	uint32_t ce = (uint32_t)__popcll(em), cc = (uint32_t)__popcll(cm);
	uint32_t pe = ce, pc = cc;
	const int lane = threadIdx.x & 63;
	for(int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)pe, o), y = (uint32_t)__shfl_up((int)pc, o); if(lane >= o) { pe += x; pc += y; } }
	__shared__ uint32_t se[4], sc[4];
	if(lane == 63) { se[threadIdx.x >> 6] = pe; sc[threadIdx.x >> 6] = pc; }
	__syncthreads();
	uint32_t be = a.blk[2 * blockIdx.x + (a.fastq ? 0 : 1)], bc = a.blk[2 * blockIdx.x + (a.fastq ? 1 : 0)];
	for(uint32_t i = 0; i < (threadIdx.x >> 6); i++) { be += se[i]; bc += sc[i]; }
	be += pe - ce; bc += pc - cc;          /* exclusive prefixes of this word */
	if(w < n_words) {
		a.cum[w] = bc;
		uint64_t g = em; uint32_t k = be + (a.fastq ? 1u : 0u);          /* (FASTQ: pos[0] = 0 is the first line, written by the host) */
		while(g) { const int b = __ffsll((long long)g) - 1; g &= g - 1; if(k < a.pos_cap) { a.pos[k] = w * 64u + (uint32_t)b + (a.fastq ? 1u : 0u); } else { a.flag[1] = 1; } k++; }
	}
}
/* bits of mask m (per-word masks `ma`, counts in front of every word `cum`) in front of byte position p */
__device__ __forceinline__ uint32_t bits_before(const uint64_t *m, const uint32_t *cum, uint32_t p, uint32_t n)
{
	if(p >= n) { p = n; }
	const uint32_t w = p >> 6, n_words = (n + 63u) / 64u;
	if(w >= n_words) { return cum[n_words - 1] + (uint32_t)__popcll(m[n_words - 1]); }
	return cum[w] + (uint32_t)__popcll(m[w] & ((1ull << (p & 63)) - 1));
}
__device__ __forceinline__ uint32_t bits_before(const uint64_t *m, const uint32_t *cum, uint32_t p, uint32_t n);
/* K0 for long sequences (a reference: a chromosome is 250 MB of text, which one wave per record would walk for seconds): a wave per tile of a record's text; where the
 * tile's bases go in the arena follows from the number of newlines between the start of the record's extent and the tile (the scan's masks and counts, FASTA only).
 * tile_base[r] = tiles of the records in front of r (n_reads + 1 entries) */
__global__ void __launch_bounds__(256) mm_text_codes_tiled_kernel(const uint8_t *text, const uint64_t *ma, const uint32_t *cum, uint32_t n, const TextRead *tr, const uint32_t *tile_base,
	uint32_t n_reads, uint32_t tile, uint8_t *codes)
{
	const int lane = lane_id();
	const uint32_t g = (uint32_t)rdfirst((int)(blockIdx.x * 4 + threadIdx.x / 64));
	if(g >= tile_base[n_reads]) { return; }
	uint32_t lo = 0, hi = n_reads;
	while(hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if(tile_base[mid] <= g) { lo = mid; } else { hi = mid; } }
	const uint32_t r = lo, ti = g - (uint32_t)rdfirst((int)tile_base[r]);
	const uint32_t t0 = (uint32_t)rdfirst64(tr[r].t_off), tl = (uint32_t)rdfirst((int)tr[r].t_len); const uint64_t q0 = rdfirst64(tr[r].q_off);
	const uint32_t a = ti * tile, b = a + tile < tl ? a + tile : tl;
	const uint64_t lut = 0x0400000020331000ull;
	uint64_t out = q0 + a - (bits_before(ma, cum, t0 + a, n) - bits_before(ma, cum, t0, n));
	for(uint32_t i0 = a; i0 < b; i0 += 64) {
		const uint32_t i = i0 + (uint32_t)lane;
		const uint8_t c = i < b ? text[t0 + i] : (uint8_t)'\n';
		const bool keep = c != (uint8_t)'\n';
		const uint64_t m = __ballot(keep);
		if(keep) { codes[out + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (uint8_t)((lut >> (4 * (c & 15))) & 15); }
		out += (uint32_t)__popcll(m);
	}
}
__global__ void __launch_bounds__(256) mm_text_fasta_kernel(ScanArgs a)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if(r >= a.n_rec) { return; }
	const uint32_t n_words = (a.n + 63u) / 64u;
	const uint32_t start = a.pos[r], total = a.blk[2 * a.n_blk + 1];
	const uint32_t end = r + 1 < total ? a.pos[r + 1] : a.n;
	/* the '\n' that ends the header line: the first one behind the delimiter (end when the header runs to the end of the record) */
	uint32_t he = end;
	for(uint32_t w = start >> 6; w < n_words && (w << 6) < end; w++) {
		uint64_t m = a.ma[w]; if(w == (start >> 6)) { m &= ~((1ull << (start & 63)) - 1); }
		if(m) { const uint32_t q = (w << 6) + (uint32_t)(__ffsll((long long)m) - 1); if(q < end) { he = q; } break; }
	}
	TextRec t; t.start = start; t.hdr_end = he; t.q_off = 0; t.q_len = 0;
	t.t_off = he < end ? he + 1 : end; t.t_len = end - t.t_off;
	t.n_bases = t.t_len - (bits_before(a.ma, a.cum, end, a.n) - bits_before(a.ma, a.cum, t.t_off, a.n));
	a.rec[r] = t;
}
__global__ void __launch_bounds__(256) mm_text_fastq_kernel(ScanArgs a)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if(r >= a.n_rec) { return; }
	/* lines 4 r .. 4 r + 3; line j = [pos[j], pos[j + 1] - 1), the last line of the text may end at n without a '\n' */
	const uint32_t n_lines = a.blk[2 * a.n_blk] + 1;          /* line starts known: pos[0 .. n_lines) */
	auto line_end = [&](uint32_t j) -> uint32_t { return j + 1 < n_lines ? a.pos[j + 1] - 1 : a.n; };
	const uint32_t l0 = a.pos[4 * r], l1 = a.pos[4 * r + 1], l2 = a.pos[4 * r + 2], l3 = a.pos[4 * r + 3];
	const uint32_t e1 = l2 - 1, e3 = line_end(4 * r + 3);
	bool ok = a.text[l0] == (uint8_t)'@' && l2 < a.n && a.text[l2] == (uint8_t)'+';
	ok = ok && bits_before(a.mb, a.cum, e1, a.n) == bits_before(a.mb, a.cum, l1, a.n);          /* no '+' among the bases (it would end them there) */
	const uint32_t nb = e1 - l1;
	uint32_t ql = e3 - l3;
	/* the qualities must reach the number of bases on their first line: counted with a trailing CR when they are only skipped, without it when they are kept (minialign.c:2050-2070) */
	if(a.keep_qual && ql > 0 && a.text[l3 + ql - 1] == (uint8_t)'\r') { ql--; }
	ok = ok && ql >= nb && l3 <= a.n;
	if(!ok) { a.flag[0] = 1; }
	TextRec t; t.start = l0; t.hdr_end = l1 - 1; t.t_off = l1; t.t_len = nb; t.n_bases = nb; t.q_off = l3; t.q_len = ql;
	a.rec[r] = t;
}
