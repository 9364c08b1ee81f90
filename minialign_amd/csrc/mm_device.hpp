/*
 * mm_device.hpp -- CDNA4 (gfx950) kernels for the seed-and-extend mapper hot path.
 *
 *   K1  mm_sketch_seed_kernel   one wavefront per read: lanes = k-mer positions over 2-bit packed sequence;
 *                               sliding-window minimum by lane shuffles; ordered compaction by ballot;
 *                               index probe (open-addressing table in HBM); seed expansion.
 *                               (reference behaviour: mm_sketch minialign.c:2410, mm_idx_get :2728,
 *                               mm_collect_seed :3454, mm_expand :3420)
 *   K2  mm_sort_chain_kernel    one lane per read: the reference's in-place MSD radix / insertion sort
 *                               reproduced step by step (ksort.h:84-131 -- its *unstable* permutation is part of
 *                               the observable result), then greedy window chaining (mm_chain_seeds :3547) and
 *                               the chain-length sort.
 *   K3  mm_extend_kernel        one wavefront per read: the extension state machine (mm_extend :4118 and
 *                               mm_search_* :3785-4067) driving the banded DP of gaba_device.hpp, with the
 *                               per-read position hash (kh_t :341-683) kept literally, slot for slot.
 *
 * Host code (mm_host.hip) owns parsing, index construction, post-map / MAPQ and SAM.
 */
#pragma once
#include "gaba_device.hpp"

namespace mm {

using gaba::rdfirst; using gaba::rdfirst64; using gaba::rdlane; using gaba::lane_id;

/* ---- index in HBM: one open-addressing table keyed by the minimizer value ---- */
struct IdxSlot { uint64_t key; uint64_t val; };      /* key = minimizer + 1 (0 = empty); val: bit 63 clear -> single hit (pos | rid << 32),
                                                       set -> (offset << 24 | count) into the value array */
struct DevIndex {
	const IdxSlot *slot; uint64_t mask;                /* table size - 1 */
	const uint64_t *val;                               /* multi-hit lists: pos | rid << 32, in the reference's list order */
	const uint32_t *seq_len;                           /* per reference sequence */
	const uint8_t *seq_circ;                           /* per reference sequence: circular (-c); NULL when none is */
	const uint64_t *seq_off;                           /* first base of each reference sequence in the a-side arena */
	uint32_t n_seq, k, w, n_occ;
	uint32_t occ[8];
};
__device__ __forceinline__ uint64_t idx_hash(uint64_t x) { x ^= x >> 31; x *= 0x9e3779b97f4a7c15ull; x ^= x >> 29; return x; }

/* ---- per-read records ---- */
struct ReadIn {
	uint64_t q_off;            /* first base in the b-side arena */
	uint32_t qlen;
	uint32_t rlen_in;          /* reference-length state carried in from the previous read (minialign.c:3864 reads it before mm_init_ref) */
};
struct MinRec { uint32_t qs, n; uint64_t ref; };      /* one looked-up minimizer: query pos (strand folded), #hits, inline hit or list offset */
struct Seed { uint32_t upos, rid, vpos, lid; };       /* mm_seed_t, minialign.c:3187 (leaf view: rsid, rid, lsid, cid -- :3190) */
struct Root { uint32_t plen, lid; };                  /* mm_root_t :3202, aliased by mm_res_t { score, iid } :3232 */
struct Resc { uint32_t qs, n; uint64_t ref; };        /* mm_resc_t :3176 */

struct ReadState {
	/* regions (element offsets into the shared pools) */
	uint64_t min_off; uint32_t min_cap, n_min;        /* MinRec pool */
	uint64_t seed_off; uint32_t seed_cap;             /* Seed pool: seeds, sentinel, leaves */
	uint32_t seed_n, n_seed;                          /* seed.n, self->n_seed */
	uint64_t resc_off; uint32_t n_resc, presc;        /* Resc pool */
	uint64_t root_off; uint32_t root_cap, n_root;     /* Root pool (also the result array) */
	uint32_t n_res;
	uint32_t rlen;                                    /* self->rlen carried across chains / rounds / reads */
	uint32_t rid_last;                                /* last reference loaded by mm_init_ref, NIL if none */
	uint32_t apos0, cond0;                            /* first mm_search_load_pos of the read: unadjusted apos and (bpos >= qlen) */
	uint32_t pred_rid;                                /* after chaining: reference of the last chain passing the length test */
	uint32_t done;                                    /* 1: finished (mapped or exhausted rounds) */
	uint32_t err;                                     /* sticky error flags */
	uint32_t kh_mask, kh_cnt, kh_ub;                  /* kh_t state of the per-read position hash (persists across rounds) */
	uint32_t seed_n0;                                 /* seed count as K1 left it: immutable, picks the size class of the first-round sort + chain */
	uint32_t n_pass, w_pass;                          /* after chaining: chains that pass the length test of mm_search_load_root (minialign.c:3849), their summed lengths -- the work the extension has with this read */
	uint32_t spec_off, spec_n;                        /* first trials of this read's chains computed by the other waves of the launch (SpecMemo entries), 0: none */
	uint32_t k3_trials, k3_hits, k3_chains;           /* diagnostics: extension trials of the read, of which taken from a chain job, chains walked */
	uint32_t k3_ticks, k3_vec, k3_fill_ticks, k3_trace_ticks;   /* diagnostics: s_memtime ticks (whole / DP fill / traceback) and DP vectors the extension kernel spent on this read */
	uint32_t k3_t0, k3_wait_ticks;                              /* ... when its wave took it (s_memtime >> 8) and the ticks it waited for a DP workspace (profiling build) */
	uint32_t n_bin; uint64_t bin_off;                 /* bin slot pool (uint64 slots) */
	uint32_t n_aln; uint64_t aln_off;                 /* alignment record pool */
	/* room of this read in the result-bin pool, the alignment pool and the position-hash pool, set by the host between chaining and extension from the number of chains
	 * that pass the length test (n_pass): a read inside a repeat family walks hundreds of chains, each with a bin header, an alignment and a few hash entries, where the
	 * typical read has one or two (0: the launch's defaults, K3Args.bin_cap_per_read / aln_cap_per_read / kh_cap at the read's number) */
	uint32_t bin_cap, aln_cap, kh_cap, dep; uint64_t kh_off;
	/* the carried reference length between reads of one launch.  The host predicts it from the chains (pred_rid); a read that has no chain worth a trial at
	 * the first occurrence threshold but rescue minimizers waiting leaves something nobody can predict -- whatever its later rounds find.  The reads whose value such a read
	 * decides (dep = its index; NIL: none) take it from the read itself inside the launch: the source (flags & RS_CARRY_SRC) publishes its final rlen (agent-scope release,
	 * carry_ready = 1), the dependent read waits for that before its first trial.  rlen_in: the value the read ran with, for the host's check */
	uint32_t flags, rlen_in, carry_ready, pad2_;
};
enum : uint32_t { RS_CARRY_SRC = 1 };
enum : uint32_t { ERR_SEED_CAP = 1, ERR_DP_SLAB = 2, ERR_PATH_CAP = 4, ERR_SEG_CAP = 8, ERR_KH_CAP = 16, ERR_BIN_CAP = 32, ERR_ALN_CAP = 64, ERR_NEXT_CAP = 128, ERR_STACK = 256, ERR_ABORT = 512 };          /* ERR_ABORT: the host's watchdog called the extension launch off while the read's wave was waiting (mm_extend_kernel, K3Args.wd): the batch runs again */

/* coordinate transforms, minialign.c:3340-3362 */
__device__ __forceinline__ int32_t OFS(int32_t x) { return (int32_t)0x40000000 - x; }
__device__ __forceinline__ uint32_t U_(int32_t x, int32_t y) { return (uint32_t)(((x << 1) - y) + OFS(0)); }
__device__ __forceinline__ uint32_t V_(int32_t x, int32_t y) { return (uint32_t)(((y << 1) - x) + OFS(0)); }
__device__ __forceinline__ int32_t AS(const Seed &p) { return (int32_t)(((p.upos - (uint32_t)OFS(0)) << 1) + (p.vpos - (uint32_t)OFS(0))) / 3; }
__device__ __forceinline__ int32_t BS(const Seed &p) { return (int32_t)(((p.vpos - (uint32_t)OFS(0)) << 1) + (p.upos - (uint32_t)OFS(0))) / 3; }

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

} /* namespace mm */
