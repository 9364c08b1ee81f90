/*
 * K4: CIGAR strings on the device (second half).
 *
 * When the extension launch of a batch is over, the path words, the segments and the alignment records of every alignment it recorded are in HBM.  The printer of the
 * reference turns the path bits of a segment into run lengths with gaba_dp_print_cigar_reverse (gaba_parse.h:168-221, called from minialign.c:5147-5200 for the CIGAR
 * column and :5390-5426 for the SA tag): from the END of the segment's stretch of path bits downwards, a run of 0 bits is a deletion, a run of 1 bits an insertion, a run
 * of 01 pairs a match -- each test on the 64 bits BELOW the cursor, counted with a leading-zero count, the deletion run one short when it is followed by anything
 * (that 0 is the first half of a 01 pair).  That parser is restated here bit for bit (cig_next), one lane per segment (mm_cigar_list_kernel lists them): a first walk counts the
 * characters, the lane takes that many bytes of the batch's text buffer (one atomic add), a second walk writes them, eight characters per store.  The kernel is queued on the
 * lane's stream right behind every extension launch (run_rounds), over the reads of that launch's work list: it starts when the launch ends, needs no turn of the host, and
 * the reads a carried-value re-run maps again simply get new strings.  What crosses PCIe for a default SAM run is then this text and a
 * (offset, length) pair per segment instead of the path words; the host splices names, flags, positions, mapping qualities (libm's log10: host), SEQ / QUAL
 * and the clips around the string (sam_record).  Runs that print MD tags, the other output formats and the mm_reg_t entries walk the path on the host as before.
 *
 * Bound: HBM latency of a lane's dependent 8-byte reads (one per run) and its byte stores -- a few milliseconds per 300 Mb batch beside the extension launches of the
 * other lanes; algorithmic bytes: path bits / 8 read twice + text written once.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "mm_device.hpp"

namespace mm {

struct CigEnt { uint32_t off, len; };                                        /* where the string of a segment slot stands in the text buffer (len = ~0: it did not fit) */

/* 64 path bits from absolute bit position p of the pool on (p >= 0: two header words stand in front of every path, gaba.h:217).  The parser asks for a fresh window per
 * test and moves down a few bits per run, so the lane keeps 256 bits of the pool in registers (four 8-byte words from word cq on) and goes to memory once per ~190 bits it
 * has walked instead of once per test: the walk is a chain of dependent reads, and a dependent read is a round trip to L2 / HBM */
struct CigWalk { const uint32_t *pool; uint64_t base; uint64_t idx; int phase; uint64_t cq; uint64_t c0, c1, c2, c3; uint64_t lp, lv; };
__host__ __device__ __forceinline__ uint64_t cig_bits(CigWalk &w, uint64_t p)
{
	if(p == w.lp) { return w.lv; }          /* (a test that took nothing leaves the cursor where it was: the next test of the turn looks at the same 64 bits -- two of three do) */
	w.lp = p;
	const uint64_t q = p >> 6; const uint32_t r = (uint32_t)p & 63u;
	if(q < w.cq || q > w.cq + 2) {
		w.cq = q >= 2 ? q - 2 : 0;
		const uint64_t *m = (const uint64_t *)w.pool + w.cq;
		w.c0 = m[0]; w.c1 = m[1]; w.c2 = m[2]; w.c3 = m[3];
	}
	const uint64_t k = q - w.cq;
	const uint64_t lo = k == 0 ? w.c0 : (k == 1 ? w.c1 : w.c2), hi = k == 0 ? w.c1 : (k == 1 ? w.c2 : w.c3);
	w.lv = r ? (lo >> r) | (hi << (64u - r)) : lo;
	return w.lv;
}
__host__ __device__ __forceinline__ uint64_t cig_lzc(uint64_t x) { return x ? (uint64_t)__builtin_clzll(x) : 64ull; }
/* the next run of the reverse parser (gaba_parse.h:183-216): base = absolute bit position of the segment's first path bit minus 64, idx = bits left; returns the run's
 * length and its letter ('D', 'I', 'M'; 0 when the step printed nothing), idx moved down */
__host__ __device__ __forceinline__ uint64_t cig_next(CigWalk &w, char &op)
{
	if(w.phase == 0) {
		w.phase = 1;
		const uint64_t m = cig_lzc(cig_bits(w, w.base + w.idx)), d = m - (m > 0 ? 1ull : 0ull), c = w.idx < d ? w.idx : d;
		w.idx -= c; op = 'D'; return c;
	}
	if(w.phase == 1) {
		w.phase = 2;
		const uint64_t m = cig_lzc(~cig_bits(w, w.base + w.idx)), c = w.idx < m ? w.idx : m;
		w.idx -= c; op = 'I'; return c;
	}
	w.phase = 0;
	const uint64_t sidx = w.idx; uint64_t c;
	do { const uint64_t m = cig_lzc(cig_bits(w, w.base + w.idx) ^ 0x5555555555555555ull); c = (w.idx < m ? w.idx : m) & ~1ull; w.idx -= c; } while(c == 64);
	op = 'M'; return (sidx - w.idx) >> 1;
}

/* the characters of a string go out eight at a time: a lane writes its string front to back, and a byte store per character -- 64 lanes, 64 cache lines per store
 * instruction, 18 000 of them per 20 kb read -- was what the first version of this kernel spent its time on (32 ms per 300 Mb batch) */
struct CigOut { char *o; uint64_t pos, acc; uint32_t n; };
__host__ __device__ __forceinline__ void cig_put(CigOut &c, char b)
{
	if(c.o == nullptr) { c.pos++; return; }
	if(c.n == 0 && (((uintptr_t)(c.o + c.pos)) & 7u) != 0) { c.o[c.pos++] = b; return; }          /* (up to the first 8-byte boundary) */
	c.acc |= (uint64_t)(uint8_t)b << (8u * c.n);
	if(++c.n == 8) { *(uint64_t *)(c.o + c.pos) = c.acc; c.pos += 8; c.acc = 0; c.n = 0; }
}
__host__ __device__ __forceinline__ uint64_t cig_flush(CigOut &c)
{
	for(uint32_t i = 0; i < c.n; i++) { c.o[c.pos + i] = (char)(c.acc >> (8u * i)); }
	c.pos += c.n; c.n = 0; c.acc = 0;
	return c.pos;
}
/* one string, by the code of the kernel below (host side: tests pin the restated parser on the reference's own without a device, mm_cigar_walk) */
__host__ __device__ __forceinline__ uint64_t cig_write(const uint32_t *pool, uint64_t base, uint64_t len, char *o)
{
	CigOut out{ o, 0, 0, 0 };
	CigWalk w{ pool, base, len, 0, ~0ull >> 1, 0, 0, 0, 0, ~0ull, 0 };
	while(w.idx != 0) {          /* (gaba_parse.h:183: all three tests per turn, also when the first one used the bits up) */
		const uint64_t before = w.idx;
		for(int ph = 0; ph < 3; ph++) {
			char op; uint64_t c = cig_next(w, op);
			if(!c) { continue; }
			if(c < 10) { cig_put(out, (char)('0' + c)); }          /* (most runs have one or two digits) */
			else if(c < 100) { cig_put(out, (char)('0' + c / 10)); cig_put(out, (char)('0' + c % 10)); }
			else { uint64_t p10 = 100; while(p10 * 10 <= c) { p10 *= 10; } for(; p10; p10 /= 10) { cig_put(out, (char)('0' + (c / p10) % 10)); } }
			cig_put(out, op);
		}
		if(w.idx == before) { break; }          /* (bits that are no path -- a lone 0 above a 1 at the bottom of the stretch: the reference's loop would turn for ever; never seen on what the traceback writes) */
	}
	return cig_flush(out);
}
struct CigItem { uint32_t slot; uint32_t pad; uint64_t path_word; };          /* one segment of a recorded alignment: its slot in the segment pool, the first path word of its alignment */
struct CigArgs { const ReadState *st; const uint32_t *work; uint32_t n_work; const AlnRec *aln_pool; const gaba::Segment *seg_pool; const uint32_t *path_pool;
	CigItem *items; uint64_t item_cap; CigEnt *ent; uint64_t ent_cap; char *text; uint64_t text_cap;
	unsigned long long *ctl; };          /* ctl[0] = segments done, [1] = text bytes taken, [2] = something did not fit, [3] = items listed by the launch at hand (zeroed in front of it) */
/* the segments of every alignment the reads of the extension launch in front recorded, as a list (thread per read of its work list): a read inside a repeat family
 * records hundreds of alignments, and with a lane per READ its strings were one lane's work -- the hard-repeat human-size set fell from 1.8 to 0.5 G bases/s */
__global__ void __launch_bounds__(256) mm_cigar_list_kernel(CigArgs a)
{
	__builtin_amdgcn_s_setprio(3);
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if(k >= a.n_work) { return; }
	const ReadState &rs = a.st[a.work[k]];
	if(rs.n_aln == 0 || rs.bin_off == ~0ull) { return; }
	const AlnRec *al = a.aln_pool + rs.aln_off;
	uint32_t n = 0; for(uint32_t i = 0; i < rs.n_aln; i++) { n += al[i].slen; }
	if(n == 0) { return; }
	unsigned long long at = atomicAdd(&a.ctl[3], (unsigned long long)n);
	if(at + n > a.item_cap) { atomicExch(&a.ctl[2], 1ull); return; }
	for(uint32_t i = 0; i < rs.n_aln; i++) { for(uint32_t j = 0; j < al[i].slen; j++) { a.items[at++] = CigItem{ al[i].seg_off + j, 0u, al[i].path_off }; } }
}
/* lane per segment of that list: count, take room, write.  At the top issue priority: the walk is a chain of dependent steps, a few hundred waves long, beside
 * extension waves that fill every SIMD -- at their priority it got a ninth of the issue slots and the lane's D2H waited for it (E.coli-size set: 2.3 -> 1.6 G bases/s) */
__global__ void __launch_bounds__(256) mm_cigar_kernel(CigArgs a)
{
	__builtin_amdgcn_s_setprio(3);
	const unsigned long long n = min(a.ctl[3], (unsigned long long)a.item_cap);
	uint32_t done = 0;
	for(unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * 256u) {
		const CigItem it = a.items[k];
		if(it.slot >= a.ent_cap) { atomicExch(&a.ctl[2], 1ull); continue; }
		const gaba::Segment sg = a.seg_pool[it.slot];
		const uint64_t len = (uint64_t)sg.alen + sg.blen;
		/* (the host's parser aligns its pointer down to 8 bytes and adds 32 to the offset when it had to, gaba_parse.h:176-177: the same absolute bit either way) */
		const uint64_t base = it.path_word * 32ull + sg.ppos - 64ull;
		const uint64_t chars = cig_write(a.path_pool, base, len, nullptr);
		const unsigned long long at = atomicAdd(&a.ctl[1], (unsigned long long)chars);
		if(at + chars > a.text_cap || chars > 0xfffffff0ull) { a.ent[it.slot] = CigEnt{ 0u, 0xffffffffu }; atomicExch(&a.ctl[2], 1ull); continue; }
		a.ent[it.slot] = CigEnt{ (uint32_t)at, (uint32_t)chars };
		(void)cig_write(a.path_pool, base, len, a.text + at);
		done++;
	}
	if(done) { atomicAdd(&a.ctl[0], (unsigned long long)done); }
}

} /* namespace mm */
