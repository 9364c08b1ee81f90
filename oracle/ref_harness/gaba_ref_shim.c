/*
 * gaba_ref_shim.c -- TEST INFRASTRUCTURE (oracle/_ref build recipe input).
 *
 * A thin exported wrapper around the *reference* libgaba, compiled from the sources where
 * they lie under /root/reference (this file only #includes the reference's public wrapper
 * header by absolute path; no reference source is copied into this repo).  The resulting
 * oracle/_ref/libgaba_ref.so is used by tests/ to pin oracle/ora_gaba.c against the real
 * implementation on random inputs, and to regenerate tests/golden/ fixtures.
 *
 * Mirrors the call pattern of mm_extend_core (minialign.c:4075) + gaba_dp_search_max
 * (minialign.c:4142) + gaba_dp_trace (minialign.c:4154).
 */
#define _GABA_PARSE_EXPORT_LEVEL static inline
#define _GABA_WRAP_EXPORT_LEVEL  static inline
#define UNITTEST 0
#define BIT 2                          /* as minialign.c:155 sets it before it includes the same headers */
#include <stdlib.h>
#include <string.h>
#include "/root/reference/gaba_wrap.h"

typedef struct {
	int64_t max; uint32_t status; uint32_t aid, bid, ascnt, bscnt; uint64_t apos, bpos;
} shim_fill_t;

typedef struct {
	/* fills */
	uint32_t n_fill; uint32_t max_fill_idx;
	shim_fill_t fill[8];
	/* search_max on the max fill */
	uint32_t p_aid, p_bid, p_apos, p_bpos; uint64_t p_plen;
	/* trace */
	int32_t traced;                 /* 0: not requested, 1: ok, -1: NULL */
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, dcnt, slen, plen;
	struct gaba_segment_s seg[16];
	uint32_t n_path_words;
} shim_result_t;

static void *shim_malloc(void *opaque, size_t size) { (void)opaque; return malloc(size); }
static void shim_free(void *opaque, void *ptr) { (void)opaque; free(ptr); }

gaba_t *shim_init(int8_t const *score_matrix, int gi, int ge, int gfa, int gfb, int xdrop)
{
	gaba_params_t p; memset(&p, 0, sizeof(p));
	memcpy(p.score_matrix, score_matrix, 16);
	p.gi = gi; p.ge = ge; p.gfa = gfa; p.gfb = gfb; p.xdrop = xdrop;
	return gaba_init(&p);
}
void shim_clean(gaba_t *ctx) { gaba_clean(ctx); }
gaba_dp_t *shim_dp_init(gaba_t *ctx) { return gaba_dp_init(ctx); }
void shim_dp_clean(gaba_dp_t *dp) { gaba_dp_clean(dp); }

/*
 * a/b: 1 byte per base (0..3, 4 = N).  arev/brev: use mirrored (reverse-complement) section.
 * Tail sections are 96 N's as in minialign.c:4512-4519.  bw_idx: 0 -> 64, 1 -> 32, 2 -> 16.
 * do_trace: 0 none, 1 trace the max fill (only if max >= trace_min).
 * path_out must hold (alen + blen + 256) / 32 + 16 words.
 */
int shim_extend(gaba_dp_t *dp0, int bw_idx,
	uint8_t const *a, uint32_t alen, uint32_t apos, int arev,
	uint8_t const *b, uint32_t blen, uint32_t bpos, int brev,
	int do_trace, shim_result_t *res, uint32_t *path_out)
{
	static uint8_t tailseq[128];
	memset(tailseq, 4, 128);
	gaba_dp_t *dp = &dp0[bw_idx];
	gaba_dp_flush(dp0);

	gaba_section_t as = gaba_build_section(arev ? 1 : 0, arev ? gaba_mirror(a, alen) : a, alen);
	gaba_section_t bs = gaba_build_section(brev ? 3 : 2, brev ? gaba_mirror(b, blen) : b, blen);
	gaba_section_t ts = gaba_build_section(0xfffffffe, tailseq, 96);
	gaba_section_t const *ap = &as, *bp = &bs;

	memset(res, 0, sizeof(*res));
	gaba_fill_t const *f = gaba_dp_fill_root(dp, ap, apos, bp, bpos, 0);
	if(f == NULL) { return -1; }
	gaba_fill_t const *m = f;
	#define _rec(_f) { shim_fill_t *s = &res->fill[res->n_fill < 8 ? res->n_fill : 7]; \
		s->max = (_f)->max; s->status = (_f)->status; s->aid = (_f)->aid; s->bid = (_f)->bid; \
		s->ascnt = (_f)->ascnt; s->bscnt = (_f)->bscnt; s->apos = (_f)->apos; s->bpos = (_f)->bpos; res->n_fill++; }
	_rec(f);
	uint32_t flag = GABA_TERM;
	while((flag & f->status) == 0) {
		if(f->status & GABA_UPDATE_A) { ap = &ts; }
		if(f->status & GABA_UPDATE_B) { bp = &ts; }
		flag |= f->status & (GABA_UPDATE_A | GABA_UPDATE_B);
		if((f = gaba_dp_fill(dp, f, ap, bp, 0)) == NULL) { return -1; }
		_rec(f);
		if(f->max > m->max) { m = f; res->max_fill_idx = res->n_fill - 1; }
	}
	gaba_pos_pair_t const *pp = gaba_dp_search_max(dp, m);
	res->p_aid = pp->aid; res->p_bid = pp->bid; res->p_apos = pp->apos; res->p_bpos = pp->bpos; res->p_plen = pp->plen;

	if(do_trace) {
		gaba_alloc_t alloc = { NULL, shim_malloc, shim_free };
		gaba_alignment_t *aln = gaba_dp_trace(dp, m, &alloc);
		if(aln == NULL) { res->traced = -1; return 0; }
		res->traced = 1;
		res->score = aln->score; res->identity = aln->identity;
		res->agcnt = aln->agcnt; res->bgcnt = aln->bgcnt; res->dcnt = aln->dcnt;
		res->slen = aln->slen; res->plen = aln->plen;
		for(uint32_t i = 0; i < aln->slen && i < 16; i++) { res->seg[i] = aln->seg[i]; }
		uint32_t nw = (aln->plen + 31) / 32;
		res->n_path_words = nw;
		for(uint32_t i = 0; i < nw; i++) {
			uint32_t w = aln->path[i];
			if(i == nw - 1 && (aln->plen & 31)) { w &= (1u << (aln->plen & 31)) - 1; }
			path_out[i] = w;
		}
		shim_free(NULL, aln);
	}
	return 0;
}

/* CIGAR dumpers (gaba_parse.h:259) for the CIGAR known-answer tests */
uint64_t shim_dump_cigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	return gaba_dump_cigar_reverse(buf, buf_size, path, offset, len);
}
uint64_t shim_dump_cigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	return gaba_dump_cigar_forward(buf, buf_size, path, offset, len);
}

/* the other text dumpers of gaba_parse.h (extended CIGAR :274-372, gapped sequence rows :380-529) and gaba_dp_calc_score (gaba.c:3493), for the golden
 * vectors of tests/golden/make_dumper_golden.py.  Sections are built as in shim_extend; `path` points behind the two header words of gaba_alignment_s.
 * out[0..4] receive the five strings (xcigar forward / reverse, row A, row B through gaba_dump_seq_ref / _query, row A reverse as the MAF printer calls it),
 * each of capacity cap; sc receives score, mcnt, xcnt, agcnt, bgcnt, aicnt, bicnt, afgcnt, bfgcnt, aficnt, bficnt, adj and *identity the identity. */
int shim_dumpers(gaba_dp_t *dp0, uint8_t const *a, uint32_t alen, int arev, uint8_t const *b, uint32_t blen, int brev,
	uint32_t const *path, struct gaba_segment_s const *seg, char *out, uint64_t cap, int64_t *sc, double *identity)
{
	gaba_section_t as = gaba_build_section(arev ? 1 : 0, arev ? gaba_mirror(a, alen) : a, alen);
	gaba_section_t bs = gaba_build_section(brev ? 3 : 2, brev ? gaba_mirror(b, blen) : b, blen);
	gaba_dump_xcigar_forward(out + 0 * cap, cap, path, seg, &as, &bs);
	gaba_dump_xcigar_reverse(out + 1 * cap, cap, path, seg, &as, &bs);
	gaba_dump_seq_ref(out + 2 * cap, cap, path, seg, &as);
	gaba_dump_seq_query(out + 3 * cap, cap, path, seg, &bs);
	if(!arev) { gaba_dump_seq_reverse(out + 4 * cap, cap, GABA_SEQ_A | GABA_SEQ_FW, path, seg->ppos, gaba_plen(seg), &a[seg->apos], '-'); } else { out[4 * cap] = 0; }
	gaba_dp_flush(dp0);
	gaba_score_t const *s = gaba_dp_calc_score(dp0, path, seg, &as, &bs);
	if(s == NULL) { return -1; }
	sc[0] = s->score; sc[1] = s->mcnt; sc[2] = s->xcnt; sc[3] = s->agcnt; sc[4] = s->bgcnt; sc[5] = s->aicnt; sc[6] = s->bicnt;
	sc[7] = s->afgcnt; sc[8] = s->bfgcnt; sc[9] = s->aficnt; sc[10] = s->bficnt; sc[11] = s->adj;
	*identity = s->identity;
	return 0;
}
