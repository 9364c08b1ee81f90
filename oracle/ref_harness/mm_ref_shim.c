/*
 * mm_ref_shim.c -- TEST INFRASTRUCTURE (oracle/_ref build recipe input).
 *
 * Stage-level taps into the *reference* mapper: this translation unit #includes the reference's single
 * source file by absolute path (all its stage functions are `static`), renames its main(), and exports a
 * few thin wrappers so that tests can compare per-stage outputs (sketch, index lookup, seed array after the
 * radix sort, chain roots) of the CPU oracle and of the HIP kernels with the reference itself.
 * No reference source is copied into the repo; the .so is built into oracle/_ref/ (git-ignored).
 */
#define UNITTEST 0
#define main ref_main
#include "/root/reference/minialign.c"
#undef main

typedef struct {
	mm_opt_t *o;
	mm_idx_t *mi;
	mm_align_t *aln;
	mm_tbuf_t *t;
	lmm_t *lmm;
} mmref_t;

/* argv-style options, e.g. { "minialign", "-xpacbio", NULL }; ref_fa is indexed on the fly */
mmref_t *mmref_open(char const *const *argv, char const *ref_fa)
{
	mmref_t *h = calloc(1, sizeof(mmref_t));
	h->o = mm_opt_init(argv);
	if(h->o == NULL) { return NULL; }
	bseq_params_t br = h->o->b; br.keep_qual = 0; br.n_tag = 0;
	bseq_file_t *fp = bseq_open(&br, ref_fa);
	if(fp == NULL) { return NULL; }
	h->mi = mm_idx_gen(&h->o->c, fp, h->o->pt);
	bseq_close(fp);
	h->aln = mm_align_init(&h->o->a, h->mi, h->o->pt);
	h->t = (mm_tbuf_t *)h->aln->t[0];
	h->lmm = lmm_init_margin(NULL, 512 * 1024, sizeof(mm_aln_t), 0);
	return h;
}
uint32_t mmref_occ(mmref_t *h, uint32_t i) { return h->mi->occ[i]; }
uint32_t mmref_n_seq(mmref_t *h) { return h->mi->n_seq; }
uint32_t mmref_kwb(mmref_t *h, int which) { return which == 0 ? h->mi->k : (which == 1 ? h->mi->w : h->mi->b); }

uint64_t mmref_sketch(mmref_t *h, uint8_t const *seq, uint32_t len, uint64_t *out)
{
	uint64_v b = { 0 };
	mm_sketch_t sk;
	mm_sketch_init(&sk, h->mi->w, h->mi->k, &b);
	mm_sketch(&sk, seq, len);
	uint64_t n = 0;
	for(uint64_t *p = b.a; !mm_sketch_is_cap(*p); p++) { out[n++] = *p; }
	free(b.a);
	return n;
}
uint32_t mmref_idx_get(mmref_t *h, uint64_t minier, uint64_t *out, uint32_t max)
{
	uint32_t n = 0;
	v2u32_t const *r = mm_idx_get(h->mi, minier, &n);
	for(uint32_t i = 0; i < n && i < max; i++) { out[i] = r[i].u64[0]; }
	return n;
}
/* runs mm_seed(0..iter) on a fresh query; copies the sorted seed array (sentinel included) */
uint64_t mmref_seed(mmref_t *h, uint8_t const *seq, uint32_t len, uint64_t iter, uint32_t *out, uint64_t max)
{
	mm_tbuf_clear(h->t, h->lmm);
	mm_init_query(h->t, len, seq, 0, 0);
	uint64_t n = 0;
	for(uint64_t i = 0; i <= iter; i++) { n = mm_seed(h->t, i); }
	for(uint64_t i = 0; i < n && i < max; i++) { memcpy(&out[4 * i], &h->t->seed.a[i], 16); }
	return n;
}
/* mm_chain on the seeds of the last mmref_seed; copies roots (plen | lid << 32) and the leaf area */
uint64_t mmref_chain(mmref_t *h, uint64_t *roots, uint64_t max, uint32_t *leaves, uint64_t *n_leaves)
{
	uint64_t n = mm_chain(h->t, 0);
	for(uint64_t i = 0; i < n && i < max; i++) { memcpy(&roots[i], &h->t->root.a[i], 8); }
	uint64_t nl = h->t->seed.n - (h->t->n_seed + 1);
	for(uint64_t i = 0; i < nl && i < max; i++) { memcpy(&leaves[4 * i], &h->t->seed.a[h->t->n_seed + 1 + i], 16); }
	*n_leaves = nl;
	return n;
}
/* full mm_align_seq: returns n_all and a compact dump of the alignments (aid, mapq, score, plen, slen, seg0...) */
uint32_t mmref_align(mmref_t *h, uint8_t const *seq, uint32_t len, int64_t *out, uint32_t max)
{
	mm_reg_t const *reg = mm_align_seq(h->t, len, seq, 0, h->lmm);
	if(reg == NULL) { return 0; }
	uint32_t k = 0;
	for(uint32_t i = 0; i < reg->n_all && k + 12 <= max; i++) {
		mm_aln_t const *a = reg->aln[i];
		out[k++] = a->aid; out[k++] = a->mapq; out[k++] = a->a->score; out[k++] = a->a->plen; out[k++] = a->a->slen;
		out[k++] = a->a->seg[0].aid; out[k++] = a->a->seg[0].bid; out[k++] = a->a->seg[0].apos; out[k++] = a->a->seg[0].bpos;
		out[k++] = a->a->seg[0].alen; out[k++] = a->a->seg[0].blen; out[k++] = (int64_t)(i < reg->n_uniq);
	}
	return reg->n_all;
}
/* debugging tap: result array after the last mm_align_seq (score | iid << 32), n_res and the bin headers */
uint32_t mmref_res(mmref_t *h, uint64_t *out, uint32_t max)
{
	uint32_t n = h->t->n_res;
	for(uint32_t i = 0; i < n + 2 && i < max; i++) { memcpy(&out[i], &h->t->root.a[i], 8); }
	return n;
}
/* debugging tap: raw bin slots after the last mm_align_seq */
uint32_t mmref_bins(mmref_t *h, uint64_t *out, uint32_t max)
{
	uint32_t n = h->t->bin.n;
	for(uint32_t i = 0; i < n && i < max; i++) { out[i] = (uint64_t)h->t->bin.a[i]; }
	return n;
}
