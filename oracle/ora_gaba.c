/*
 * ora_gaba.c -- TEST INFRASTRUCTURE (see ora_gaba.h).  Plain-C restatement of the reference's
 * adaptive banded semi-global Smith-Waterman-Gotoh extension (libgaba), AFFINE and COMBINED gap
 * models, band widths 64 / 32 / 16, 2-bit ("BIT=2") sequence encoding, exactly as the reference is
 * built by Makefile.core:27-28.  Every vector operation of the x86 SIMD shim is written out as a
 * scalar loop over the W lanes with the same int8 / int16 wrap and saturation rules
 * (arch/x86_64_avx2/v64i8.h:113-235).
 *
 * Memory model: the reference lays [phantom][block]*[tail] out in a bump stack (gaba.c:308-366).
 * Here each fill owns an array blk[0..n] where blk[0] plays the phantom ("head") and carries a
 * link to the block that precedes the fill; results do not depend on the stack mechanics
 * (gaba.c:2057-2099 only decide *where* blocks live).
 *
 * The linear-gap model (gi == 0, the `ava' preset) runs the affine recurrences with gi = 0: same observable results (see og_init).
 * Not implemented: gaba_dp_merge, breakpoint masks (abrk/bbrk are always 0 in minialign's call pattern).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ora_gaba.h"

#define BLK         32
#define WMAX        64
#define MIN_BULK_BLOCKS 32          /* gaba.c:186 */
#define INIT_FETCH_APOS (-1)        /* gaba.c:194-195 */
#define INIT_FETCH_BPOS (-1)

/* block status, gaba.c:678-690 */
enum { CONT = 0, ZERO = 0x01, TERM = 0x80, STAT_MASK = ZERO | TERM | CONT, HEAD = 0x20, MERGE = 0x40, ROOT = HEAD | MERGE };
enum { MODEL_AFFINE = 1, MODEL_COMBINED = 2 };

#define MAX2(x, y)  ( (x) > (y) ? (x) : (y) )
#define MIN2(x, y)  ( (x) < (y) ? (x) : (y) )
/* test instrumentation: counts int8 wrap events of the diff-vector arithmetic (the device code may drop its sign
 * re-extension only if this stays zero; see tests/test_oracle_gaba.py::test_no_int8_wrap_in_diff_vectors) */
uint64_t og_wrap_events = 0;
static inline int8_t add8(int a, int b) { int r = a + b; if(r > 127 || r < -128) { og_wrap_events++; } return (int8_t)r; }
static inline int8_t sub8(int a, int b) { int r = a - b; if(r > 127 || r < -128) { og_wrap_events++; } return (int8_t)r; }
static inline int8_t addw8(int a, int b) { return (int8_t)(a + b); }     /* delta: wraps by design (gaba.c:1649) */
static inline int8_t subs8(int a, int b) { int r = a - b; return (int8_t)(r > 127 ? 127 : (r < -128 ? -128 : r)); }
static inline int8_t max8(int8_t a, int8_t b) { return a > b ? a : b; }
static inline uint64_t tz64(uint64_t x) { return x == 0 ? 64 : (uint64_t)__builtin_ctzll(x); }
static inline uint64_t lz64(uint64_t x) { return x == 0 ? 64 : (uint64_t)__builtin_clzll(x); }
static inline uint64_t wmask(int W) { return W == 64 ? ~0ULL : ((1ULL << W) - 1); }

/* gaba.c:308-315; the phantom (gaba.c:316-322) is stored as blk[0] of each fill with `link` set */
typedef struct og_block_s {
	uint64_t mh[BLK], mv[BLK], me[BLK], mf[BLK];
	int8_t dh[WMAX], dv[WMAX], de[WMAX], df[WMAX];
	int8_t acc, xstat, acnt, bcnt;
	uint32_t dir_mask;
	uint64_t max_mask;
	struct og_block_s *link;
} og_block_t;

/* gaba.c:351-366 */
typedef struct og_tail_s {
	uint8_t ch[WMAX];
	int8_t xd[WMAX];
	int16_t md[WMAX];
	int16_t mdrop;
	uint16_t istat;
	uint32_t pridx;
	uint32_t ridx[2], adv[2];       /* [0]: a, [1]: b */
	struct og_tail_s const *tail;
	uint8_t const *tptr[2];
	og_fill_t f;
	og_block_t *last;               /* _last_block(tail), gaba.c:323 */
	int W;
} og_tail_t;
#define TAIL_OF(_f)     ( (og_tail_t *)((uint8_t *)(_f) - offsetof(og_tail_t, f)) )
#define OFFSET_OF_TAIL(_t)  ( (_t)->f.max - (_t)->mdrop )      /* _offset(), gaba.c:374 */

struct og_ctx_s {
	int model;
	int8_t sb[16];                  /* gaba.c:3657 */
	int8_t adjh, adjv, ofsh, ofsv, gfh, gfv;   /* gaba.c:3671-3675, arch_util.h:170-204 */
	int8_t tx;
	int8_t gi, ge, gfa, gfb;
	double imx, xmx;
	og_block_t root_blk[3];
	og_tail_t root_tail[3];
};

typedef struct og_alloc_s { struct og_alloc_s *next; } og_alloc_t;

struct og_dp_s {
	og_ctx_t const *ctx;
	og_alloc_t *allocs;
	/* reader work, gaba.c:407-432 */
	struct {
		uint8_t _guard0[64];
		uint8_t bufa[WMAX + BLK + 64];
		uint8_t bufb[WMAX + BLK + 64];
		uint32_t rlim[2], id[2];
		uint8_t const *tptr[2];
		uint32_t pridx; int32_t ofsd;
		uint32_t rem[2], sridx[2];
		og_tail_t const *tail;
		int8_t xd[WMAX];
		int16_t md[WMAX];
		int W;
	} r;
	/* current fill's block array */
	og_block_t *arr; uint64_t n, cap;
	/* leaf / writer work, gaba.c:486-520 */
	struct {
		og_block_t const *blk;
		uint32_t p, q;
		int32_t gidx[2], sgidx[2];
		uint32_t ofs[2], id[2];
		og_tail_t const *tl[2];
		uint32_t state;
		uint32_t icnt[2], ecnt[2], fcnt[2];
		uint64_t ppos;
	} l;
};

/* ---- memory ---- */
static void *dp_malloc(og_dp_t *dp, size_t size)
{
	og_alloc_t *a = (og_alloc_t *)malloc(sizeof(og_alloc_t) + 16 + size);
	a->next = dp->allocs; dp->allocs = a;
	return (void *)((uint8_t *)a + 16 + sizeof(og_alloc_t) - (sizeof(og_alloc_t) % 16));
}
void og_dp_flush(og_dp_t *dp)
{
	og_alloc_t *a = dp->allocs;
	while(a) { og_alloc_t *n = a->next; free(a); a = n; }
	dp->allocs = NULL;
	if(dp->arr) { /* the current array is registered below via arr_owner */ }
	dp->arr = NULL; dp->n = dp->cap = 0;
}

/* ---- scoring helpers, gaba.c:818-838 ---- */
static int max_match(og_params_t const *p) { int m = -128; for(int i = 0; i < 16; i++) { m = MAX2(m, p->score_matrix[i]); } return m; }
static int min_match(og_params_t const *p) { int m = 127; for(int i = 0; i < 16; i++) { m = MIN2(m, p->score_matrix[i]); } return m; }
static int gap_h(og_params_t const *p, int model, int l)
{
	int aff = -1 * (l > 0) * p->gi - p->ge * l;
	return model == MODEL_COMBINED ? MAX2(aff, -1 * p->gfb * l) : aff;
}
static int gap_v(og_params_t const *p, int model, int l)
{
	int aff = -1 * (l > 0) * p->gi - p->ge * l;
	return model == MODEL_COMBINED ? MAX2(aff, -1 * p->gfa * l) : aff;
}
static int gap_e(og_params_t const *p, int l) { return -1 * (l > 0) * p->gi - p->ge * l; }
#define OFS_H(p)    ( (p)->gi + (p)->ge )
#define OFS_E(p)    ( (p)->gi )

/* gaba_init_check_score, gaba.c:3614-3640 (evaluated with W = 16, the first wrapper init, gaba_wrap.h:286) */
static int check_score(og_params_t const *p, int model)
{
	int mm = max_match(p), mn = min_match(p);
	if(mm <= 0 || mm > 6) { return -1; }
	if(mn >= 0 || mn < -7) { return -1; }
	if(mn < -2 * (p->gi + p->ge)) { return -1; }
	if(p->gfa != 0 && p->gfb != 0 && mn <= -1 * (p->gfa + p->gfb)) { return -1; }
	if(p->ge <= 0) { return -1; }
	if(p->gi < 0) { return -1; }
	if(p->gfa < 0 || (p->gfa != 0 && p->gfa <= p->ge)) { return -1; }
	if(p->gfb < 0 || (p->gfb != 0 && p->gfb <= p->ge)) { return -1; }
	if((p->gfa == 0) ^ (p->gfb == 0)) { return -1; }
	for(int i = 0; i < 16 / 2; i++) {
		int t1 = OFS_H(p) + gap_h(p, model, i*2 + 1) - gap_h(p, model, i*2);
		int t2 = OFS_H(p) + (mm + gap_v(p, model, i*2 + 1)) - gap_v(p, model, (i + 1) * 2);
		int t3 = OFS_H(p) + (mm + gap_h(p, model, i*2 + 1)) - gap_h(p, model, (i + 1) * 2);
		int t4 = OFS_H(p) + gap_h(p, model, i*2 + 1) - gap_h(p, model, i*2);
		if(MAX2(MAX2(t1, t2), MAX2(t3, t4)) > 127) { return -1; }
		if(MIN2(MIN2(t2, t2), MIN2(t3, t4)) < 0) { return -1; }
	}
	return 0;
}

/* gaba_init_phantom, gaba.c:3739-3800 + gaba_init_middle_delta :3684 + gaba_init_diff_vectors :3705 */
static void init_root(og_ctx_t *ctx, og_params_t const *p, int idx, int W)
{
	int model = ctx->model, mm = max_match(p);
	og_block_t *b = &ctx->root_blk[idx];
	og_tail_t *t = &ctx->root_tail[idx];
	memset(b, 0, sizeof(*b)); memset(t, 0, sizeof(*t));
	b->acc = 0; b->xstat = ROOT; b->acnt = 0; b->bcnt = 0; b->link = NULL;
	for(int i = 0; i < W/2; i++) {
		int dh_lo = OFS_H(p) + gap_h(p, model, i*2 + 1) - gap_h(p, model, i*2);
		int dh_hi = OFS_H(p) + mm + gap_v(p, model, i*2 + 1) - gap_v(p, model, (i + 1) * 2);
		int dv_lo = OFS_H(p) + mm + gap_h(p, model, i*2 + 1) - gap_h(p, model, (i + 1) * 2);
		int dv_hi = OFS_H(p) + gap_v(p, model, i*2 + 1) - gap_v(p, model, i*2);
		b->dh[W/2 - 1 - i] = (int8_t)dh_lo; b->dh[W/2 + i] = (int8_t)dh_hi;
		b->dv[W/2 - 1 - i] = (int8_t)dv_lo; b->dv[W/2 + i] = (int8_t)dv_hi;
		b->de[W/2 - 1 - i] = (int8_t)(OFS_E(p) + (int8_t)dv_lo + gap_e(p, i*2 + 1) - gap_h(p, model, i*2 + 1));
		b->de[W/2     + i] = (int8_t)(OFS_E(p) + (int8_t)dv_hi - p->gi);
		b->df[W/2 - 1 - i] = (int8_t)(OFS_E(p) + (int8_t)dh_lo - p->gi);
		b->df[W/2     + i] = (int8_t)(OFS_E(p) + (int8_t)dh_hi + gap_e(p, i*2 + 1) - gap_v(p, model, i*2 + 1));
	}
	for(int i = 0; i < W; i++) { b->dh[i] = sub8(0, b->dh[i]); }     /* negate dh, gaba.c:3730 */

	int64_t init_max = -(mm + gap_h(p, model, 1));
	t->f.max = init_max; t->f.status = CONT | OG_UPDATE_A | OG_UPDATE_B;
	t->f.apos = (uint64_t)(int64_t)(-W/2); t->f.bpos = (uint64_t)(int64_t)(-W/2);
	t->tail = NULL;
	t->mdrop = (int16_t)(init_max - 128);
	t->ch[0] = 0x0c; t->ch[W - 1] = 0x03 << 4;                       /* gaba.c:3781, BIT == 2 bases */
	for(int i = 0; i < W; i++) { t->xd[i] = -128; }
	for(int i = 0; i < W/2; i++) {
		t->md[W/2 - 1 - i] = (int16_t)(-(i + 1) * mm + gap_h(p, model, i*2 + 1));
		t->md[W/2     + i] = (int16_t)(-(i + 1) * mm + gap_v(p, model, i*2 + 1));
	}
	t->last = b; t->W = W;
}

og_ctx_t *og_init(og_params_t const *params)
{
	if(params == NULL) { return NULL; }
	og_params_t p = *params;
	/* gaba_wrap.h:213-221.  gi == 0 selects the reference's linear-gap build; its fills, max positions, paths, segments and counts are those of the affine
	 * recurrences run with gi = 0 (gf ignored) -- pinned against the compiled reference on random jobs over five score sets (tests/test_oracle_gaba.py,
	 * tests/golden/gaba_extend.json group `linear') -- so that is what runs */
	int model = (p.gi != 0) ? ((p.gfa != 0 && p.gfb != 0) ? MODEL_COMBINED : MODEL_AFFINE) : MODEL_AFFINE;
	if(p.gi == 0) { p.gfa = p.gfb = 0; }
	/* gaba_init_restore_default, gaba.c:3582-3608 (only xdrop matters for non-zero matrices) */
	int allzero = 1; for(int i = 0; i < 16; i++) { allzero &= p.score_matrix[i] == 0; }
	if(allzero) { return NULL; }
	if(p.xdrop == 0) { p.xdrop = 50; }
	if(check_score(&p, model) != 0) { return NULL; }

	og_ctx_t *ctx = (og_ctx_t *)calloc(1, sizeof(og_ctx_t));
	ctx->model = model;
	int ge = -p.ge, gi = -p.gi;
	for(int i = 0; i < 16; i++) { ctx->sb[i] = add8(p.score_matrix[i], -2 * (ge + gi)); }
	ctx->adjh = ctx->adjv = (int8_t)(-gi);
	ctx->ofsh = ctx->ofsv = (int8_t)(ge + gi);
	ctx->gfh = (int8_t)(-(ge + gi) - p.gfb);
	ctx->gfv = (int8_t)(-(ge + gi) - p.gfa);
	ctx->tx = (int8_t)(p.xdrop - 128);
	ctx->gi = p.gi; ctx->ge = p.ge; ctx->gfa = p.gfa; ctx->gfb = p.gfb;
	int64_t acc[2] = { 0, 0 };
	for(int i = 0; i < 16; i++) { acc[(i & 0x03) == (i >> 2)] += p.score_matrix[i]; }
	double m = (double)acc[1] / 4.0, x = (double)acc[0] / 12.0;
	ctx->imx = 1 / (m - x); ctx->xmx = x / (m - x);
	init_root(ctx, &p, 0, 64); init_root(ctx, &p, 1, 32); init_root(ctx, &p, 2, 16);
	return ctx;
}
void og_clean(og_ctx_t *ctx) { free(ctx); }
og_dp_t *og_dp_init(og_ctx_t const *ctx)
{
	og_dp_t *dp = (og_dp_t *)calloc(1, sizeof(og_dp_t));
	dp->ctx = ctx;
	return dp;
}
void og_dp_clean(og_dp_t *dp) { if(dp) { og_dp_flush(dp); free(dp); } }

/* ---- sequence fetch, gaba.c:846-1144 ---- */
static uint8_t const comp_mask_a[16]      = { 3, 2, 1, 0, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4 };
static uint8_t const shift_mask_b[16]     = { 0x00, 0x04, 0x08, 0x0c, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2 };
static uint8_t const compshift_mask_b[16] = { 0x0c, 0x08, 0x04, 0x00, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2 };
static inline uint8_t shuf16(uint8_t const *tbl, uint8_t idx) { return (idx & 0x80) ? 0 : tbl[idx & 0x0f]; }
#define REAL(_p)    ( (uint8_t const *)(OG_EOU + (uint64_t)OG_EOU - (uint64_t)(_p) - 1) )  /* _rev(), gaba.c:700 */

/* k-th base after `pos` on stream a / b, as it appears in the band buffers */
static inline uint8_t fetch_a(uint8_t const *pos, uint64_t k)
{
	if(pos < OG_EOU) { return pos[k]; }                              /* forward section: raw (gaba.c:969-976) */
	return shuf16(comp_mask_a, *REAL(pos + k));                      /* mirrored: complement (gaba.c:978-987) */
}
static inline uint8_t fetch_b(uint8_t const *pos, uint64_t k)
{
	if(pos < OG_EOU) { return shuf16(shift_mask_b, pos[k]); }        /* gaba.c:1056-1059 */
	return shuf16(compshift_mask_b, *REAL(pos + k));                 /* gaba.c:1061-1071 */
}
#define BUFA_WIN(dp)    ( (dp)->r.bufa + BLK )                       /* _rd_bufa(k, 0, W), gaba.c:806-809 */
#define BUFB_WIN(dp)    ( (dp)->r.bufb )

/* fill_fetch_core, gaba.c:1125-1144: slide both windows by (acnt, bcnt), then append look-ahead bases */
static void fetch_core(og_dp_t *dp, uint32_t acnt, uint32_t alen, uint32_t bcnt, uint32_t blen)
{
	int W = dp->r.W;
	uint8_t tmp[WMAX];
	memcpy(tmp, BUFA_WIN(dp) - acnt, W);
	uint8_t const *apos = dp->r.tptr[0] - dp->r.rem[0];
	for(uint32_t k = 0; k < alen; k++) { dp->r.bufa[BLK - 1 - k] = fetch_a(apos, k); }
	memcpy(BUFA_WIN(dp), tmp, W);
	memcpy(tmp, BUFB_WIN(dp) + bcnt, W);
	uint8_t const *bpos = dp->r.tptr[1] - dp->r.rem[1];
	for(uint32_t k = 0; k < blen; k++) { dp->r.bufb[W + k] = fetch_b(bpos, k); }
	memcpy(BUFB_WIN(dp), tmp, W);
}

/* ---- block array of the current fill ---- */
static og_block_t *arr_at(og_dp_t *dp, uint64_t i)
{
	if(i >= dp->cap) {
		/* blocks must not move once linked; allocate generously up front instead (see fill_start) */
		fprintf(stderr, "[ora_gaba] block array overflow (%lu >= %lu)\n", (unsigned long)i, (unsigned long)dp->cap);
		abort();
	}
	if(i >= dp->n) { dp->n = i + 1; }
	return &dp->arr[i];
}
static void fill_start(og_dp_t *dp, og_block_t *prev_blk)
{
	/* upper bound on #blocks: every block but the last consumes 32 bases in total */
	uint64_t tot = (uint64_t)dp->r.rem[0] + (uint64_t)dp->r.rem[1];
	uint64_t lim = 2 * (uint64_t)MIN2(dp->r.rem[0], dp->r.rem[1]) + 8192;     /* the band cannot run far along one axis before X-drop */
	uint64_t cap = MIN2(tot, lim) / BLK + 8;
	dp->arr = (og_block_t *)dp_malloc(dp, sizeof(og_block_t) * cap);
	dp->cap = cap; dp->n = 1;
	/* fill_create_phantom, gaba.c:1315-1333 */
	og_block_t *ph = &dp->arr[0];
	memcpy(ph->dh, prev_blk->dh, 4 * WMAX);
	ph->acc = prev_blk->acc;
	ph->xstat = (int8_t)((prev_blk->xstat & ROOT) | HEAD);
	ph->acnt = 0; ph->bcnt = 0;
	ph->dir_mask = 0; ph->max_mask = 0;
	ph->link = prev_blk;
}

/* ---- band registers ---- */
typedef struct {
	int8_t dh[WMAX], dv[WMAX], de[WMAX], df[WMAX], delta[WMAX], drop[WMAX];
	uint32_t dmask; int32_t dacc;
	uint8_t const *aptr, *bptr;
} regs_t;

/* _fill_load_context, gaba.c:1527-1552 */
static void load_context(og_dp_t *dp, og_block_t const *prev, regs_t *g)
{
	g->aptr = BUFA_WIN(dp); g->bptr = BUFB_WIN(dp);
	memcpy(g->dh, prev->dh, WMAX); memcpy(g->dv, prev->dv, WMAX);
	memcpy(g->de, prev->de, WMAX); memcpy(g->df, prev->df, WMAX);
	memset(g->delta, 0, WMAX);
	memcpy(g->drop, dp->r.xd, WMAX);
	g->dmask = 0; g->dacc = prev->acc;                               /* _dir_init, gaba.c:749 */
}

/* one anti-diagonal: _dir_fetch (gaba.c:753) must have been applied by the caller.
 * _fill_right / _fill_down (gaba.c:1666-1699) + _fill_body (gaba.c:1576-1641) + _fill_update_delta (gaba.c:1647-1655) */
static void fill_vector(og_dp_t *dp, regs_t *g, int down, uint64_t m[4])
{
	og_ctx_t const *c = dp->ctx;
	int W = dp->r.W;
	int8_t t[WMAX];
	if(!down) {
		g->aptr--;
		for(int l = W - 1; l > 0; l--) { g->dh[l] = g->dh[l - 1]; g->df[l] = g->df[l - 1]; }
		g->dh[0] = 0; g->df[0] = 0;                                   /* _bsl_n, v64i8.h:152 */
	} else {
		g->bptr++;
		for(int l = 0; l < W - 1; l++) { g->dv[l] = g->dv[l + 1]; g->de[l] = g->de[l + 1]; }
		g->dv[W - 1] = 0; g->de[W - 1] = 0;                           /* _bsr_n, v64i8.h:164 */
	}
	uint64_t mh = 0, mv = 0, me = 0, mf = 0;
	for(int l = 0; l < W; l++) {
		int8_t dh = g->dh[l], dv = g->dv[l], de = g->de[l], df = g->df[l];
		int8_t s = (int8_t)shuf16((uint8_t const *)c->sb, (uint8_t)(g->aptr[l] | g->bptr[l]));
		uint64_t bit = 1ULL << l;
		int8_t tt;
		if(c->model == MODEL_COMBINED) {
			int8_t dfh = add8(dv, c->gfh), dfv = sub8(c->gfv, dh);
			int8_t ss = max8(de, df);
			ss = max8(ss, dfh);
			tt = max8(s, dfv);
			tt = max8(tt, ss);
			uint64_t gfh = tt == dfh, gh = tt == de, gfv = tt == dfv, gv = tt == df;
			if(gfh | gh) { mh |= bit; }
			gh &= ~gfh;
			if(gfv | gv) { mv |= bit; }
			gv &= ~gfv;
			de = add8(de, c->adjh);
			int8_t te = max8(de, tt);
			if(gh | (te == tt)) { me |= bit; }
			de = add8(te, dh);
			dh = add8(dh, tt);
			df = add8(df, c->adjv);
			int8_t tf = max8(df, tt);
			if(gv | (tf == tt)) { mf |= bit; }
			df = sub8(tf, dv);
		} else {
			tt = max8(de, s);
			tt = max8(df, tt);
			if(tt == de) { mh |= bit; }
			if(tt == df) { mv |= bit; }
			de = add8(de, c->adjh);
			int8_t te = max8(de, tt);
			if(te == tt) { me |= bit; }
			de = add8(te, dh);
			dh = add8(dh, tt);
			df = add8(df, c->adjv);
			int8_t tf = max8(df, tt);
			if(tf == tt) { mf |= bit; }
			df = sub8(tf, dv);
		}
		int8_t t2 = sub8(dv, tt);
		g->dv[l] = dh; g->dh[l] = t2; g->de[l] = de; g->df[l] = df;
		t[l] = !down ? sub8(c->ofsh, t2) : add8(c->ofsv, dh);         /* uses the *new* dh / dv */
	}
	for(int l = 0; l < W; l++) {
		g->delta[l] = addw8(g->delta[l], t[l]);
		g->drop[l] = subs8(g->drop[l], t[l]);
	}
	g->dacc += (int32_t)t[0] - (int32_t)t[W - 1];                    /* _dir_update, gaba.c:761 */
	m[0] = mh; m[1] = mv; m[2] = me; m[3] = mf;
}

/* _fill_store_context, gaba.c:1734-1778 */
static void store_context(og_dp_t *dp, og_block_t *blk, regs_t *g)
{
	int W = dp->r.W;
	memcpy(blk->dh, g->dh, WMAX); memcpy(blk->dv, g->dv, WMAX);
	memcpy(blk->de, g->de, WMAX); memcpy(blk->df, g->df, WMAX);
	blk->dir_mask = g->dmask; blk->acc = (int8_t)g->dacc;
	blk->xstat = (int8_t)(((int)dp->ctx->tx - (int)g->drop[W/2]) & TERM);
	int32_t cofs = g->delta[W/2];
	int32_t acnt = (int32_t)(BUFA_WIN(dp) - g->aptr), bcnt = (int32_t)(g->bptr - BUFB_WIN(dp));
	blk->acnt = (int8_t)acnt; blk->bcnt = (int8_t)bcnt;
	dp->r.ofsd += cofs; dp->r.rem[0] -= acnt; dp->r.rem[1] -= bcnt;
	uint64_t mm = 0;
	for(int l = 0; l < W; l++) {
		int8_t prev_drop = dp->r.xd[l];
		if(addw8(g->drop[l], g->delta[l]) > prev_drop) { mm |= 1ULL << l; }
	}
	blk->max_mask = mm;
	cofs += 0x0100;
	for(int l = 0; l < W; l++) {
		int8_t drop = g->drop[l], delta = g->delta[l];
		int16_t md = dp->r.md[l];
		md = (int16_t)(md + (int16_t)delta);
		int8_t ov = (int8_t)(~addw8(drop, delta) & (drop & delta));          /* _andn_n(a, b) = ~a & b */
		md = (int16_t)(md + (0x0100 & (int16_t)ov));
		int8_t uv = (int8_t)(subs8(delta, 0x40) | drop);
		md = (int16_t)(md + (0x0100 & (int16_t)uv));
		md = (int16_t)(md + (int16_t)(-cofs));
		dp->r.md[l] = md;
		dp->r.xd[l] = drop;
	}
}

/* fill_bulk_block, gaba.c:1821-1860 */
static void fill_bulk_block(og_dp_t *dp, uint64_t i)
{
	og_block_t *blk = arr_at(dp, i), *prev = &dp->arr[i - 1];
	fetch_core(dp, prev->acnt, BLK, prev->bcnt, BLK);
	regs_t g; load_context(dp, prev, &g);
	for(int k = 0; k < BLK; k++) {
		g.dmask = (g.dmask << 1) | (uint32_t)(g.dacc < 0);            /* _dir_fetch */
		uint64_t m[4];
		fill_vector(dp, &g, g.dmask & 1, m);
		blk->mh[k] = m[0]; blk->mv[k] = m[1]; blk->me[k] = m[2]; blk->mf[k] = m[3];
	}
	dp->r.pridx -= BLK;
	store_context(dp, blk, &g);
}

/* fill_cap_seq_bounded, gaba.c:1925-1975 */
static uint64_t fill_cap_seq_bounded(og_dp_t *dp, uint64_t i)
{
	while(dp->arr[i].xstat >= 0) {
		og_block_t *prev = &dp->arr[i]; i++;
		og_block_t *blk = arr_at(dp, i); prev = &dp->arr[i - 1];
		uint32_t alen = MIN2(dp->r.rem[0], BLK), blen = MIN2(dp->r.rem[1], BLK);    /* fill_cap_fetch, gaba.c:1150 */
		fetch_core(dp, prev->acnt, alen, prev->bcnt, blen);
		int64_t arem = dp->r.rem[0], brem = dp->r.rem[1], prem = dp->r.pridx;
		regs_t g; load_context(dp, prev, &g);
		int k = 0;
		while(k < BLK) {
			g.dmask = (g.dmask << 1) | (uint32_t)(g.dacc < 0);
			int down = g.dmask & 1;
			/* _fill_cap_test_idx, gaba.c:1800-1809, evaluated after the pointer update */
			int64_t ac = (int64_t)(BUFA_WIN(dp) - g.aptr) + (down ? 0 : 1);
			int64_t bc = (int64_t)(g.bptr - BUFB_WIN(dp)) + (down ? 1 : 0);
			int64_t ta = arem - ac, tb = brem - bc, tp = tb + ta + prem;
			if((ta | tb | tp) < 0) { g.dmask >>= 1; break; }          /* windback */
			uint64_t m[4];
			fill_vector(dp, &g, down, m);
			blk->mh[k] = m[0]; blk->mv[k] = m[1]; blk->me[k] = m[2]; blk->mf[k] = m[3];
			k++;
		}
		dp->r.pridx -= (uint32_t)k;
		g.dmask = (k == 0) ? g.dmask : (g.dmask << (BLK - k));        /* _dir_adjust_remainder (x86 shl masks the count) */
		store_context(dp, blk, &g);
		if(k != BLK) { break; }
	}
	return i;
}

/* fill_seq_bounded, gaba.c:2027-2051 (stack-bounded variant gaba.c:2057 only relocates blocks) */
static uint64_t fill_section(og_dp_t *dp, uint64_t i)
{
	#define _min_blocks()   ( MIN2((uint64_t)MIN2(dp->r.rem[0], dp->r.rem[1]), (uint64_t)dp->r.pridx) / BLK )
	uint64_t cnt;
	while((cnt = _min_blocks()) > MIN_BULK_BLOCKS) {
		/* fill_bulk_k_blocks, gaba.c:1873-1892 */
		uint64_t t = i + cnt;
		while(((int64_t)dp->arr[i].xstat | (int64_t)(t - i)) > 0) { fill_bulk_block(dp, ++i); }
		if((dp->arr[i].xstat & STAT_MASK) != CONT) { return i; }
	}
	/* fill_bulk_seq_bounded, gaba.c:1898-1913 */
	while(1) {
		int64_t test = (int64_t)((uint64_t)dp->r.rem[0] - BLK) | (int64_t)((uint64_t)dp->r.rem[1] - BLK) | (int64_t)((uint64_t)dp->r.pridx - BLK);
		if(((int64_t)dp->arr[i].xstat | test) < 0) { break; }
		fill_bulk_block(dp, ++i);
	}
	if((dp->arr[i].xstat & STAT_MASK) != CONT) { return i; }
	return fill_cap_seq_bounded(dp, i);
	#undef _min_blocks
}

/* fill_load_section, gaba.c:1269-1308 (breakpoint masks are zero: rem = ridx, rlim = 0) */
static void load_section(og_dp_t *dp, og_tail_t const *tail, og_section_t const *a, og_section_t const *b, uint32_t pridx)
{
	og_section_t const *s[2] = { a, b };
	for(int i = 0; i < 2; i++) {
		uint32_t ridx = tail->ridx[i] == 0 ? s[i]->len : tail->ridx[i];
		dp->r.rlim[i] = 0; dp->r.id[i] = s[i]->id;
		dp->r.tptr[i] = s[i]->base + s[i]->len;
		dp->r.rem[i] = ridx; dp->r.sridx[i] = ridx;
	}
	dp->r.pridx = pridx; dp->r.ofsd = 0;
	dp->r.tail = tail;
}

/* fill_load_vectors, gaba.c:1376-1399 */
static void load_vectors(og_dp_t *dp, og_tail_t const *tail)
{
	int W = tail->W;
	dp->r.W = W;
	memset(dp->r.bufa, 0, sizeof(dp->r.bufa)); memset(dp->r.bufb, 0, sizeof(dp->r.bufb));
	for(int l = 0; l < W; l++) {
		BUFA_WIN(dp)[l] = tail->ch[l] & 0x0f;
		BUFB_WIN(dp)[l] = (tail->ch[l] >> 4) & 0x0f;
	}
	memcpy(dp->r.xd, tail->xd, WMAX); memcpy(dp->r.md, tail->md, sizeof(int16_t) * WMAX);
	fill_start(dp, tail->last);
}

/* fill_init_fetch, gaba.c:1168-1210; returns bpos after the fetch */
static int64_t init_fetch(og_dp_t *dp, og_block_t *blk, int64_t apos, int64_t bpos)
{
	int32_t irem[2] = { (int32_t)(INIT_FETCH_APOS - (int32_t)apos), (int32_t)(INIT_FETCH_BPOS - (int32_t)bpos) };
	int32_t srem[2] = { (int32_t)dp->r.rem[0], (int32_t)dp->r.rem[1] };
	int32_t adj[2] = { 1, 0 };
	int32_t len[2];
	for(int i = 0; i < 2; i++) {
		int32_t o = (srem[1 - i] - irem[1 - i]) + (adj[i] + irem[i]);
		len[i] = MIN2(MIN2(irem[i], srem[i]), o);
	}
	fetch_core(dp, 0, (uint32_t)len[0], 0, (uint32_t)len[1]);
	blk->acnt = (int8_t)len[0]; blk->bcnt = (int8_t)len[1];
	dp->r.rem[0] = (uint32_t)(srem[0] - len[0]); dp->r.rem[1] = (uint32_t)(srem[1] - len[1]);
	return bpos + len[1];
}

/* fill_create_tail, gaba.c:1406-1499 */
static og_fill_t *create_tail(og_dp_t *dp, uint64_t i)
{
	int W = dp->r.W;
	og_block_t *blk = &dp->arr[i];
	uint32_t acnt = (uint8_t)blk->acnt, bcnt = (uint8_t)blk->bcnt;
	uint32_t xstat = (uint32_t)(int32_t)blk->xstat;
	og_tail_t *tail = (og_tail_t *)dp_malloc(dp, sizeof(og_tail_t));
	memset(tail, 0, sizeof(*tail));
	tail->W = W;
	if(acnt != 0 || bcnt != 0) { tail->last = blk; }
	else { tail->last = i > 0 ? &dp->arr[i - 1] : blk->link; }      /* squash the empty block, gaba.c:1492 */

	/* fill_save_vectors */
	uint8_t const *ach = BUFA_WIN(dp) - acnt, *bch = BUFB_WIN(dp) + bcnt;
	int32_t mdrop = -32768;
	for(int l = 0; l < W; l++) {
		tail->ch[l] = (uint8_t)(ach[l] | (bch[l] << 4));
		tail->xd[l] = dp->r.xd[l]; tail->md[l] = dp->r.md[l];
		int16_t v = (int16_t)(dp->r.md[l] + (int16_t)dp->r.xd[l]);
		mdrop = MAX2(mdrop, (int32_t)v);
	}
	/* fill_save_section */
	og_tail_t const *prev = dp->r.tail;
	tail->mdrop = (int16_t)mdrop; tail->istat = 0; tail->pridx = dp->r.pridx;
	tail->tail = prev;
	for(int k = 0; k < 2; k++) {
		uint32_t ridx = dp->r.rem[k] + dp->r.rlim[k];
		uint32_t adv = dp->r.sridx[k] - ridx;
		tail->ridx[k] = ridx; tail->adv[k] = adv;
		tail->tptr[k] = dp->r.tptr[k] + dp->r.rlim[k];
	}
	tail->f.aid = dp->r.id[0]; tail->f.bid = dp->r.id[1];
	tail->f.ascnt = prev->f.ascnt + (tail->ridx[0] == 0);
	tail->f.bscnt = prev->f.bscnt + (tail->ridx[1] == 0);
	tail->f.apos = prev->f.apos + (uint64_t)(int64_t)(int32_t)tail->adv[0];
	tail->f.bpos = prev->f.bpos + (uint64_t)(int64_t)(int32_t)tail->adv[1];
	tail->f.max = OFFSET_OF_TAIL(prev) + dp->r.ofsd + mdrop;
	tail->f.status = ((xstat & (TERM | CONT)) << 8)
		| (tail->ridx[0] == 0 ? OG_UPDATE_A : 0) | (tail->ridx[1] == 0 ? OG_UPDATE_B : 0);
	return &tail->f;
}

og_fill_t *og_dp_fill_root(og_dp_t *dp, int bw_idx, og_section_t const *a, uint32_t apos, og_section_t const *b, uint32_t bpos, uint32_t pridx)
{
	og_tail_t const *root = &dp->ctx->root_tail[bw_idx];
	/* fill_create_bridge, gaba.c:1339-1370 */
	og_tail_t *brg = (og_tail_t *)dp_malloc(dp, sizeof(og_tail_t));
	memcpy(brg, root, sizeof(og_tail_t));
	brg->istat = root->istat | 1;
	brg->ridx[0] = a->len - apos; brg->ridx[1] = b->len - bpos;
	brg->adv[0] = apos; brg->adv[1] = bpos;
	brg->tail = root;
	brg->tptr[0] = a->base + a->len; brg->tptr[1] = b->base + b->len;
	brg->f.aid = a->id; brg->f.bid = b->id;
	brg->last = NULL;

	load_section(dp, brg, a, b, pridx == 0 ? UINT32_MAX : pridx);
	load_vectors(dp, root);
	if(init_fetch(dp, &dp->arr[0], (int64_t)root->f.apos, (int64_t)root->f.bpos) < INIT_FETCH_BPOS) {
		return create_tail(dp, 0);
	}
	return create_tail(dp, fill_section(dp, 0));
}

og_fill_t *og_dp_fill(og_dp_t *dp, og_fill_t const *fill, og_section_t const *a, og_section_t const *b, uint32_t pridx)
{
	og_tail_t const *tail = TAIL_OF(fill);
	load_section(dp, tail, a, b, pridx == 0 ? tail->pridx : pridx);
	load_vectors(dp, tail);
	if((int64_t)tail->f.bpos < INIT_FETCH_BPOS) {
		if(init_fetch(dp, &dp->arr[0], (int64_t)tail->f.apos, (int64_t)tail->f.bpos) < INIT_FETCH_BPOS) {
			return create_tail(dp, 0);
		}
	}
	return create_tail(dp, fill_section(dp, 0));
}

/* ---- max search, gaba.c:2604-2817 ---- */
/* fill_restore_fetch, gaba.c:1217-1264 */
static void restore_fetch(og_dp_t *dp, og_tail_t const *tail, int32_t const ridx[2])
{
	int W = tail->W;
	dp->r.W = W;
	og_tail_t const *prev_tail = tail->tail;
	memset(dp->r.bufa, 0, sizeof(dp->r.bufa)); memset(dp->r.bufb, 0, sizeof(dp->r.bufb));
	int32_t ofs[2], len[2], cridx[2];
	for(int k = 0; k < 2; k++) {
		int32_t sridx = (int32_t)(tail->ridx[k] + tail->adv[k]);
		int32_t dridx = ridx[k] + W;
		cridx[k] = MIN2(dridx, sridx);
		ofs[k] = dridx - cridx[k];
		len[k] = MIN2(cridx[k], W + BLK - ofs[k]);
	}
	/* a: stream offset t lands at _rd_bufa(ofs + t, 1); the first `ofs` lanes come from the previous tail */
	uint8_t const *apos = tail->tptr[0] - cridx[0];
	for(int32_t t = 0; t < len[0]; t++) { dp->r.bufa[BLK + W - ofs[0] - 1 - t] = fetch_a(apos, (uint64_t)t); }
	for(int32_t i = 0; i < ofs[0] && i < W; i++) { dp->r.bufa[BLK + W - ofs[0] + i] = prev_tail->ch[i] & 0x0f; }
	/* b */
	for(int32_t i = 0; i < ofs[1] && i < W; i++) { dp->r.bufb[i] = (prev_tail->ch[W - ofs[1] + i] >> 4) & 0x0f; }
	uint8_t const *bpos = tail->tptr[1] - cridx[1];
	for(int32_t t = 0; t < len[1]; t++) { dp->r.bufb[ofs[1] + t] = fetch_b(bpos, (uint64_t)t); }
}

/* leaf_search, gaba.c:2708-2770; returns plen */
static uint64_t leaf_search(og_dp_t *dp, og_tail_t const *tail)
{
	int W = tail->W;
	/* leaf_load_max_mask, gaba.c:2609-2631 */
	uint64_t max_mask = 0;
	for(int l = 0; l < W; l++) {
		if((int16_t)(tail->md[l] + (int16_t)tail->xd[l]) == tail->mdrop) { max_mask |= 1ULL << l; }
	}
	og_block_t const *b = tail->last + 1;
	int32_t ridx[2] = { (int32_t)tail->ridx[0], (int32_t)tail->ridx[1] };
	while(1) {
		--b;
		if((b->xstat & ROOT) == ROOT) { return 0; }
		while(b->xstat & HEAD) { b = b->link; }
		ridx[0] += (int32_t)b->acnt; ridx[1] += (int32_t)b->bcnt;
		if((max_mask & ~b->max_mask) == 0) { break; }
		max_mask &= ~b->max_mask;
	}
	restore_fetch(dp, tail, ridx);

	/* leaf_detect_pos, gaba.c:2663-2700: refill the block recording cell-wise update masks */
	uint64_t marr[BLK]; int n = 0;
	{
		og_block_t const *prev = b - 1;                                /* (blk - 1): previous block or the head */
		regs_t g; load_context(dp, prev, &g);
		int8_t mx[WMAX]; memcpy(mx, g.delta, WMAX);
		int cnt = (int)b->acnt + (int)b->bcnt;
		for(int i = 0; i < cnt; i++) {
			g.dmask = (g.dmask << 1) | (uint32_t)(g.dacc < 0);
			uint64_t m[4];
			fill_vector(dp, &g, g.dmask & 1, m);
			uint64_t um = 0;
			for(int l = 0; l < W; l++) { if(g.delta[l] > mx[l]) { um |= 1ULL << l; } mx[l] = max8(g.delta[l], mx[l]); }
			marr[n++] = um;
		}
	}
	/* leaf_search_pos, gaba.c:2636-2652 */
	int mi = n;
	while(mi > 0 && (max_mask & ~marr[--mi]) != 0) { max_mask &= ~marr[mi]; }
	dp->l.p = (uint32_t)mi;
	dp->l.q = (uint32_t)tz64((n > 0 ? marr[mi] : 0) & max_mask);
	dp->l.blk = b;

	int64_t fcnt = dp->l.p + 1;
	uint32_t dir_mask = b->dir_mask >> (BLK - fcnt);
	int32_t pc = __builtin_popcount(dir_mask);
	ridx[0] -= (int32_t)((fcnt - pc) - (1 + (int32_t)dp->l.q));
	ridx[1] -= (int32_t)((0 + pc) - (W - (int32_t)dp->l.q));
	for(int k = 0; k < 2; k++) {
		int32_t gidx = 1 - ridx[k] + (int32_t)tail->ridx[k];
		dp->l.gidx[k] = gidx; dp->l.sgidx[k] = gidx;
	}
	int32_t rem0 = ridx[0] - (int32_t)tail->ridx[0], rem1 = ridx[1] - (int32_t)tail->ridx[1];
	uint64_t plen = tail->f.apos + tail->f.bpos - (uint64_t)(int64_t)(INIT_FETCH_APOS + INIT_FETCH_BPOS) + (uint64_t)W
		- (uint64_t)(int64_t)rem1 - (uint64_t)(int64_t)rem0;
	return plen;
}

/* gaba_dp_search_max, gaba.c:2776-2817 */
og_pos_pair_t *og_dp_search_max(og_dp_t *dp, og_fill_t const *fill)
{
	og_tail_t const *tail = TAIL_OF(fill);
	og_pos_pair_t *pos = (og_pos_pair_t *)dp_malloc(dp, sizeof(og_pos_pair_t));
	pos->plen = leaf_search(dp, tail);
	int32_t gidx[2] = { dp->l.gidx[0], dp->l.gidx[1] }, acc[2] = { 0, 0 };
	uint32_t id[2] = { tail->f.aid, tail->f.bid };
	while(tail->tail != NULL) {
		int upd[2] = { 1 > gidx[0], 1 > gidx[1] };
		if(!upd[0] && !upd[1]) { break; }
		uint32_t nid[2] = { tail->f.aid, tail->f.bid };
		acc[0] += (int32_t)tail->adv[0]; acc[1] += (int32_t)tail->adv[1];
		tail = tail->tail;
		for(int k = 0; k < 2; k++) {
			int mask = upd[k] && (tail->ridx[k] == 0);
			if(mask) { gidx[k] += acc[k]; id[k] = nid[k]; acc[k] = 0; }
		}
	}
	pos->aid = id[0]; pos->bid = id[1];
	pos->apos = (uint32_t)gidx[0]; pos->bpos = (uint32_t)gidx[1];
	return pos;
}

/* ---- traceback, gaba.c:2820-3407 ---- */
enum { TS_H = 1, TS_V = 2, TS_S = 4, ts_d = TS_H | TS_V, ts_v0 = TS_V, ts_v1 = TS_V | TS_S, ts_h0 = TS_H, ts_h1 = TS_H | TS_S };

/* trace_reload_section, gaba.c:2826-2860 */
static void trace_reload_section(og_dp_t *dp, int i)
{
	og_tail_t const *tail = dp->l.tl[i], *prev_tail = tail;
	int32_t gidx = dp->l.gidx[i];
	while(gidx <= 0) {
		do {
			gidx += tail->istat ? 0 : (int32_t)tail->adv[i];
			prev_tail = tail; tail = tail->tail;
		} while(tail->ridx[i] != 0);
	}
	dp->l.tl[i] = tail;
	dp->l.id[i] = i == 0 ? prev_tail->f.aid : prev_tail->f.bid;
	dp->l.ofs[i] = prev_tail->istat ? prev_tail->adv[i] : 0;
	dp->l.gidx[i] = gidx; dp->l.sgidx[i] = gidx;
}

typedef struct {
	og_dp_t *dp; int W; int model;
	og_block_t const *blk; int32_t p; uint32_t q, save, dir_mask; int bulk;
	int32_t gidx[2];
	uint32_t *path; uint64_t ppos;
	int oob;            /* out-of-band exit taken in the bulk loop (gaba.c:3062) */
} tr_t;

/* x86 shift-count masking of the (mask >> q) tests, gaba.c:2931-2951 */
static inline uint64_t bit_at(uint64_t m, uint32_t q, int W) { return W == 64 ? (m >> (q & 63)) & 1 : ((q & 31) >= 32 ? 0 : ((uint64_t)(uint32_t)m >> (q & 31)) & 1); }
#define MH(t)   bit_at((t)->blk->mh[(t)->p], (t)->q, (t)->W)
#define MV(t)   bit_at((t)->blk->mv[(t)->p], (t)->q, (t)->W)
#define ME(t)   bit_at((t)->blk->me[(t)->p], (t)->q, (t)->W)
#define MF(t)   bit_at((t)->blk->mf[(t)->p], (t)->q, (t)->W)
static inline int test_diag_h(tr_t *t) { return MH(t) == 0; }
static inline int test_diag_v(tr_t *t) { return MV(t) == 0; }
static inline int test_gap_h(tr_t *t) { return t->model == MODEL_COMBINED ? bit_at(~t->blk->mh[t->p] & t->blk->me[t->p], t->q, t->W) == 0 : ME(t) == 0; }
static inline int test_gap_v(tr_t *t) { return t->model == MODEL_COMBINED ? bit_at(~t->blk->mv[t->p] & t->blk->mf[t->p], t->q, t->W) == 0 : MF(t) == 0; }
static inline int test_fgap_h(tr_t *t) { return t->model == MODEL_COMBINED ? ME(t) == 0 : 0; }
static inline int test_fgap_v(tr_t *t) { return t->model == MODEL_COMBINED ? MF(t) == 0 : 0; }

/* _trace_test_bulk, gaba.c:3035-3046 */
static int trace_test_bulk(tr_t *t)
{
	int32_t ga = t->gidx[0] - (int32_t)t->blk->acnt, gb = t->gidx[1] - (int32_t)t->blk->bcnt;
	if(!(t->W > ga) && !(t->W > gb)) { t->gidx[0] = ga; t->gidx[1] = gb; return 1; }
	return 0;
}
/* _trace_reload_block (gaba.c:3020-3031) / _trace_reload_tail (gaba.c:3000-3016): step to the previous
 * block, hopping over head (phantom) blocks; both macros leave (blk, mask index, dir_mask) in the same state */
static void trace_reload(tr_t *t)
{
	og_block_t const *blk = t->blk - 1;
	while(blk != NULL && (blk->xstat & HEAD) != 0) { blk = blk->link; }
	if(blk == NULL) { t->blk = NULL; t->p = -1; t->dir_mask = 0; return; }   /* fell off the root; never dereferenced afterwards */
	int cnt = (int)blk->acnt + (int)blk->bcnt;
	t->p = cnt - 1; t->dir_mask = blk->dir_mask >> (BLK - cnt);
	t->blk = blk;
}
#define TRACE_HEAD_CNT(_W)  ( (uint32_t)((_W) / BLK + ((_W) == 16)) )        /* gaba.c:3051 */

/* _pop_vector (gaba.c:3114-3122) with _trace_{bulk,tail}_load_n (gaba.c:3052-3089);
 * returns 1 when the walk must stop immediately (out-of-band exit of the bulk loop) */
static int trace_pop(tr_t *t, int is_v)
{
	if(!t->bulk) { t->gidx[is_v]--; }                                 /* _trace_tail_*_update_index */
	t->ppos--;
	if(is_v) { t->path[t->ppos >> 5] |= 1u << (t->ppos & 31); }
	t->q += (t->dir_mask & 1) - (uint32_t)is_v;
	t->dir_mask >>= 1;
	t->p--;
	if(t->p >= 0) { return 0; }
	if(getenv("OG_DEBUG_TRACE")) { fprintf(stderr, "blk ppos %lu q %d bulk %d g %d %d save %u\n", (unsigned long)t->ppos, (int)t->q, t->bulk, t->gidx[0], t->gidx[1], t->save); }
	if(t->bulk) {
		trace_reload(t);
		if(!trace_test_bulk(t)) {
			if(t->q >= (uint32_t)t->W) { t->oob = 1; return 1; }
			t->gidx[1] += (int32_t)(t->q - t->save);
			t->gidx[0] += (int32_t)(t->save - t->q);
			t->save = TRACE_HEAD_CNT(t->W);
			t->bulk = 0;
		}
	} else {
		if((t->blk - 1)->xstat & HEAD) {
			trace_reload(t);
		} else {
			trace_reload(t);
			t->save--;
			if(t->save >= TRACE_HEAD_CNT(t->W) && trace_test_bulk(t)) { t->save = t->q; t->bulk = 1; }
		}
	}
	return 0;
}

/* trace_core, gaba.c:3111-3228, restated as an explicit state machine over the same labels */
enum { L_D_HEAD, L_D_MID, L_D_TAIL, L_H_HEAD, L_H_LOOP, L_H_TAIL, L_V_HEAD, L_V_LOOP, L_V_TAIL };
static void trace_core(og_dp_t *dp, tr_t *t)
{
	int lbl;
	t->bulk = 0; t->save = TRACE_HEAD_CNT(t->W); t->oob = 0;
	t->blk = dp->l.blk; t->p = (int32_t)dp->l.p; t->q = dp->l.q;
	t->dir_mask = t->blk->dir_mask >> (BLK - (t->p + 1));
	t->gidx[0] = dp->l.gidx[0]; t->gidx[1] = dp->l.gidx[1];
	switch(dp->l.state) {
		case ts_d:  lbl = L_D_HEAD; break;
		case ts_v0: lbl = L_V_HEAD; break;
		case ts_v1: lbl = L_V_TAIL; break;
		case ts_h0: lbl = L_H_HEAD; break;
		case ts_h1: lbl = L_H_TAIL; break;
		default: return;
	}
	#define IDX0(_k)    ( !t->bulk && t->gidx[_k] == 0 )
	while(1) {
		switch(lbl) {
		case L_D_HEAD:
			if(!test_diag_h(t)) { lbl = L_H_HEAD; break; }
			if(!t->bulk && (t->gidx[0] == 0 || t->gidx[1] == 0)) { dp->l.state = ts_d; goto _term; }
			if(trace_pop(t, 0)) { goto _term; }
			lbl = L_D_MID; break;
		case L_D_MID:
			if(trace_pop(t, 1)) { goto _term; }
			lbl = L_D_TAIL; break;
		case L_D_TAIL:
			lbl = !test_diag_v(t) ? L_V_HEAD : L_D_HEAD; break;
		case L_H_HEAD:
			if(test_fgap_h(t)) {
				if(IDX0(0)) { dp->l.state = ts_h0; goto _term; }
				dp->l.fcnt[0]++;
				if(trace_pop(t, 0)) { goto _term; }
				lbl = L_D_HEAD; break;
			}
			dp->l.icnt[0]++;
			lbl = L_H_LOOP; break;
		case L_H_LOOP:
			if(IDX0(0)) { dp->l.state = ts_h1; goto _term; }
			dp->l.ecnt[0]++;
			if(trace_pop(t, 0)) { goto _term; }
			lbl = L_H_TAIL; break;
		case L_H_TAIL:
			lbl = test_gap_h(t) ? L_H_LOOP : L_D_HEAD; break;
		case L_V_HEAD:
			if(test_fgap_v(t)) {
				if(IDX0(1)) { dp->l.state = ts_v0; goto _term; }
				dp->l.fcnt[1]++;
				if(trace_pop(t, 1)) { goto _term; }
				lbl = L_D_TAIL; break;
			}
			dp->l.icnt[1]++;
			lbl = L_V_LOOP; break;
		case L_V_LOOP:
			if(IDX0(1)) { dp->l.state = ts_v1; goto _term; }
			dp->l.ecnt[1]++;
			if(trace_pop(t, 1)) { goto _term; }
			lbl = L_V_TAIL; break;
		case L_V_TAIL:
			lbl = test_gap_v(t) ? L_V_LOOP : L_D_TAIL; break;
		}
	}
_term:
	dp->l.blk = t->blk; dp->l.p = (uint32_t)t->p; dp->l.q = t->q;
	dp->l.gidx[0] = t->gidx[0]; dp->l.gidx[1] = t->gidx[1];
	#undef IDX0
}

/* gaba_dp_trace, gaba.c:3372 -> trace_body :3299 -> trace_init :3244, trace_push_segment :2865 */
og_alignment_t *og_dp_trace(og_dp_t *dp, og_fill_t const *fill)
{
	og_tail_t const *tail = TAIL_OF(fill);
	uint64_t plen = (int64_t)fill->bpos < INIT_FETCH_BPOS ? 0 : leaf_search(dp, tail);

	uint64_t sn = (uint64_t)tail->f.ascnt + tail->f.bscnt + 2, pn = (plen + 31) / 32 + 2;
	og_alignment_t *aln = (og_alignment_t *)calloc(1, sizeof(og_alignment_t));
	/* two words precede path[] in gaba_alignment_s (gaba.h:217: plen, padding = 0x40000000, gaba.c:3272); the
	 * reverse CIGAR parser peeks at them, so keep the same bits there */
	uint32_t *pbase = (uint32_t *)calloc(pn + 8 + 2, sizeof(uint32_t));
	pbase[0] = (uint32_t)plen; pbase[1] = 0x40000000;
	aln->path = pbase + 2;
	og_segment_t *segbuf = (og_segment_t *)calloc(sn + 64, sizeof(og_segment_t));
	uint64_t segcap = sn + 64, slen = 0;

	dp->l.tl[0] = tail; dp->l.tl[1] = tail;
	memset(dp->l.icnt, 0, sizeof(dp->l.icnt)); memset(dp->l.ecnt, 0, sizeof(dp->l.ecnt)); memset(dp->l.fcnt, 0, sizeof(dp->l.fcnt));
	dp->l.state = ts_d;
	aln->path[plen >> 5] = 1u << (plen & 31);                        /* sentinel bit just past the path */

	tr_t t; memset(&t, 0, sizeof(t));
	t.dp = dp; t.W = tail->W; t.model = dp->ctx->model; t.path = aln->path; t.ppos = plen;
	while(t.ppos > 0) {
		if(dp->l.gidx[0] < (int32_t)((dp->l.state & TS_H) != 0)) { trace_reload_section(dp, 0); }
		if(dp->l.gidx[1] < (int32_t)((dp->l.state & TS_V) != 0)) { trace_reload_section(dp, 1); }
		trace_core(dp, &t);
		if(getenv("OG_DEBUG_TRACE")) { fprintf(stderr, "tc ppos %lu q %u state %u gidx %d %d\n", (unsigned long)t.ppos, dp->l.q, dp->l.state, dp->l.gidx[0], dp->l.gidx[1]); }
		if(dp->l.q >= (uint32_t)t.W) {                                /* out of band: abort, gaba.c:3324 */
			free(aln->path - 2); free(segbuf); free(aln);
			return NULL;
		}
		/* trace_push_segment */
		if(slen >= segcap) { segcap *= 2; segbuf = (og_segment_t *)realloc(segbuf, segcap * sizeof(og_segment_t)); }
		og_segment_t *s = &segbuf[slen++];
		s->aid = dp->l.id[0]; s->bid = dp->l.id[1];
		s->apos = dp->l.ofs[0] + (uint32_t)dp->l.gidx[0]; s->bpos = dp->l.ofs[1] + (uint32_t)dp->l.gidx[1];
		s->alen = (uint32_t)(dp->l.sgidx[0] - dp->l.gidx[0]); s->blen = (uint32_t)(dp->l.sgidx[1] - dp->l.gidx[1]);
		s->ppos = t.ppos;
		dp->l.sgidx[0] = dp->l.gidx[0]; dp->l.sgidx[1] = dp->l.gidx[1];
	}
	/* segments are pushed backward in the reference (seg--): the last pushed one is seg[0] */
	aln->seg = (og_segment_t *)calloc(slen + 1, sizeof(og_segment_t));
	for(uint64_t i = 0; i < slen; i++) { aln->seg[i] = segbuf[slen - 1 - i]; }
	free(segbuf);
	aln->slen = (uint32_t)slen;

	/* identity estimate, gaba.c:3334-3355 */
	og_ctx_t const *c = dp->ctx;
	int32_t gcnt[2], g[2];
	for(int k = 0; k < 2; k++) { gcnt[k] = (int32_t)(dp->l.ecnt[k] + dp->l.fcnt[k]); }
	/* _mul_v2i32 is _mm_mul_epi32 (v2i32.h:106): a 32x32->64 multiply of lane 0 only, whose upper half
	 * lands in lane 1 -- i.e. only the a-side gap penalties enter the identity estimate. Reproduced as is. */
	int64_t p1 = (int64_t)c->gi * (int64_t)(int32_t)dp->l.icnt[0];
	int64_t p2 = (int64_t)c->ge * (int64_t)(int32_t)dp->l.ecnt[0];
	int64_t p3 = (int64_t)c->gfa * (int64_t)(int32_t)dp->l.fcnt[0];
	g[0] = (int32_t)((uint32_t)p1 + (uint32_t)p2 + (uint32_t)p3);
	g[1] = (int32_t)((uint32_t)(p1 >> 32) + (uint32_t)(p2 >> 32) + (uint32_t)(p3 >> 32));
	uint64_t dlen = (plen - (uint64_t)(int64_t)gcnt[1] - (uint64_t)(int64_t)gcnt[0]) >> 1;
	int64_t dsc = tail->f.max + g[1] + g[0];
	aln->score = tail->f.max;
	aln->identity = dlen == 0 ? 0.0 : (((double)dsc / (double)dlen) * c->imx - c->xmx);
	aln->agcnt = (uint32_t)gcnt[0]; aln->bgcnt = (uint32_t)gcnt[1];
	aln->dcnt = (uint32_t)dlen;
	aln->plen = (uint32_t)plen;
	return aln;
}
void og_aln_free(og_alignment_t *aln) { if(aln) { free(aln->path - 2); free(aln->seg); free(aln); } }

/* ---- CIGAR, gaba_parse.h:107-263 ---- */
static inline uint64_t parse_u64(uint64_t const *ptr, int64_t pos)    /* gaba_parse_u64, gaba_parse.h:110 */
{
	int64_t rem = pos & 63;
	return (ptr[pos >> 6] >> rem) | ((ptr[(pos >> 6) + 1] << (63 - rem)) << 1);
}
static uint64_t dump_num(char *buf, uint64_t len, char ch)            /* gaba_parse_dump_num, gaba_parse.h:194 */
{
	char tmp[24]; int n = 0;
	if(len == 0) { tmp[n++] = '0'; }
	while(len != 0) { tmp[n++] = (char)('0' + len % 10); len /= 10; }
	uint64_t adv = 0;
	while(n > 0) { buf[adv++] = tmp[--n]; }
	buf[adv++] = ch;
	return adv;
}
/* the reference casts (uint32_t *)path to a 64-bit aligned base + bit offset (gaba_parse.h:80-90) */
#define PARSE_PTR(_p)   ( (uint64_t const *)((uint64_t)(_p) & ~(uint64_t)(sizeof(uint64_t) - 1)) )
#define PARSE_OFS(_p)   ( ((uint64_t)(_p) & sizeof(uint32_t)) ? 32 : 0 )
uint64_t og_dump_cigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	(void)buf_size;
	char *b = buf;
	uint64_t const *p = PARSE_PTR(path);
	uint64_t ofs = (uint64_t)((int64_t)offset + PARSE_OFS(path) - 64), idx = len;
	while((int64_t)idx > 0) {                                        /* _parser_loop_rv, gaba_parse.h:168-188 */
		uint64_t m, c;
		m = lz64(parse_u64(p, (int64_t)(ofs + idx)));
		c = MIN2(idx, m - (m > 0));
		idx -= c; if(c) { b += dump_num(b, c, 'D'); }
		m = lz64(~parse_u64(p, (int64_t)(ofs + idx)));
		c = MIN2(idx, m);
		idx -= c; if(c) { b += dump_num(b, c, 'I'); }
		uint64_t sidx = idx;
		do {
			m = lz64(parse_u64(p, (int64_t)(ofs + idx)) ^ 0x5555555555555555ULL);
			c = MIN2(idx, m) & ~0x01ULL;
			idx -= c;
		} while(c == 64);
		if((sidx - idx) >> 1) { b += dump_num(b, (sidx - idx) >> 1, 'M'); }
	}
	*b = '\0';
	return (uint64_t)(b - buf);
}
void og_parse_path_reverse(uint32_t const *path, uint64_t offset, uint64_t len, void (*fn)(void *ctx, char op, uint64_t cnt), void *ctx)
{
	uint64_t const *p = PARSE_PTR(path);
	uint64_t ofs = (uint64_t)((int64_t)offset + PARSE_OFS(path) - 64), idx = len;
	while((int64_t)idx > 0) {
		uint64_t m, c;
		m = lz64(parse_u64(p, (int64_t)(ofs + idx)));
		c = MIN2(idx, m - (m > 0));
		idx -= c; if(c) { fn(ctx, 'D', c); }
		m = lz64(~parse_u64(p, (int64_t)(ofs + idx)));
		c = MIN2(idx, m);
		idx -= c; if(c) { fn(ctx, 'I', c); }
		uint64_t sidx = idx;
		do {
			m = lz64(parse_u64(p, (int64_t)(ofs + idx)) ^ 0x5555555555555555ULL);
			c = MIN2(idx, m) & ~0x01ULL;
			idx -= c;
		} while(c == 64);
		if((sidx - idx) >> 1) { fn(ctx, 'M', (sidx - idx) >> 1); }
	}
}
typedef struct { char *r; uint8_t const *q; uint32_t conf; char gap; } dump_seq_t;
static void dump_seq_step(void *ctx, char op, uint64_t c)
{
	dump_seq_t *d = (dump_seq_t *)ctx;
	int const is_b = (d->conf & OG_SEQ_B) != 0;
	if((op == 'D' && is_b) || (op == 'I' && !is_b)) { memset(d->r, d->gap, c); d->r += c; return; }
	if(d->conf & OG_SEQ_RV) { for(uint64_t t = 0; t < c; t++) { *d->r++ = "TGCANNNNNNNNNNNN"[*--d->q & 15]; } }
	else { for(uint64_t t = 0; t < c; t++) { *d->r++ = "ACGTNNNNNNNNNNNN"[*d->q++ & 15]; } }
}
uint64_t og_dump_seq_reverse(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap)
{
	(void)buf_size;
	dump_seq_t d = { buf, seq, conf, gap };
	og_parse_path_reverse(path, offset, len, dump_seq_step, &d);
	*d.r = 0;
	return (uint64_t)(d.r - buf);
}
uint64_t og_dump_cigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	(void)buf_size;
	char *b = buf;
	uint64_t const *p = PARSE_PTR(path);
	uint64_t lim = offset + PARSE_OFS(path) + len, ridx = len;
	while((int64_t)ridx > 0) {                                       /* _parser_loop_fw, gaba_parse.h:147-167 */
		uint64_t m, c;
		m = tz64(~parse_u64(p, (int64_t)(lim - ridx)));
		c = MIN2(ridx, m - (m > 0));
		ridx -= c; if(c) { b += dump_num(b, c, 'I'); }
		m = tz64(parse_u64(p, (int64_t)(lim - ridx)));
		c = MIN2(ridx, m);
		ridx -= c; if(c) { b += dump_num(b, c, 'D'); }
		uint64_t sridx = ridx;
		do {
			m = tz64(parse_u64(p, (int64_t)(lim - ridx)) ^ 0x5555555555555555ULL);
			c = MIN2(ridx, m) & ~0x01ULL;
			ridx -= c;
		} while(c == 64);
		if((sridx - ridx) >> 1) { b += dump_num(b, (sridx - ridx) >> 1, 'M'); }
	}
	*b = '\0';
	return (uint64_t)(b - buf);
}

/* ---- test convenience: same call pattern as mm_extend_core (minialign.c:4075-4112) ---- */
int og_extend(og_dp_t *dp, int bw_idx,
	uint8_t const *a, uint32_t alen, uint32_t apos, int arev,
	uint8_t const *b, uint32_t blen, uint32_t bpos, int brev,
	int do_trace, og_xresult_t *res, uint32_t *path_out)
{
	static uint8_t tailseq[128];
	memset(tailseq, 4, 128);
	og_dp_flush(dp);
	og_section_t as = { arev ? 1u : 0u, alen, arev ? og_mirror(a, alen) : a };
	og_section_t bs = { brev ? 3u : 2u, blen, brev ? og_mirror(b, blen) : b };
	og_section_t ts = { 0xfffffffeu, 96, tailseq };
	og_section_t const *ap = &as, *bp = &bs;
	memset(res, 0, sizeof(*res));
	og_fill_t const *f = og_dp_fill_root(dp, bw_idx, ap, apos, bp, bpos, 0);
	og_fill_t const *m = f;
	#define REC(_f) { og_xfill_t *s = &res->fill[res->n_fill < 8 ? res->n_fill : 7]; \
		s->max = (_f)->max; s->status = (_f)->status; s->aid = (_f)->aid; s->bid = (_f)->bid; \
		s->ascnt = (_f)->ascnt; s->bscnt = (_f)->bscnt; s->apos = (_f)->apos; s->bpos = (_f)->bpos; res->n_fill++; }
	REC(f);
	uint32_t flag = OG_TERM;
	while((flag & f->status) == 0) {
		if(f->status & OG_UPDATE_A) { ap = &ts; }
		if(f->status & OG_UPDATE_B) { bp = &ts; }
		flag |= f->status & (OG_UPDATE_A | OG_UPDATE_B);
		f = og_dp_fill(dp, f, ap, bp, 0);
		REC(f);
		if(f->max > m->max) { m = f; res->max_fill_idx = res->n_fill - 1; }
	}
	og_pos_pair_t const *pp = og_dp_search_max(dp, m);
	res->p_aid = pp->aid; res->p_bid = pp->bid; res->p_apos = pp->apos; res->p_bpos = pp->bpos; res->p_plen = pp->plen;
	if(do_trace) {
		og_alignment_t *aln = og_dp_trace(dp, m);
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
		og_aln_free(aln);
	}
	return 0;
}
