/*
 * gaba_batch.hip -- batched extension kernel + the host side of include/gaba.h.
 *
 * Kernel: persistent wavefronts (4 per 256-thread workgroup, no inter-wave communication) pull jobs from
 * an atomic counter; each wave owns a slab of HBM for its blocks and tails (the role of libgaba's per-thread
 * bump stack, gaba.c:3895-3969).  Host: score validation and root-block construction follow
 * gaba_init (gaba.c:3614-3830); sequences are packed 2-bit + N-mask before upload.
 */
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "gaba_device.hpp"
#include "../../include/gaba.h"

using namespace gaba;

#include "gaba_host.hpp"

struct DevJob { Sec a, b; uint32_t apos, bpos, bw_idx, do_trace; };

/* ---- kernel ---- */
#ifndef GABA_WAVES_PER_SIMD
#define GABA_WAVES_PER_SIMD 8
#endif
__global__ void __launch_bounds__(256, GABA_WAVES_PER_SIMD)
gaba_extend_batch_kernel(const Consts c, const uint8_t *roots, SeqArena ar_a, SeqArena ar_b,
	const DevJob *jobs, uint32_t njobs, gaba_xresult_t *res, uint32_t *paths, uint32_t path_stride,
	uint8_t *slabs, uint64_t slab_bytes, uint32_t *counter, uint64_t *stats, int *errs)
{
	SeqArena ar[2] = { ar_a, ar_b };
	Ctx x;
	x.c = c; x.ar = ar; x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
	uint32_t wave = (uint32_t)rdfirst((int)(blockIdx.x * 4 + threadIdx.x / 64));
	x.slab = slabs + (uint64_t)wave * slab_bytes;
	x.cap = (uint32_t)slab_bytes; x.top = SLAB_HEAD;
	/* private copy of the root blocks at the head of the slab */
	for(uint32_t i = (uint32_t)x.lane; i < SLAB_HEAD / 4; i += 64) { ((uint32_t *)x.slab)[i] = ((const uint32_t *)roots)[i]; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

	const Sec tailsec = { 0xfffffffeu, 96, 0, 2, 0 };
	while(true) {
		uint32_t j = 0;
		if(x.lane == 0) { j = atomicAdd(counter, 1u); }
		j = (uint32_t)rdfirst((int)j);
		if(j >= njobs) { break; }
		DevJob job = jobs[j];
		Sec sa = job.a, sb = job.b;
		sa.id = (uint32_t)rdfirst((int)sa.id); sa.len = (uint32_t)rdfirst((int)sa.len); sa.off = rdfirst64(sa.off); sa.arena = 0; sa.rev = (uint32_t)rdfirst((int)sa.rev);
		sb.id = (uint32_t)rdfirst((int)sb.id); sb.len = (uint32_t)rdfirst((int)sb.len); sb.off = rdfirst64(sb.off); sb.arena = 1; sb.rev = (uint32_t)rdfirst((int)sb.rev);
		uint32_t apos = (uint32_t)rdfirst((int)job.apos), bpos = (uint32_t)rdfirst((int)job.bpos);
		int bw = rdfirst((int)job.bw_idx); bool do_trace = rdfirst((int)job.do_trace) != 0;
		gaba_xresult_t *r = &res[j];
		x.err = 0; x.no_trace = !do_trace;          /* no traceback requested: the masks are never read */
		dp_flush(x);

		/* mm_extend_core call pattern (minialign.c:4075-4112), recording every fill */
		Sec ca = sa, cb = sb;
		uint32_t f = NIL, m = NIL, nfill = 0, maxidx = 0;
		int64_t mmax = 0;
		uint32_t flag = STATUS_TERM;
		while(true) {
			f = dp_fill_any(x, f, bw, ca, apos, cb, bpos, 0);
			const Tail *t = tail_at(x, f);
			uint32_t st = (uint32_t)rdfirst((int)t->f.status);
			int64_t fmax = (int64_t)rdfirst64((uint64_t)t->f.max);
			if(x.lane == 0) {
				gaba_xfill_t *o = &r->fill[nfill < 8 ? nfill : 7];
				o->max = t->f.max; o->status = t->f.status; o->aid = t->f.aid; o->bid = t->f.bid;
				o->ascnt = t->f.ascnt; o->bscnt = t->f.bscnt; o->apos = t->f.apos; o->bpos = t->f.bpos;
			}
			if(nfill == 0 || fmax > mmax) { m = f; mmax = fmax; maxidx = nfill; }
			nfill++;
			if((flag & st) != 0 || x.err) { break; }
			if(st & UPDATE_A) { ca = tailsec; }
			if(st & UPDATE_B) { cb = tailsec; }
			flag |= st & (UPDATE_A | UPDATE_B);
		}
		Leaf lf;
		PosPair pp = dp_search_max(x, m, lf);
		AlnOut ao; ao.status = 0;
		if(do_trace && !x.err) {
			ao = dp_trace(x, m, paths + (uint64_t)j * path_stride, path_stride, (Segment *)r->seg, 16);
		}
		if(x.lane == 0) {
			r->n_fill = nfill; r->max_fill_idx = maxidx;
			r->p_aid = pp.aid; r->p_bid = pp.bid; r->p_apos = pp.apos; r->p_bpos = pp.bpos; r->p_plen = pp.plen;
			r->traced = do_trace ? ao.status : 0;
			if(do_trace && ao.status == 1) {
				r->score = ao.score; r->identity = ao.identity; r->agcnt = ao.agcnt; r->bgcnt = ao.bgcnt;
				r->dcnt = ao.dcnt; r->slen = ao.slen; r->plen = ao.plen; r->n_path_words = (ao.plen + 31) / 32;
			}
			errs[j] = x.err;
		}
	}
	if(x.lane == 0) {
		atomicAdd((unsigned long long *)&stats[0], (unsigned long long)x.n_vec);
		atomicAdd((unsigned long long *)&stats[1], (unsigned long long)x.n_blk);
		atomicAdd((unsigned long long *)&stats[2], (unsigned long long)x.n_tr);
	}
}

/* ---- host: scoring constants and root blocks (gaba.c:3614-3830) ---- */
namespace {
struct HP { int8_t sm[16]; int gi, ge, gfa, gfb, xdrop; int model; };
int maxm(const HP &p) { int m = -128; for(int i = 0; i < 16; i++) m = p.sm[i] > m ? p.sm[i] : m; return m; }
int minm(const HP &p) { int m = 127; for(int i = 0; i < 16; i++) m = p.sm[i] < m ? p.sm[i] : m; return m; }
int gap_aff(const HP &p, int l) { return -1 * (l > 0) * p.gi - p.ge * l; }
int gap_h(const HP &p, int l) { int a = gap_aff(p, l); int f = -p.gfb * l; return p.model == MODEL_COMBINED ? (a > f ? a : f) : a; }   /* gaba.c:818-838 */
int gap_v(const HP &p, int l) { int a = gap_aff(p, l); int f = -p.gfa * l; return p.model == MODEL_COMBINED ? (a > f ? a : f) : a; }

bool scores_ok(const HP &p)            /* gaba_init_check_score, gaba.c:3614-3640, run at W = 16 (gaba_wrap.h:286) */
{
	int mm = maxm(p), mn = minm(p), ofs = p.gi + p.ge;
	if(mm <= 0 || mm > 6 || mn >= 0 || mn < -7) return false;
	if(mn < -2 * ofs) return false;
	if(p.gfa != 0 && p.gfb != 0 && mn <= -(p.gfa + p.gfb)) return false;
	if(p.ge <= 0 || p.gi < 0) return false;
	if(p.gfa < 0 || (p.gfa != 0 && p.gfa <= p.ge)) return false;
	if(p.gfb < 0 || (p.gfb != 0 && p.gfb <= p.ge)) return false;
	if((p.gfa == 0) != (p.gfb == 0)) return false;
	for(int i = 0; i < 8; i++) {
		int t1 = ofs + gap_h(p, 2*i + 1) - gap_h(p, 2*i);
		int t2 = ofs + (mm + gap_v(p, 2*i + 1)) - gap_v(p, 2*(i + 1));
		int t3 = ofs + (mm + gap_h(p, 2*i + 1)) - gap_h(p, 2*(i + 1));
		int t4 = t1;
		int mx = t1; if(t2 > mx) mx = t2; if(t3 > mx) mx = t3; if(t4 > mx) mx = t4;
		int mi = t2; if(t3 < mi) mi = t3; if(t4 < mi) mi = t4;
		if(mx > 127 || mi < 0) return false;
	}
	return true;
}

void build_root(const HP &p, int W, Blk *b, Tail *t, uint32_t blk_off)   /* gaba_init_phantom, gaba.c:3739-3800 */
{
	memset(b, 0, sizeof(Blk)); memset(t, 0, sizeof(Tail));
	int mm = maxm(p), ofs = p.gi + p.ge;
	int8_t dh[64] = {0}, dv[64] = {0}, de[64] = {0}, df[64] = {0};
	for(int i = 0; i < W / 2; i++) {                                   /* gaba_init_diff_vectors, gaba.c:3705-3733 */
		int lo = W/2 - 1 - i, hi = W/2 + i;
		dh[lo] = (int8_t)(ofs + gap_h(p, 2*i + 1) - gap_h(p, 2*i));
		dh[hi] = (int8_t)(ofs + mm + gap_v(p, 2*i + 1) - gap_v(p, 2*(i + 1)));
		dv[lo] = (int8_t)(ofs + mm + gap_h(p, 2*i + 1) - gap_h(p, 2*(i + 1)));
		dv[hi] = (int8_t)(ofs + gap_v(p, 2*i + 1) - gap_v(p, 2*i));
		de[lo] = (int8_t)(p.gi + dv[lo] + gap_aff(p, 2*i + 1) - gap_h(p, 2*i + 1));
		de[hi] = (int8_t)(p.gi + dv[hi] - p.gi);
		df[lo] = (int8_t)(p.gi + dh[lo] - p.gi);
		df[hi] = (int8_t)(p.gi + dh[hi] + gap_aff(p, 2*i + 1) - gap_v(p, 2*i + 1));
	}
	for(int l = 0; l < 64; l++) {
		int8_t ndh = (int8_t)(0 - dh[l]);
		b->diff[l] = (uint32_t)(uint8_t)ndh | ((uint32_t)(uint8_t)dv[l] << 8) | ((uint32_t)(uint8_t)de[l] << 16) | ((uint32_t)(uint8_t)df[l] << 24);
	}
	b->s.acc = 0; b->s.xstat = ROOT; b->s.acnt = 0; b->s.bcnt = 0; b->s.dir_mask = 0; b->s.max_mask = 0; b->s.link = NIL;
	int64_t init_max = -(mm + gap_h(p, 1));
	t->f.max = init_max; t->f.status = CONT | UPDATE_A | UPDATE_B;
	t->f.apos = (uint64_t)(int64_t)(-W / 2); t->f.bpos = (uint64_t)(int64_t)(-W / 2);
	t->tail = NIL; t->last = blk_off; t->W = W;
	t->mdrop = (int16_t)(init_max - 128);
	t->ch[0] = 0x0c; t->ch[W - 1] = 0x03 << 4;
	for(int l = 0; l < 64; l++) t->xd[l] = -128;
	for(int i = 0; i < W / 2; i++) {                                   /* gaba_init_middle_delta, gaba.c:3684-3697 */
		t->md[W/2 - 1 - i] = (int16_t)(-(i + 1) * mm + gap_h(p, 2*i + 1));
		t->md[W/2 + i]     = (int16_t)(-(i + 1) * mm + gap_v(p, 2*i + 1));
	}
}
} /* anonymous */

extern "C" {

gaba_t *gaba_init(gaba_params_t const *params)
{
	if(params == NULL) return NULL;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
		fprintf(stderr, "[minialign_amd] gaba_init: no HIP device available (this library has no CPU path)\n");
		return NULL;
	}
	HP p; memcpy(p.sm, params->score_matrix, 16);
	p.gi = params->gi; p.ge = params->ge; p.gfa = params->gfa; p.gfb = params->gfb;
	p.xdrop = params->xdrop == 0 ? 50 : params->xdrop;                  /* gaba_init_restore_default, gaba.c:3605 */
	/* gaba_wrap.h:213-221.  gi == 0 selects the reference's linear-gap build (the `ava' preset); its fills, max positions, paths, segments and counts
	 * are those of the affine recurrences run with gi = 0 and gf ignored (checked against the compiled reference on random jobs over five score sets,
	 * tests/golden/gaba_extend.json group `linear'), so the AFFINE kernels run */
	p.model = p.gi != 0 ? ((p.gfa != 0 && p.gfb != 0) ? MODEL_COMBINED : MODEL_AFFINE) : MODEL_AFFINE;
	if(p.gi == 0) { p.gfa = p.gfb = 0; }
	if(!scores_ok(p)) return NULL;

	gaba_t *ctx = (gaba_t *)calloc(1, sizeof(gaba_t));
	Consts &c = ctx->hc;
	c.model = p.model;
	for(int i = 0; i < 16; i++) { ((int8_t *)c.sb)[i] = (int8_t)(p.sm[i] + 2 * (p.ge + p.gi)); }   /* gaba.c:3657 */
	c.adjh = c.adjv = p.gi; c.ofsh = c.ofsv = -(p.gi + p.ge);
	c.gfh = (p.gi + p.ge) - p.gfb; c.gfv = (p.gi + p.ge) - p.gfa;        /* gaba.c:3671-3675 */
	c.tx = (int8_t)(p.xdrop - 128);
	c.gi = p.gi; c.ge = p.ge; c.gfa = p.gfa; c.gfb = p.gfb;
	int64_t acc[2] = { 0, 0 };
	for(int i = 0; i < 16; i++) acc[(i & 3) == (i >> 2)] += p.sm[i];
	double m = (double)acc[1] / 4.0, xx = (double)acc[0] / 12.0;           /* gaba.c:3815-3827 */
	c.imx = 1 / (m - xx); c.xmx = xx / (m - xx);
	/* single-v_perm lookup (gaba_device.hpp:step): rows of the table per a code, one score for a b side N */
	{
		const int8_t *sb = (const int8_t *)c.sb;
		c.fast_score = (sb[2] == sb[3] && sb[2] == sb[6]) ? 1 : 2;          /* (2: the entries a b side N reads -- sb[a | 2] -- differ: blocks with such an N take the general lookup, fill_block) */
		c.score_n = sb[2];
		for(int a = 0; a < 5; a++) {
			uint32_t row = 0;
			for(int j = 0; j < 4; j++) { row |= (uint32_t)(uint8_t)sb[(a | (4 * j)) & 15] << (8 * j); }
			c.arow[a] = row;
		}
	}

	std::vector<uint8_t> roots(SLAB_HEAD);
	int Ws[3] = { 64, 32, 16 };
	for(int i = 0; i < 3; i++) {
		build_root(p, Ws[i], (Blk *)&roots[i * ROOT_STRIDE], (Tail *)&roots[i * ROOT_STRIDE + sizeof(Blk)], i * ROOT_STRIDE);
	}
	HIP_OK(hipMalloc(&ctx->dc, sizeof(Consts)), NULL);
	HIP_OK(hipMemcpy(ctx->dc, &c, sizeof(Consts), hipMemcpyHostToDevice), NULL);
	HIP_OK(hipMalloc(&ctx->droots, SLAB_HEAD), NULL);
	HIP_OK(hipMemcpy(ctx->droots, roots.data(), SLAB_HEAD, hipMemcpyHostToDevice), NULL);
	HIP_OK(hipMalloc(&ctx->counter, sizeof(uint32_t)), NULL);
	HIP_OK(hipMalloc(&ctx->dstats, 4 * sizeof(uint64_t)), NULL);
	HIP_OK(hipStreamCreate(&ctx->stream), NULL);
	HIP_OK(hipEventCreate(&ctx->ev0), NULL); HIP_OK(hipEventCreate(&ctx->ev1), NULL);
	return ctx;
}

void gaba_clean(gaba_t *ctx)
{
	if(!ctx) return;
	hipFree(ctx->dc); hipFree(ctx->droots); hipFree(ctx->counter); hipFree(ctx->dstats);
	if(ctx->slabs) hipFree(ctx->slabs);
	hipEventDestroy(ctx->ev0); hipEventDestroy(ctx->ev1); hipStreamDestroy(ctx->stream);
	free(ctx);
}

/* arenas by the host range they were uploaded from: a gaba_section_t of the per-call API points into one of them */
static std::vector<gaba_arena_t *> g_arenas;
static void arena_registry_add(gaba_arena_t *ar) { g_arenas.push_back(ar); }
static void arena_registry_del(gaba_arena_t *ar) { for(size_t i = 0; i < g_arenas.size(); i++) if(g_arenas[i] == ar) { g_arenas.erase(g_arenas.begin() + i); break; } }

gaba_arena_t *gaba_arena_upload(uint8_t const *bases, uint64_t n)
{
	uint64_t nw = (n + 15) / 16 + 4, nn = (n + 31) / 32 + 4;
	std::vector<uint32_t> pk(nw, 0), nm(nn, 0);
	/* stretches that start on a multiple of 32 bases own whole words of both arrays: packed on host threads for long inputs (a genome) */
	auto pack = [&](uint64_t lo, uint64_t hi) {
		for(uint64_t i = lo; i < hi; i++) {
			uint32_t c = bases[i];
			if(c > 3) { nm[i >> 5] |= 1u << (i & 31); c = 0; }
			pk[i >> 4] |= c << (2 * (i & 15));
		}
	};
	const uint32_t nth = n < (64u << 20) ? 1u : std::min<uint32_t>(std::max<uint32_t>(1, std::thread::hardware_concurrency()), 32);
	if(nth == 1) pack(0, n);
	else {
		std::vector<std::thread> th; const uint64_t step = ((n / nth) + 31) & ~31ull;
		for(uint32_t t = 0; t < nth; t++) { const uint64_t lo = std::min<uint64_t>(n, t * step), hi = t + 1 == nth ? n : std::min<uint64_t>(n, (t + 1) * step); if(lo < hi) th.emplace_back(pack, lo, hi); }
		for(auto &x : th) x.join();
	}
	gaba_arena_t *ar = (gaba_arena_t *)calloc(1, sizeof(gaba_arena_t));
	ar->n = n; ar->host = bases;
	HIP_OK(hipMalloc(&ar->pk, nw * 4), NULL); HIP_OK(hipMalloc(&ar->nm, nn * 4), NULL);
	HIP_OK(hipMemcpy(ar->pk, pk.data(), nw * 4, hipMemcpyHostToDevice), NULL);
	HIP_OK(hipMemcpy(ar->nm, nm.data(), nn * 4, hipMemcpyHostToDevice), NULL);
	arena_registry_add(ar);
	return ar;
}
void gaba_arena_unregister(gaba_arena_t *ar) { if(ar) { arena_registry_del(ar); ar->host = NULL; } }
void gaba_arena_free(gaba_arena_t *ar) { if(ar) { arena_registry_del(ar); hipFree(ar->pk); hipFree(ar->nm); free(ar); } }

int gaba_dp_extend_batch(gaba_t *ctx, gaba_arena_t const *a, gaba_arena_t const *b,
	gaba_job_t const *jobs, uint32_t n, gaba_xresult_t *results, uint32_t *paths, uint32_t path_stride)
{
	if(!ctx || !a || !b || n == 0) return -1;
	/* workspace: every vector costs 40.5 B (1296 B per 32-vector block); a job needs at most about
	 * 2 x (alen' + blen') / 32 blocks for its DOWN-style fill chain */
	uint64_t need = 0;
	std::vector<DevJob> dj(n);
	for(uint32_t i = 0; i < n; i++) {
		const gaba_job_t &j = jobs[i];
		if(j.apos >= j.alen || j.bpos >= j.blen || j.bw_idx > 2) return -1;
		dj[i].a = Sec{ j.arev ? 1u : 0u, j.alen, j.a_off, 0, j.arev ? 1u : 0u };
		dj[i].b = Sec{ j.brev ? 3u : 2u, j.blen, j.b_off, 1, j.brev ? 1u : 0u };
		dj[i].apos = j.apos; dj[i].bpos = j.bpos; dj[i].bw_idx = j.bw_idx; dj[i].do_trace = j.do_trace;
		uint64_t ra = j.alen - j.apos, rb = j.blen - j.bpos, mn = ra < rb ? ra : rb;
		uint64_t blocks = (2 * mn + 8192) / 32 + 64;
		if(blocks * sizeof(Blk) > need) need = blocks * sizeof(Blk);
	}
	need += SLAB_HEAD + 16 * sizeof(Tail) + 4096;
	need = (need + 255) & ~255ull;
	int dev = 0; hipDeviceProp_t prop;
	HIP_OK(hipGetDevice(&dev), -1); HIP_OK(hipGetDeviceProperties(&prop, dev), -1);
	uint32_t max_waves = (uint32_t)prop.multiProcessorCount * 4u * GABA_WAVES_PER_SIMD;          /* default: as many 4-wave workgroups per CU as the launch bound allows */
	uint32_t waves = n < max_waves ? ((n + 3) & ~3u) : max_waves;
	if(ctx->slab_bytes < need || ctx->n_waves < waves) {
		if(ctx->slabs) hipFree(ctx->slabs);
		ctx->slab_bytes = need; ctx->n_waves = waves;
		HIP_OK(hipMalloc(&ctx->slabs, (uint64_t)waves * need), -1);
	}
	DevJob *djobs; gaba_xresult_t *dres; uint32_t *dpaths; int *derr;
	HIP_OK(hipMalloc(&djobs, n * sizeof(DevJob)), -1);
	HIP_OK(hipMalloc(&dres, n * sizeof(gaba_xresult_t)), -1);
	HIP_OK(hipMalloc(&dpaths, (uint64_t)n * path_stride * 4 + 64), -1);
	HIP_OK(hipMalloc(&derr, n * sizeof(int)), -1);
	HIP_OK(hipMemcpyAsync(djobs, dj.data(), n * sizeof(DevJob), hipMemcpyHostToDevice, ctx->stream), -1);
	HIP_OK(hipMemsetAsync(dres, 0, n * sizeof(gaba_xresult_t), ctx->stream), -1);
	HIP_OK(hipMemsetAsync(ctx->counter, 0, 4, ctx->stream), -1);
	HIP_OK(hipMemsetAsync(ctx->dstats, 0, 32, ctx->stream), -1);
	SeqArena sa{ a->pk, a->nm }, sb{ b->pk, b->nm };
	HIP_OK(hipEventRecord(ctx->ev0, ctx->stream), -1);
	hipLaunchKernelGGL(gaba_extend_batch_kernel, dim3(waves / 4), dim3(256), 0, ctx->stream,
		ctx->hc, ctx->droots, sa, sb, djobs, n, dres, dpaths, path_stride, ctx->slabs, ctx->slab_bytes, ctx->counter, ctx->dstats, derr);
	HIP_OK(hipGetLastError(), -1);
	HIP_OK(hipEventRecord(ctx->ev1, ctx->stream), -1);
	std::vector<int> herr(n);
	uint64_t hstats[4];
	HIP_OK(hipMemcpyAsync(results, dres, n * sizeof(gaba_xresult_t), hipMemcpyDeviceToHost, ctx->stream), -1);
	if(paths) HIP_OK(hipMemcpyAsync(paths, dpaths, (uint64_t)n * path_stride * 4, hipMemcpyDeviceToHost, ctx->stream), -1);
	HIP_OK(hipMemcpyAsync(herr.data(), derr, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream), -1);
	HIP_OK(hipMemcpyAsync(hstats, ctx->dstats, 32, hipMemcpyDeviceToHost, ctx->stream), -1);
#ifdef GABA_TRACE_PROF
	{
		unsigned long long hp[16]; HIP_OK(hipStreamSynchronize(ctx->stream), -1);
		HIP_OK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(gaba::g_trace_prof), sizeof(hp)), -1);
		const char *nm[4] = { "reload", "diag-batch", "gap-batch", "stepwise" };
		for(int i = 0; i < 4; i++) fprintf(stderr, "[trace prof] %-10s ticks %llu  events %llu  ticks/event %.1f\n", nm[i], hp[i], hp[8 + i], hp[8 + i] ? (double)hp[i] / hp[8 + i] : 0.0);
		unsigned long long z[16] = { 0 }; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(gaba::g_trace_prof), z, sizeof(z)), -1);
	}
#endif
	HIP_OK(hipStreamSynchronize(ctx->stream), -1);
	float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
	ctx->last.kernel_ms = ms; ctx->last.vectors = hstats[0]; ctx->last.blocks = hstats[1]; ctx->last.trace_steps = hstats[2];
	hipFree(djobs); hipFree(dres); hipFree(dpaths); hipFree(derr);
	int rc = 0;
	for(uint32_t i = 0; i < n; i++) {
		if(herr[i] == 1) rc = -2; else if(herr[i] == 2 && rc == 0) rc = -3;
		/* mask the bits past plen in the last path word, as the tests compare whole words */
		if(paths && results[i].traced == 1 && (results[i].plen & 31)) {
			paths[(uint64_t)i * path_stride + results[i].n_path_words - 1] &= (1u << (results[i].plen & 31)) - 1;
		}
	}
	return rc;
}

void gaba_last_stats(gaba_t *ctx, gaba_batch_stats_t *out) { if(ctx && out) *out = ctx->last; }

/* =====================================================================================================
 * The per-call API of gaba.h:266-357 on a persistent device workspace: one single-wave launch per call.
 * A gaba_dp_t owns one slab in HBM (root blocks at its head, bump pointer kept on the host); the gaba_fill_t
 * returned to the caller carries the slab offset of its tail in reserved[0].
 * ===================================================================================================== */
struct ScalarOut { uint32_t tail, top; int32_t err; uint32_t _pad; Fill f; PosPair pp; AlnOut ao; };

__global__ void __launch_bounds__(64) gaba_scalar_fill_kernel(const Consts c, SeqArena ar_a, SeqArena ar_b, uint8_t *slab, uint32_t top, uint32_t cap,
	uint32_t prev_tail, int bw_idx, Sec sa, uint32_t apos, Sec sb, uint32_t bpos, uint32_t pridx, ScalarOut *out)
{
	SeqArena ar[2] = { ar_a, ar_b };
	Ctx x; x.c = c; x.ar = ar; x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
	x.slab = slab; x.cap = cap; x.top = top;
	const uint32_t f = dp_fill_any(x, prev_tail, bw_idx, sa, apos, sb, bpos, pridx);
	if(x.lane == 0) { out->tail = f; out->top = x.top; out->err = x.err; out->f = tail_at(x, f)->f; }
}
__global__ void __launch_bounds__(64) gaba_scalar_search_kernel(const Consts c, SeqArena ar_a, SeqArena ar_b, uint8_t *slab, uint32_t top, uint32_t cap,
	uint32_t tail, ScalarOut *out)
{
	SeqArena ar[2] = { ar_a, ar_b };
	Ctx x; x.c = c; x.ar = ar; x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
	x.slab = slab; x.cap = cap; x.top = top;
	Leaf lf;
	const PosPair pp = dp_search_max(x, tail, lf);
	if(x.lane == 0) { out->pp = pp; out->top = x.top; out->err = x.err; }
}
__global__ void __launch_bounds__(64) gaba_scalar_trace_kernel(const Consts c, SeqArena ar_a, SeqArena ar_b, uint8_t *slab, uint32_t top, uint32_t cap,
	uint32_t tail, uint32_t *path, uint64_t path_words, Segment *seg, uint32_t max_seg, ScalarOut *out)
{
	SeqArena ar[2] = { ar_a, ar_b };
	Ctx x; x.c = c; x.ar = ar; x.lane = lane_id(); x.err = 0; x.no_trace = false; x.n_vec = x.n_blk = x.n_tr = 0;
	x.slab = slab; x.cap = cap; x.top = top;
	const AlnOut ao = dp_trace(x, tail, path, path_words, seg, max_seg);
	if(x.lane == 0) { out->ao = ao; out->top = x.top; out->err = x.err; }
}

struct gaba_dp_context_s {
	gaba_t const *ctx; int bw_idx;
	uint8_t *slab; uint32_t cap, top;
	const gaba_arena_t *ar[2];               /* bound at the first fill: a sections come from ar[0], b sections from ar[1] */
	ScalarOut *dout;
	std::vector<gaba_fill_t *> fills; std::vector<gaba_pos_pair_t *> pps; std::vector<gaba_score_t *> scores;
	hipStream_t stream;
};

/* a host section -> device descriptor: the arena whose uploaded range holds it, mirrored pointers (gaba.h:151-155) included */
static bool section_to_dev(gaba_dp_t *dp, int side, gaba_section_t const *s, Sec *out)
{
	const uint64_t eou = 0x800000000000ull;
	uint64_t p = (uint64_t)(uintptr_t)s->base; bool rev = false;
	if(p >= eou) { p = 2 * eou - p - s->len; rev = true; }          /* gaba_mirror */
	const gaba_arena_t *hit = NULL;
	for(gaba_arena_t *ar : g_arenas) { const uint64_t h = (uint64_t)(uintptr_t)ar->host; if(p >= h && p + s->len <= h + ar->n) { hit = ar; break; } }
	if(!hit) { fprintf(stderr, "[minialign_amd] gaba_dp: section %u does not lie in an uploaded arena (gaba_arena_upload)\n", s->id); return false; }
	if(dp->ar[side] == NULL) { dp->ar[side] = hit; }
	if(dp->ar[side] != hit) { fprintf(stderr, "[minialign_amd] gaba_dp: all %c-side sections of a context must come from one arena\n", side ? 'b' : 'a'); return false; }
	*out = Sec{ s->id, s->len, p - (uint64_t)(uintptr_t)hit->host, (uint32_t)side, rev ? 1u : 0u };
	return true;
}

gaba_dp_t *gaba_dp_init_bw(gaba_t const *ctx, int bw_idx)
{
	if(!ctx || bw_idx < 0 || bw_idx > 2) return NULL;
	gaba_dp_t *dp = new gaba_dp_t();
	dp->ctx = ctx; dp->bw_idx = bw_idx; dp->cap = 64u << 20; dp->top = SLAB_HEAD; dp->ar[0] = dp->ar[1] = NULL;
	if(hipMalloc(&dp->slab, dp->cap) != hipSuccess || hipMalloc(&dp->dout, sizeof(ScalarOut)) != hipSuccess || hipStreamCreate(&dp->stream) != hipSuccess) {
		fprintf(stderr, "[minialign_amd] gaba_dp_init: no device workspace\n"); delete dp; return NULL;
	}
	HIP_OK(hipMemcpy(dp->slab, ctx->droots, SLAB_HEAD, hipMemcpyDeviceToDevice), NULL);
	return dp;
}
gaba_dp_t *gaba_dp_init(gaba_t const *ctx) { return gaba_dp_init_bw(ctx, 0); }
void gaba_dp_flush(gaba_dp_t *dp)
{
	if(!dp) return;
	dp->top = SLAB_HEAD;
	for(gaba_fill_t *f : dp->fills) free(f);
	for(gaba_pos_pair_t *q : dp->pps) free(q);
	for(gaba_score_t *q : dp->scores) free(q);
	dp->fills.clear(); dp->pps.clear(); dp->scores.clear();
}
void gaba_dp_clean(gaba_dp_t *dp)
{
	if(!dp) return;
	gaba_dp_flush(dp);
	(void)hipFree(dp->slab); (void)hipFree(dp->dout); (void)hipStreamDestroy(dp->stream);
	delete dp;
}
static gaba_fill_t *scalar_fill(gaba_dp_t *dp, uint32_t prev_tail, gaba_section_t const *a, uint32_t apos, gaba_section_t const *b, uint32_t bpos, uint32_t pridx)
{
	Sec sa, sb;
	if(!dp || !a || !b || !section_to_dev(dp, 0, a, &sa) || !section_to_dev(dp, 1, b, &sb)) return NULL;
	SeqArena da = { dp->ar[0]->pk, dp->ar[0]->nm }, db = { dp->ar[1]->pk, dp->ar[1]->nm };
	hipLaunchKernelGGL(gaba_scalar_fill_kernel, dim3(1), dim3(64), 0, dp->stream, dp->ctx->hc, da, db, dp->slab, dp->top, dp->cap,
		prev_tail, dp->bw_idx, sa, apos, sb, bpos, pridx, dp->dout);
	ScalarOut o;
	HIP_OK(hipGetLastError(), NULL); HIP_OK(hipMemcpyAsync(&o, dp->dout, sizeof(o), hipMemcpyDeviceToHost, dp->stream), NULL); HIP_OK(hipStreamSynchronize(dp->stream), NULL);
	if(o.err) { fprintf(stderr, "[minialign_amd] gaba_dp_fill: device workspace exhausted (%u B)\n", dp->cap); return NULL; }
	dp->top = o.top;
	gaba_fill_t *f = (gaba_fill_t *)calloc(1, sizeof(gaba_fill_t));
	f->aid = o.f.aid; f->bid = o.f.bid; f->ascnt = o.f.ascnt; f->bscnt = o.f.bscnt; f->apos = o.f.apos; f->bpos = o.f.bpos; f->max = o.f.max; f->status = o.f.status;
	f->reserved[0] = o.tail;
	dp->fills.push_back(f);
	return f;
}
gaba_fill_t *gaba_dp_fill_root(gaba_dp_t *dp, gaba_section_t const *a, uint32_t apos, gaba_section_t const *b, uint32_t bpos, uint32_t pridx)
{
	return scalar_fill(dp, NIL, a, apos, b, bpos, pridx);
}
gaba_fill_t *gaba_dp_fill(gaba_dp_t *dp, gaba_fill_t const *prev, gaba_section_t const *a, gaba_section_t const *b, uint32_t pridx)
{
	if(!prev) return NULL;
	return scalar_fill(dp, prev->reserved[0], a, 0, b, 0, pridx);
}
/* gaba.h:329: never called on the mapper's path, no COMBINED branch in the reference (gaba.c:2452-2472); NULL is the reference's "unmergeable" answer */
gaba_fill_t *gaba_dp_merge(gaba_dp_t *dp, gaba_fill_t const *const *sec, uint8_t const *qofs, uint32_t cnt) { (void)dp; (void)sec; (void)qofs; (void)cnt; return NULL; }
gaba_pos_pair_t *gaba_dp_search_max(gaba_dp_t *dp, gaba_fill_t const *fill)
{
	if(!dp || !fill || !dp->ar[0] || !dp->ar[1]) return NULL;
	SeqArena da = { dp->ar[0]->pk, dp->ar[0]->nm }, db = { dp->ar[1]->pk, dp->ar[1]->nm };
	hipLaunchKernelGGL(gaba_scalar_search_kernel, dim3(1), dim3(64), 0, dp->stream, dp->ctx->hc, da, db, dp->slab, dp->top, dp->cap, fill->reserved[0], dp->dout);
	ScalarOut o;
	HIP_OK(hipGetLastError(), NULL); HIP_OK(hipMemcpyAsync(&o, dp->dout, sizeof(o), hipMemcpyDeviceToHost, dp->stream), NULL); HIP_OK(hipStreamSynchronize(dp->stream), NULL);
	gaba_pos_pair_t *q = (gaba_pos_pair_t *)calloc(1, sizeof(gaba_pos_pair_t));
	q->aid = o.pp.aid; q->bid = o.pp.bid; q->apos = o.pp.apos; q->bpos = o.pp.bpos; q->plen = o.pp.plen;
	dp->pps.push_back(q);
	return q;
}
gaba_alignment_t *gaba_dp_trace(gaba_dp_t *dp, gaba_fill_t const *fill, gaba_alloc_t const *alloc)
{
	if(!dp || !fill || !dp->ar[0] || !dp->ar[1]) return NULL;
	SeqArena da = { dp->ar[0]->pk, dp->ar[0]->nm }, db = { dp->ar[1]->pk, dp->ar[1]->nm };
	/* the path cannot be longer than the two sequences walked: bounded by the fill's positions */
	const uint64_t max_plen = (uint64_t)(fill->apos + fill->bpos) + 4 * 1024 * 1024;
	const uint64_t words = (std::min<uint64_t>(max_plen, dp->cap / 8) + 31) / 32 + 8;
	const uint32_t max_seg = 64;
	uint32_t *dpath = NULL; Segment *dseg = NULL;
	HIP_OK(hipMalloc(&dpath, words * 4), NULL); HIP_OK(hipMalloc(&dseg, max_seg * sizeof(Segment)), NULL);
	hipLaunchKernelGGL(gaba_scalar_trace_kernel, dim3(1), dim3(64), 0, dp->stream, dp->ctx->hc, da, db, dp->slab, dp->top, dp->cap,
		fill->reserved[0], dpath, words, dseg, max_seg, dp->dout);
	ScalarOut o;
	HIP_OK(hipGetLastError(), NULL); HIP_OK(hipMemcpyAsync(&o, dp->dout, sizeof(o), hipMemcpyDeviceToHost, dp->stream), NULL); HIP_OK(hipStreamSynchronize(dp->stream), NULL);
	gaba_alignment_t *aln = NULL;
	if(!o.err && o.ao.status == 1) {
		const uint64_t pw = ((uint64_t)o.ao.plen + 31) / 32 + 2;
		/* one allocation: header | path words (two header words {plen, 0x40000000} live in plen / padding, gaba.h:217) | segments */
		const size_t bytes = sizeof(gaba_alignment_t) + pw * 4 + o.ao.slen * sizeof(gaba_path_section_t);
		/* the caller's allocator when one is given (gaba.c:3263-3265); it and its handle ride in the two reserved words for gaba_dp_res_free */
		aln = (gaba_alignment_t *)(alloc && alloc->lmalloc ? alloc->lmalloc(alloc->opaque, bytes) : malloc(bytes));
		if(!aln) { (void)hipFree(dpath); (void)hipFree(dseg); return NULL; }
		memset(aln, 0, bytes);
		if(alloc && alloc->lmalloc) { aln->reserved[0] = alloc->opaque; aln->reserved[1] = (void *)alloc->lfree; }
		aln->score = o.ao.score; aln->identity = o.ao.identity; aln->agcnt = o.ao.agcnt; aln->bgcnt = o.ao.bgcnt; aln->dcnt = o.ao.dcnt;
		aln->slen = o.ao.slen; aln->plen = o.ao.plen; aln->padding = 0x40000000u;
		gaba_path_section_t *seg = (gaba_path_section_t *)((uint8_t *)aln + sizeof(gaba_alignment_t) + pw * 4);
		if(hipMemcpy(aln->path, dpath, pw * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(seg, dseg, o.ao.slen * sizeof(Segment), hipMemcpyDeviceToHost) != hipSuccess) { gaba_dp_res_free(dp, aln); aln = NULL; }
		else { aln->seg = seg; }
	}
	(void)hipFree(dpath); (void)hipFree(dseg);
	return aln;
}
void gaba_dp_res_free(gaba_dp_t *dp, gaba_alignment_t *aln)
{
	(void)dp;
	if(!aln) return;
	if(aln->reserved[1]) { ((gaba_lfree_t)aln->reserved[1])(aln->reserved[0], (void *)aln); } else { free(aln); }       /* gaba.c:3398-3407 */
}

/* ---- CIGAR printers over a path (gaba_parse.h:147-263): run-length decode of the path bits; host side ---- */
static inline uint64_t cg_u64(const uint64_t *ptr, int64_t pos) { int64_t rem = pos & 63; return (ptr[pos >> 6] >> rem) | ((ptr[(pos >> 6) + 1] << (63 - rem)) << 1); }
static inline uint64_t cg_lz(uint64_t x) { return x ? (uint64_t)__builtin_clzll(x) : 64; }
static inline uint64_t cg_tz(uint64_t x) { return x ? (uint64_t)__builtin_ctzll(x) : 64; }
static inline char *cg_put(char *b, uint64_t n, char op) { char t[24]; int k = 0; if(!n) t[k++] = '0'; while(n) { t[k++] = (char)('0' + n % 10); n /= 10; } while(k) *b++ = t[--k]; *b++ = op; return b; }
} /* extern "C" */
/* the two run-length walks of gaba_parse.h:168-246 (_parser_loop_rv / _fw), emitting through a functor */
template<typename F> static inline void cigar_walk_reverse(uint32_t const *path, uint64_t offset, uint64_t len, F emit)
{
	const uint64_t *p = (const uint64_t *)((uintptr_t)path & ~(uintptr_t)7);
	uint64_t ofs = (uint64_t)((int64_t)offset + (((uintptr_t)path & 4) ? 32 : 0) - 64), idx = len;
	while((int64_t)idx > 0) {
		uint64_t m = cg_lz(cg_u64(p, (int64_t)(ofs + idx))), c = m - (m > 0); if(c > idx) c = idx;
		idx -= c; if(c) emit(c, 'D');
		m = cg_lz(~cg_u64(p, (int64_t)(ofs + idx))); c = m < idx ? m : idx;
		idx -= c; if(c) emit(c, 'I');
		uint64_t s0 = idx;
		do { m = cg_lz(cg_u64(p, (int64_t)(ofs + idx)) ^ 0x5555555555555555ull); c = (m < idx ? m : idx) & ~1ull; idx -= c; } while(c == 64);
		if((s0 - idx) >> 1) emit((s0 - idx) >> 1, 'M');
	}
}
template<typename F> static inline void cigar_walk_forward(uint32_t const *path, uint64_t offset, uint64_t len, F emit)
{
	const uint64_t *p = (const uint64_t *)((uintptr_t)path & ~(uintptr_t)7);
	uint64_t lim = offset + (((uintptr_t)path & 4) ? 32 : 0) + len, ridx = len;
	while((int64_t)ridx > 0) {
		uint64_t m = cg_tz(~cg_u64(p, (int64_t)(lim - ridx))), c = m - (m > 0); if(c > ridx) c = ridx;
		ridx -= c; if(c) emit(c, 'I');
		m = cg_tz(cg_u64(p, (int64_t)(lim - ridx))); c = m < ridx ? m : ridx;
		ridx -= c; if(c) emit(c, 'D');
		uint64_t s0 = ridx;
		do { m = cg_tz(cg_u64(p, (int64_t)(lim - ridx)) ^ 0x5555555555555555ull); c = (m < ridx ? m : ridx) & ~1ull; ridx -= c; } while(c == 64);
		if((s0 - ridx) >> 1) emit((s0 - ridx) >> 1, 'M');
	}
}
extern "C" {
uint64_t gaba_dump_cigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	(void)buf_size; char *b = buf;
	cigar_walk_reverse(path, offset, len, [&](uint64_t n, char op) { b = cg_put(b, n, op); });
	*b = 0; return (uint64_t)(b - buf);
}
uint64_t gaba_dump_cigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len)
{
	(void)buf_size; char *b = buf;
	cigar_walk_forward(path, offset, len, [&](uint64_t n, char op) { b = cg_put(b, n, op); });
	*b = 0; return (uint64_t)(b - buf);
}
/* gaba.h:394-406: the same walks through a caller-supplied printer; returns the sum of what the printer returned */
uint64_t gaba_print_cigar_reverse(gaba_printer_t printer, void *fp, uint32_t const *path, uint64_t offset, uint64_t len)
{
	uint64_t clen = 0;
	cigar_walk_reverse(path, offset, len, [&](uint64_t n, char op) { clen += (uint64_t)printer(fp, n, op); });
	return clen;
}
uint64_t gaba_print_cigar_forward(gaba_printer_t printer, void *fp, uint32_t const *path, uint64_t offset, uint64_t len)
{
	uint64_t clen = 0;
	cigar_walk_forward(path, offset, len, [&](uint64_t n, char op) { clen += (uint64_t)printer(fp, n, op); });
	return clen;
}
} /* extern "C" */
/* one row of a gapped alignment (gaba_dump_seq_forward / _reverse, gaba_parse.h:380-493): the walk gives runs of 'D' (a advances), 'I' (b advances) and
 * 'M'; row A prints gaps for I, row B for D.  GABA_SEQ_RV reads backward from seq[-1] and complements. */
template<typename W> static inline uint64_t dump_seq_walk(W walk, char *buf, uint32_t conf, const uint8_t *seq, char gap)
{
	static const char fw[] = "ACGTNNNNNNNNNNNN", rv[] = "TGCANNNNNNNNNNNN";
	char *r = buf; const uint8_t *q = seq; const bool is_b = (conf & GABA_SEQ_B) != 0, rev = (conf & GABA_SEQ_RV) != 0;
	walk([&](uint64_t c, char op) {
		if((op == 'D' && is_b) || (op == 'I' && !is_b)) { memset(r, gap, c); r += c; return; }
		if(rev) { for(uint64_t t = 0; t < c; t++) *r++ = rv[*--q & 15]; } else { for(uint64_t t = 0; t < c; t++) *r++ = fw[*q++ & 15]; }
	});
	*r = 0; return (uint64_t)(r - buf);
}
/* extended CIGAR with = and X (gaba_parse.h:274-372): base getters hide the direction of the two sections; the bases compare raw (N equals N) */
struct xc_side { const uint8_t *p; bool rev; uint8_t get(uint64_t i) const { static const uint8_t comp[16] = { 3, 2, 1, 0, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4 }; return rev ? comp[p[-1 - (int64_t)i] & 15] : p[i]; } void adv(uint64_t c) { if(rev) p -= c; else p += c; } };
static inline xc_side xc_open(gaba_section_t const *sec, uint64_t pos)
{
	const uint8_t *q = sec->base + pos;
	if(sec->base < GABA_EOU) return xc_side{ q, false };
	return xc_side{ gaba_mirror(q, 0), true };
}
template<typename W, typename E> static inline void xcigar_walk(W walk, xc_side a, xc_side b, E emit)
{
	walk([&](uint64_t c, char op) {
		if(op == 'D') { a.adv(c); emit(c, 'D'); return; }
		if(op == 'I') { b.adv(c); emit(c, 'I'); return; }
		uint64_t i = 0;
		while(i < c) {
			uint64_t s0 = i; while(i < c && a.get(i) == b.get(i)) i++;
			if(i > s0) emit(i - s0, '=');
			s0 = i; while(i < c && a.get(i) != b.get(i)) i++;
			if(i > s0) emit(i - s0, 'X');
		}
		a.adv(c); b.adv(c);
	});
}
extern "C" {
uint64_t gaba_dump_seq_forward(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap)
{
	(void)buf_size;
	return dump_seq_walk([&](auto fn) { cigar_walk_forward(path, offset, len, fn); }, buf, conf, seq, gap);
}
uint64_t gaba_dump_seq_reverse(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap)
{
	(void)buf_size;
	return dump_seq_walk([&](auto fn) { cigar_walk_reverse(path, offset, len, fn); }, buf, conf, seq, gap);
}
uint64_t gaba_dump_seq_ref(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a)
{
	const bool fwd = a->base < GABA_EOU;
	return gaba_dump_seq_forward(buf, buf_size, GABA_SEQ_A | (fwd ? GABA_SEQ_FW : GABA_SEQ_RV), path, s->ppos, (uint64_t)s->alen + s->blen,
		fwd ? &a->base[s->apos] : gaba_mirror(&a->base[s->apos], 0), '-');
}
uint64_t gaba_dump_seq_query(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *b)
{
	const bool fwd = b->base < GABA_EOU;
	return gaba_dump_seq_forward(buf, buf_size, GABA_SEQ_B | (fwd ? GABA_SEQ_FW : GABA_SEQ_RV), path, s->ppos, (uint64_t)s->alen + s->blen,
		fwd ? &b->base[s->bpos] : gaba_mirror(&b->base[s->bpos], 0), '-');
}
/* forward: from (apos, bpos) upward; reverse: from the segment's end downward, so each side is read in the opposite sense (gaba_parse.h:325-352) */
uint64_t gaba_print_xcigar_forward(gaba_printer_t printer, void *fp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b)
{
	uint64_t clen = 0;
	xcigar_walk([&](auto fn) { cigar_walk_forward(path, s->ppos, (uint64_t)s->alen + s->blen, fn); }, xc_open(a, s->apos), xc_open(b, s->bpos),
		[&](uint64_t n, char op) { clen += (uint64_t)printer(fp, n, op); });
	return clen;
}
uint64_t gaba_print_xcigar_reverse(gaba_printer_t printer, void *fp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b)
{
	uint64_t clen = 0;
	xc_side sa = xc_open(a, (uint64_t)s->apos + s->alen), sb = xc_open(b, (uint64_t)s->bpos + s->blen); sa.rev = !sa.rev; sb.rev = !sb.rev;
	xcigar_walk([&](auto fn) { cigar_walk_reverse(path, s->ppos, (uint64_t)s->alen + s->blen, fn); }, sa, sb, [&](uint64_t n, char op) { clen += (uint64_t)printer(fp, n, op); });
	return clen;
}
uint64_t gaba_dump_xcigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b)
{
	(void)buf_size; char *q = buf;
	xcigar_walk([&](auto fn) { cigar_walk_forward(path, s->ppos, (uint64_t)s->alen + s->blen, fn); }, xc_open(a, s->apos), xc_open(b, s->bpos),
		[&](uint64_t n, char op) { q = cg_put(q, n, op); });
	*q = 0; return (uint64_t)(q - buf);
}
uint64_t gaba_dump_xcigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b)
{
	(void)buf_size; char *q = buf;
	xc_side sa = xc_open(a, (uint64_t)s->apos + s->alen), sb = xc_open(b, (uint64_t)s->bpos + s->blen); sa.rev = !sa.rev; sb.rev = !sb.rev;
	xcigar_walk([&](auto fn) { cigar_walk_reverse(path, s->ppos, (uint64_t)s->alen + s->blen, fn); }, sa, sb, [&](uint64_t n, char op) { q = cg_put(q, n, op); });
	*q = 0; return (uint64_t)(q - buf);
}
/* gaba_dp_calc_score (gaba.h:357-371, gaba.c:3409-3560): score, identity, match / mismatch / gap counts of one segment, recomputed on the host from the
 * path and the two sections.  What callers of the reference get is its public wrapper, which always dispatches to the linear-model build of this
 * function (gaba_wrap.h:446-454): gaps are charged gi per region + ge per base whatever the model, the short-gap counts stay zero and adj is never set
 * -- reproduced as such.  The forward walk hands over gaps in pieces of at most 64 (63 for the first kind), so a longer gap counts as several
 * regions; the lookup index is a | shift(b) with N -> 4 | 2 of the p-fetch tables (gaba.c:866-905).  The object lives until gaba_dp_flush. */
gaba_score_t *gaba_dp_calc_score(gaba_dp_t *dp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b)
{
	if(!dp || !path || !s || !a || !b) return NULL;
	const Consts &c = dp->ctx->hc;
	const int ofs = 2 * (c.gi + c.ge);
	int8_t sbt[16]; for(int i = 0; i < 16; i++) sbt[i] = (int8_t)(((const int8_t *)c.sb)[i] - ofs);
	static const uint8_t shift_b[16] = { 0, 4, 8, 12, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2 }, comp_a[16] = { 3, 2, 1, 0, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4 }, compshift_b[16] = { 12, 8, 4, 0, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2 };
	const bool arev = a->base >= GABA_EOU, brev = b->base >= GABA_EOU;
	const uint8_t *ap = !arev ? &a->base[s->apos] : gaba_mirror(&a->base[s->apos], 0), *bp = !brev ? &b->base[s->bpos] : gaba_mirror(&b->base[s->bpos], 0);
	int64_t gac[2] = { 0, 0 }, gbc[2] = { 0, 0 };       /* { bases, regions } */
	uint64_t xc = 0, dc = 0; int64_t score = 0;
	const uint64_t *p = (const uint64_t *)((uintptr_t)path & ~(uintptr_t)7);
	const uint64_t len = (uint64_t)s->alen + s->blen;
	uint64_t lim = s->ppos + (((uintptr_t)path & 4) ? 32 : 0) + len, ridx = len;
	while((int64_t)ridx > 0) {                                                                 /* _parser_loop_fw, gaba_parse.h:147-167 */
		uint64_t m = cg_tz(~cg_u64(p, (int64_t)(lim - ridx))), n = m - (m > 0); if(n > ridx) n = ridx;
		ridx -= n;                                                                              /* insertion: b advances */
		if(brev) bp -= n; else bp += n;
		gac[0] += (int64_t)n; gac[1] += n > 0;
		m = cg_tz(cg_u64(p, (int64_t)(lim - ridx))); n = m < ridx ? m : ridx;
		ridx -= n;                                                                              /* deletion: a advances */
		if(arev) ap -= n; else ap += n;
		gbc[0] += (int64_t)n; gbc[1] += n > 0;
		do {
			m = cg_tz(cg_u64(p, (int64_t)(lim - ridx)) ^ 0x5555555555555555ull); n = (m < ridx ? m : ridx) & ~1ull; ridx -= n;
			const uint64_t d = n >> 1;
			for(uint64_t t = 0; t < d; t++) {
				const uint8_t av = arev ? comp_a[ap[-1 - (int64_t)t] & 15] : ap[t], bv = brev ? compshift_b[bp[-1 - (int64_t)t] & 15] : shift_b[bp[t] & 15];
				const int8_t sc = sbt[(av | bv) & 15];
				score += sc; xc += sc < 0;
			}
			dc += d; if(arev) ap -= d; else ap += d; if(brev) bp -= d; else bp += d;
		} while(n == 64);
	}
	gaba_score_t *sc = (gaba_score_t *)calloc(1, sizeof(gaba_score_t));
	if(!sc) return NULL;
	dp->scores.push_back(sc);
	sc->score = score - (int64_t)c.gi * (gac[1] + gbc[1]) - (int64_t)c.ge * (gac[0] + gbc[0]);
	sc->identity = dc > 0 ? (double)(dc - xc) / (double)dc : 0.0;
	sc->agcnt = (uint32_t)gbc[0]; sc->bgcnt = (uint32_t)gac[0]; sc->mcnt = (uint32_t)(dc - xc); sc->xcnt = (uint32_t)xc;
	sc->aicnt = (uint32_t)gbc[1]; sc->bicnt = (uint32_t)gac[1];
	return sc;
}
/* gaba_dp_save_stack / gaba_dp_flush_stack (gaba.h:280-289): the bump pointer of the context's device workspace */
struct gaba_stack_s { uint32_t top; size_t n_fill, n_pp; };
gaba_stack_t const *gaba_dp_save_stack(gaba_dp_t *dp)
{
	if(!dp) return NULL;
	gaba_stack_t *s = (gaba_stack_t *)malloc(sizeof(gaba_stack_t));
	s->top = dp->top; s->n_fill = dp->fills.size(); s->n_pp = dp->pps.size();
	return s;
}
void gaba_dp_flush_stack(gaba_dp_t *dp, gaba_stack_t const *stack)
{
	if(!dp || !stack) return;
	dp->top = stack->top;
	while(dp->fills.size() > stack->n_fill) { free(dp->fills.back()); dp->fills.pop_back(); }
	while(dp->pps.size() > stack->n_pp) { free(dp->pps.back()); dp->pps.pop_back(); }
	free((void *)stack);
}

} /* extern "C" */
