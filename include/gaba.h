/*
 * include/gaba.h -- C-ABI of the MI355X-native banded-extension library (libminialign_amd.so).
 *
 * Drop-in surface for the reference's libgaba header (/root/reference/gaba.h): the structs below keep the
 * reference's sizes and field order (gaba.h:81-98 params 40 B, :151-155 section 16 B, :169-178 fill 64 B,
 * :183-188 pos pair, :193-200 segment 32 B; sizes asserted at gaba.c:220-224), status codes are gaba.h:45-51.
 *
 * The GPU entry point is *batched*: one wavefront per job runs the reference's call sequence
 *   gaba_dp_fill_root (gaba.c:2110) -> gaba_dp_fill* (gaba.c:2161) -> gaba_dp_search_max (gaba.c:2776)
 *   -> gaba_dp_trace (gaba.c:3372)
 * exactly as minialign's mm_extend_core drives it (minialign.c:4075-4112), on sequences resident in HBM
 * (2-bit packed + N mask).  The per-call scalar API of gaba.h:245-371 (one fill per call on host pointers)
 * maps onto a 1-job batch; see INTEGRATION.md for the binding a maintainer would add.
 *
 * All functions return NULL / a negative value (and print the reason to stderr) when no gfx950 device is
 * usable: there is no CPU fallback in this library.
 */
#ifndef MINIALIGN_AMD_GABA_H
#define MINIALIGN_AMD_GABA_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum gaba_status {                 /* gaba.h:45-51 */
	GABA_CONT = 0, GABA_UPDATE_A = 0x000f, GABA_UPDATE_B = 0x00f0, GABA_TERM = 0x8000, GABA_OOM = 0x0400
};

struct gaba_params_s {             /* gaba.h:81-98 */
	int8_t score_matrix[16];
	int8_t gi, ge, gfa, gfb;
	int8_t xdrop;
	uint8_t filter_thresh;
	void *reserved;
	uint64_t _pad;
};
typedef struct gaba_params_s gaba_params_t;

struct gaba_fill_s {               /* gaba.h:169-178 */
	uint32_t aid, bid;
	uint32_t ascnt, bscnt;
	uint64_t apos, bpos;
	int64_t max;
	uint32_t status;
	uint32_t reserved[5];
};
typedef struct gaba_fill_s gaba_fill_t;

struct gaba_pos_pair_s {           /* gaba.h:183-188 */
	uint32_t aid, bid;
	uint32_t apos, bpos;
	uint64_t plen;
};
typedef struct gaba_pos_pair_s gaba_pos_pair_t;

struct gaba_segment_s {            /* gaba.h:193-200 */
	uint32_t aid, bid;
	uint32_t apos, bpos;
	uint32_t alen, blen;
	uint64_t ppos;
};
typedef struct gaba_segment_s gaba_path_section_t;
#define gaba_plen(seg)                ( (seg)->alen + (seg)->blen )          /* gaba.h:200: path length of a segment */

struct gaba_section_s {            /* gaba.h:131-135; base >= GABA_EOU selects the mirrored (reverse-complement) view */
	uint32_t id, len;
	uint8_t const *base;
};
typedef struct gaba_section_s gaba_section_t;
#define GABA_EOU                    ( (uint8_t const *)0x800000000000 )
#define gaba_mirror(base, len)      ( GABA_EOU + (uint64_t)GABA_EOU - (uint64_t)(base) - (uint64_t)(len) )
#define gaba_rev(pos, len)          ( (len) + (uint64_t)(len) - (uint64_t)(pos) - 1 )          /* gaba.h:155 (deprecated there; kept for callers that still use it) */
#define gaba_build_section(_id, _base, _len)  ( (struct gaba_section_s){ .id = (_id), .len = (_len), .base = (uint8_t const *)(_base) } )

struct gaba_alignment_s {          /* gaba.h:205-220 */
	void *reserved[2];
	int64_t score;
	double identity;
	uint32_t agcnt, bgcnt;
	uint32_t dcnt;
	uint32_t slen;
	struct gaba_segment_s const *seg;
	uint32_t plen, padding;
	uint32_t path[];
};
typedef struct gaba_alignment_s gaba_alignment_t;

typedef struct gaba_context_s gaba_t;          /* gaba.h:117 */
typedef struct gaba_dp_context_s gaba_dp_t;    /* gaba.h:160 */
typedef struct gaba_arena_s gaba_arena_t;      /* a set of sequences resident in HBM */

/* gaba_init (gaba.h:245, gaba_wrap.h:245-295): validates the scores exactly as gaba_init_check_score
 * (gaba.c:3614-3640) and uploads the score vectors and the three root blocks (64/32/16 cells). */
gaba_t *gaba_init(gaba_params_t const *params);
void gaba_clean(gaba_t *ctx);                  /* gaba.h:252 */

/* upload `n` bases (one byte per base: A,C,G,T = 0..3, N = 4 -- minialign.c:214-220) as 2-bit + N-mask */
gaba_arena_t *gaba_arena_upload(uint8_t const *bases, uint64_t n);
void gaba_arena_free(gaba_arena_t *ar);

/*
 * The per-call API of gaba.h:266-357, same names and argument meaning.  Sections point into host arrays that were
 * handed to gaba_arena_upload (the bases themselves are read from the HBM copy); all a-side sections of one context
 * must lie in one arena and all b-side sections in one arena.  Every call is one single-wavefront launch on the
 * context's persistent device workspace, so this form is for drop-in use and testing -- throughput comes from
 * gaba_dp_extend_batch.  gaba_dp_init selects the 64-cell band, gaba_dp_init_bw the 32 / 16-cell variants the
 * reference exposes through its wrapper (gaba_wrap.h:57).  NULL on error (reason on stderr).
 */
gaba_dp_t *gaba_dp_init(gaba_t const *ctx);                    /* gaba.h:266 */
gaba_dp_t *gaba_dp_init_bw(gaba_t const *ctx, int bw_idx);     /* 0: 64 cells, 1: 32, 2: 16 */
void gaba_dp_flush(gaba_dp_t *dp);                             /* gaba.h:273: drops every fill / position of the context */
void gaba_dp_clean(gaba_dp_t *dp);                             /* gaba.h:295 */
typedef struct gaba_stack_s gaba_stack_t;                      /* gaba.h:124 */
gaba_stack_t const *gaba_dp_save_stack(gaba_dp_t *dp);         /* gaba.h:280: remember the workspace level ... */
void gaba_dp_flush_stack(gaba_dp_t *dp, gaba_stack_t const *stack);   /* gaba.h:287: ... and drop everything filled since (consumes `stack`) */
gaba_fill_t *gaba_dp_fill_root(gaba_dp_t *dp, gaba_section_t const *a, uint32_t apos, gaba_section_t const *b, uint32_t bpos, uint32_t pridx);   /* gaba.h:302 */
gaba_fill_t *gaba_dp_fill(gaba_dp_t *dp, gaba_fill_t const *prev_sec, gaba_section_t const *a, gaba_section_t const *b, uint32_t pridx);          /* gaba.h:315 */
gaba_pos_pair_t *gaba_dp_search_max(gaba_dp_t *dp, gaba_fill_t const *sec);                                                                       /* gaba.h:339 */
/* gaba.h:329 (gaba.c:2581): merging of up to 14 bands on one anti-diagonal.  minialign never calls it and the reference has no COMBINED-model branch in
 * merge_slice_vectors (gaba.c:2452-2472); exported so that libgaba callers link, always answers NULL ("unmergeable", the reference's own error return). */
gaba_fill_t *gaba_dp_merge(gaba_dp_t *dp, gaba_fill_t const *const *sec, uint8_t const *qofs, uint32_t cnt);
/* gaba.h:61-75: the caller may supply where alignment objects live; NULL = this library's own heap.  lmalloc gets one request per alignment
 * (header + path + segments), lfree gets that pointer back from gaba_dp_res_free. */
typedef void *(*gaba_lmalloc_t)(void *opaque, size_t size);
typedef void (*gaba_lfree_t)(void *opaque, void *ptr);
struct gaba_alloc_s { void *opaque; gaba_lmalloc_t lmalloc; gaba_lfree_t lfree; };
typedef struct gaba_alloc_s gaba_alloc_t;
gaba_alignment_t *gaba_dp_trace(gaba_dp_t *dp, gaba_fill_t const *tail, gaba_alloc_t const *alloc);                                               /* gaba.h:348 */
void gaba_dp_res_free(gaba_dp_t *dp, gaba_alignment_t *aln);                                                                                      /* gaba.h:357 */

/* one extension job: the arguments of gaba_dp_fill_root (gaba.c:2110) with host pointers replaced by
 * arena offsets; rev = 1 selects the mirrored (reverse-complement) view (gaba.h:151-155 gaba_mirror) */
typedef struct {
	uint64_t a_off; uint32_t alen, apos;
	uint64_t b_off; uint32_t blen, bpos;
	uint8_t arev, brev;
	uint8_t bw_idx;                /* 0: 64 cells, 1: 32, 2: 16 (gaba_wrap.h:57) */
	uint8_t do_trace;
} gaba_job_t;

typedef struct {
	int64_t max; uint32_t status; uint32_t aid, bid, ascnt, bscnt; uint64_t apos, bpos;
} gaba_xfill_t;

typedef struct {
	uint32_t n_fill, max_fill_idx;
	gaba_xfill_t fill[8];          /* every gaba_fill_t the call sequence produced */
	uint32_t p_aid, p_bid, p_apos, p_bpos; uint64_t p_plen;    /* gaba_dp_search_max on the max fill */
	int32_t traced;                /* 0: not requested, 1: ok, -1: path left the band (reference returns NULL) */
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, dcnt, slen, plen;
	struct gaba_segment_s seg[16];
	uint32_t n_path_words;
} gaba_xresult_t;

/*
 * Run `n` jobs.  results[n]; paths: n * path_stride words (job i's path bits start at paths + i * path_stride,
 * bit k = k-th step from the root, 1 = b-advance, as gaba_alignment_s.path, gaba.h:219).
 * Returns 0, or a negative error (-2: a job ran out of device workspace, -3: path_stride too small).
 */
int gaba_dp_extend_batch(gaba_t *ctx, gaba_arena_t const *a, gaba_arena_t const *b,
	gaba_job_t const *jobs, uint32_t n, gaba_xresult_t *results, uint32_t *paths, uint32_t path_stride);

/* gaba_dp_calc_score (gaba.h:223-243, 357-371): score, identity and counts of one segment, recomputed on the host from the path and the two sections
 * (host memory; mirrored sections allowed).  The object belongs to dp and dies at gaba_dp_flush.  agcnt / aicnt count gap bases / gap regions where a
 * advances alone, bgcnt / bicnt where b does.  As through the reference's public wrapper (gaba_wrap.h:446-454, which always calls its linear-model
 * build), gaps cost gi per region + ge per base whatever the model, and the short-gap fields and adj stay zero. */
struct gaba_score_s {
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, mcnt, xcnt, aicnt, bicnt, afgcnt, bfgcnt, aficnt, bficnt;
	int32_t adj; uint32_t reserved;
};
typedef struct gaba_score_s gaba_score_t;
gaba_score_t *gaba_dp_calc_score(gaba_dp_t *dp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b);

/* CIGAR printers over a path (gaba.h:393-420, gaba_parse.h:247-263); host side, operate on host memory.
 * `path` must be preceded by the two header words {plen, 0x40000000} as in gaba_alignment_s (gaba.h:217). */
uint64_t gaba_dump_cigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len);
uint64_t gaba_dump_cigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len);
/* gaba.h:385-406: the same through a printer callback, e.g. int pr(void *fp, uint64_t len, char c) { return fprintf(fp, "%c%lu", c, len); } */
typedef int (*gaba_printer_t)(void *, uint64_t, char);
uint64_t gaba_print_cigar_forward(gaba_printer_t printer, void *fp, uint32_t const *path, uint64_t offset, uint64_t len);
uint64_t gaba_print_cigar_reverse(gaba_printer_t printer, void *fp, uint32_t const *path, uint64_t offset, uint64_t len);
/* extended CIGAR with '=' and 'X' for one segment over its two sections (gaba.h:427-461, gaba_parse.h:274-372); bases compare raw, so N equals N */
uint64_t gaba_print_xcigar_forward(gaba_printer_t printer, void *fp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b);
uint64_t gaba_print_xcigar_reverse(gaba_printer_t printer, void *fp, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b);
uint64_t gaba_dump_xcigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b);
uint64_t gaba_dump_xcigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a, gaba_section_t const *b);
/* one row of the gapped alignment as text (gaba.h:463-505, gaba_parse.h:380-529): conf = GABA_SEQ_A | GABA_SEQ_B with GABA_SEQ_FW | GABA_SEQ_RV; seq is one
 * byte per base (0..3, 4 = N), read forward from seq or, with GABA_SEQ_RV, backward from seq[-1] and complemented.  buf needs len + 1 bytes. */
#define GABA_SEQ_FW                 ( 0x00 )
#define GABA_SEQ_RV                 ( 0x01 )
#define GABA_SEQ_A                  ( 0x00 )
#define GABA_SEQ_B                  ( 0x02 )
uint64_t gaba_dump_seq_forward(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap);
uint64_t gaba_dump_seq_reverse(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap);
uint64_t gaba_dump_seq_ref(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *a);
uint64_t gaba_dump_seq_query(char *buf, uint64_t buf_size, uint32_t const *path, gaba_path_section_t const *s, gaba_section_t const *b);

/* last kernel time of gaba_dp_extend_batch in milliseconds (HIP events on the launch stream) and work counters */
typedef struct { double kernel_ms; uint64_t vectors, blocks, trace_steps; } gaba_batch_stats_t;
void gaba_last_stats(gaba_t *ctx, gaba_batch_stats_t *out);

#ifdef __cplusplus
}
#endif
#endif
