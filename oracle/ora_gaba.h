/*
 * ora_gaba.h -- TEST INFRASTRUCTURE.  CPU restatement (plain C, no intrinsics) of the reference's
 * adaptive-banded DP library (libgaba: /root/reference/gaba.c, gaba.h, gaba_parse.h).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (minialign_amd/) never includes, links or executes anything from oracle/.
 *
 * Parity status: PINNED -- checked against the compiled reference (oracle/_ref/libgaba_ref.so,
 * built from /root/reference by oracle/Makefile) on seeded random inputs (tests/test_oracle_gaba.py)
 * and against the committed golden vectors in tests/golden/ that were generated from it.
 *
 * Struct layouts follow gaba.h:81-220 (params 40 B, section 16 B, fill 64 B, segment 32 B).
 */
#ifndef ORA_GABA_H
#define ORA_GABA_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum og_status {            /* gaba.h:45-51 */
	OG_CONT = 0, OG_UPDATE_A = 0x000f, OG_UPDATE_B = 0x00f0, OG_TERM = 0x8000
};

typedef struct {            /* gaba.h:81-98 */
	int8_t score_matrix[16];
	int8_t gi, ge, gfa, gfb;
	int8_t xdrop;
	uint8_t filter_thresh;
	void *reserved;
	uint64_t _pad;
} og_params_t;

typedef struct {            /* gaba.h:151-155; base >= OG_EOU means mirrored (reverse-complement) */
	uint32_t id, len;
	uint8_t const *base;
} og_section_t;
#define OG_EOU              ( (uint8_t const *)0x800000000000ULL )
#define og_mirror(base, len) ( OG_EOU + (uint64_t)OG_EOU - (uint64_t)(base) - (uint64_t)(len) )

typedef struct {            /* gaba.h:169-178 */
	uint32_t aid, bid;
	uint32_t ascnt, bscnt;
	uint64_t apos, bpos;
	int64_t max;
	uint32_t status;
	uint32_t reserved[5];
} og_fill_t;

typedef struct {            /* gaba.h:183-188 */
	uint32_t aid, bid;
	uint32_t apos, bpos;
	uint64_t plen;
} og_pos_pair_t;

typedef struct {            /* gaba.h:193-200 */
	uint32_t aid, bid;
	uint32_t apos, bpos;
	uint32_t alen, blen;
	uint64_t ppos;
} og_segment_t;

typedef struct {            /* gaba.h:205-220 (pointer members replaced by explicit arrays) */
	int64_t score;
	double identity;
	uint32_t agcnt, bgcnt, dcnt;
	uint32_t slen;
	og_segment_t *seg;      /* slen entries, seg[0] is the segment closest to the root */
	uint32_t plen;
	uint32_t *path;         /* (plen + 31) / 32 + 2 words; bit i = i-th step from the root, 1 = b-advance */
} og_alignment_t;

typedef struct og_ctx_s og_ctx_t;
typedef struct og_dp_s og_dp_t;

og_ctx_t *og_init(og_params_t const *p);                /* gaba_init, gaba.c:3848 + gaba_wrap.h:245; NULL if scores are rejected */
void og_clean(og_ctx_t *ctx);
og_dp_t *og_dp_init(og_ctx_t const *ctx);               /* gaba_dp_init, gaba.c:3895 */
void og_dp_flush(og_dp_t *dp);                          /* gaba_dp_flush, gaba.c:3969 */
void og_dp_clean(og_dp_t *dp);

/* bw_idx: 0 -> 64 cells, 1 -> 32, 2 -> 16 (gaba_wrap.h:57, `&dp[n]`) */
og_fill_t *og_dp_fill_root(og_dp_t *dp, int bw_idx, og_section_t const *a, uint32_t apos, og_section_t const *b, uint32_t bpos, uint32_t pridx);
og_fill_t *og_dp_fill(og_dp_t *dp, og_fill_t const *prev, og_section_t const *a, og_section_t const *b, uint32_t pridx);
og_pos_pair_t *og_dp_search_max(og_dp_t *dp, og_fill_t const *fill);
og_alignment_t *og_dp_trace(og_dp_t *dp, og_fill_t const *fill);   /* malloc'd; free with og_aln_free */
void og_aln_free(og_alignment_t *aln);

/* gaba_parse.h:259 gaba_dump_cigar_reverse / :258 forward */
uint64_t og_dump_cigar_reverse(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len);
uint64_t og_dump_cigar_forward(char *buf, uint64_t buf_size, uint32_t const *path, uint64_t offset, uint64_t len);
/* _parser_loop_rv (gaba_parse.h:168-188) with a callback per nonzero run, in the order the reverse dumpers see them: op is 'D', 'I' or 'M' */
/* gaba_dump_seq_reverse (gaba_parse.h:445-493): one row of a gapped alignment; conf = OG_SEQ_A / OG_SEQ_B | OG_SEQ_FW / OG_SEQ_RV, seq one byte per
 * base (0..4), read forward from seq or, with OG_SEQ_RV, backward from seq[-1] and complemented; returns the length written (plus a NUL) */
enum { OG_SEQ_FW = 0, OG_SEQ_RV = 1, OG_SEQ_A = 0, OG_SEQ_B = 2 };
uint64_t og_dump_seq_reverse(char *buf, uint64_t buf_size, uint32_t conf, uint32_t const *path, uint64_t offset, uint64_t len, uint8_t const *seq, char gap);
void og_parse_path_reverse(uint32_t const *path, uint64_t offset, uint64_t len, void (*fn)(void *ctx, char op, uint64_t cnt), void *ctx);

/* convenience used by the tests: same record as oracle/ref_harness/gaba_ref_shim.c:shim_result_t */
typedef struct {
	int64_t max; uint32_t status; uint32_t aid, bid, ascnt, bscnt; uint64_t apos, bpos;
} og_xfill_t;
typedef struct {
	uint32_t n_fill; uint32_t max_fill_idx;
	og_xfill_t fill[8];
	uint32_t p_aid, p_bid, p_apos, p_bpos; uint64_t p_plen;
	int32_t traced;
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, dcnt, slen, plen;
	og_segment_t seg[16];
	uint32_t n_path_words;
} og_xresult_t;
extern uint64_t og_wrap_events;      /* test instrumentation, see ora_gaba.c */
int og_extend(og_dp_t *dp, int bw_idx,
	uint8_t const *a, uint32_t alen, uint32_t apos, int arev,
	uint8_t const *b, uint32_t blen, uint32_t bpos, int brev,
	int do_trace, og_xresult_t *res, uint32_t *path_out);

#ifdef __cplusplus
}
#endif
#endif
