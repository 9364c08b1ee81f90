/*
 * ora_mm.h -- TEST INFRASTRUCTURE.  CPU restatement (plain C) of the reference's mapper:
 * minimizer sketch, index build + lookup, seed collection, array-based chaining, extension driver,
 * post-map (prune / supplementary / MAPQ) and the SAM printer (/root/reference/minialign.c, ksort.h).
 * Builds on ora_gaba.c for the DP.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it; the product never does.
 *
 * Parity status: PINNED -- byte-identical SAM against the compiled reference (oracle/_ref/minialign) on the
 * seeded synthetic sets of tests/ (tests/test_oracle_mm.py), and against committed golden SAM in tests/golden/.
 */
#ifndef ORA_MM_H
#define ORA_MM_H
#include <stdint.h>
#include <stdio.h>
#include "ora_gaba.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	/* indexing, minialign.c:6146-6149 */
	uint32_t k, w, b, n_frq; float frq[16];
	/* mapping, minialign.c:6151-6160 */
	uint32_t wlen, glen, min_score; float min_ratio;
	og_params_t p;
	char const *arg_line;       /* @PG CL: text */
	uint32_t min_len;           /* -L, minialign.c:6077, 6145 */
	/* output (minialign.c:5880-5967): flag = -X 0x01 | -P 0x08 | -A 0x10 and bit 0 when -R is given; tags = bits 1 << MM_xx of -T.  The printer
	 * ORs the two into one word (minialign.c:5677) -- so -P also switches IH on and -T IH also omits secondaries: kept */
	uint64_t flag, tags;
	char *rg_line, *rg_id;      /* -R, unescaped line and the "ID:..." token */
	uint32_t keep_qual;         /* -Q */
	uint32_t format;            /* -O: 0 sam, 1 maf, 2 blast6, 5 paf (minialign.c:2543-2549, 5940) */
	uint32_t ava;               /* -X (MM_AVA in the mapper's flag word, minialign.c:5965, 6377) */
	uint32_t circ_set; char *circ_names;   /* -c: given at all / comma list of circular reference names (NULL or empty: all), minialign.c:2457, 5986 */
} om_opt_t;
enum { OM_RG = 0, OM_CO = 1, OM_NH = 2, OM_IH = 3, OM_AS = 4, OM_XS = 5, OM_NM = 6, OM_SA = 7, OM_MD = 8, OM_CG = 9, OM_ID = 10, OM_SQ = 11 };     /* minialign.c:2530-2538 */

/* defaults (minialign.c:6141-6162) followed by a preset string such as "pacbio" or "ont.1dsq" (minialign.c:5846-5900);
 * returns nonzero on unknown preset */
int om_opt_init(om_opt_t *o, char const *preset);
/* defaults, then the options of an argv in order (-x presets and the single-letter options of minialign.c:5990-6099, `-k15` or `-k 15`), then
 * mm_opt_check_sanity (minialign.c:6097); positional arguments go to files[]; returns nonzero on an option the reference would reject */
int om_opt_parse(om_opt_t *o, int argc, char const *const *argv, char const **files, int max_files, int *n_files);

typedef struct {
	char *name; uint32_t l_name;
	uint8_t *seq; uint32_t l_seq;       /* 0..4 per base, minialign.c:214-229 */
	char *qual;                         /* NULL unless kept */
	char *comment;                      /* NULL unless kept (-T CO): text after the name, tabs turned into spaces, trailing spaces cut (minialign.c:2030-2036) */
} om_seq_t;
typedef struct { om_seq_t *a; uint64_t n; } om_seqs_t;
om_seqs_t om_read_fasta(char const *fn);        /* FASTA / FASTQ, plain text (bseq_read_fasta, minialign.c:1996) */
om_seqs_t om_read_fasta_ex(char const *fn, int keep_qual, int keep_comment);
extern int om_read_error;      /* set by the readers when the text is not FASTA / FASTQ as bseq_read_fasta accepts it */
void om_seqs_drop_short(om_seqs_t *s, uint32_t min_len);    /* the -L filter of the reader (minialign.c:2077) */
void om_seqs_free(om_seqs_t *s);

typedef struct om_idx_s om_idx_t;
om_idx_t *om_idx_build(om_opt_t const *o, om_seq_t const *ref, uint32_t n_ref);     /* mm_idx_gen, minialign.c:2951 */
void om_idx_free(om_idx_t *mi);
uint32_t om_idx_occ(om_idx_t const *mi, uint32_t i);
/* mm_idx_get, minialign.c:2728: returns the (pos, rid) list of a minimizer; values are u64 = pos | rid << 32 */
uint64_t const *om_idx_get(om_idx_t const *mi, uint64_t minier, uint32_t *n);

/* mm_sketch, minialign.c:2410: appends the minimizer words (cap excluded) to out (must hold 4 * len / w + 256); returns count */
uint64_t om_sketch(uint32_t w, uint32_t k, uint8_t const *seq, uint32_t len, uint64_t *out);

typedef struct om_align_s om_align_t;
om_align_t *om_align_init(om_opt_t const *o, om_idx_t const *mi);       /* mm_align_init + mm_tbuf_init, minialign.c:4671, 4499 */
void om_align_free(om_align_t *a);

typedef struct {
	uint32_t aid, mapq;                 /* mm_aln_t, minialign.c:3260 */
	og_alignment_t *a;
} om_aln_t;
typedef struct {
	uint32_t n_all, n_uniq;             /* mm_reg_t, minialign.c:3264 */
	om_aln_t *aln;
} om_reg_t;
om_reg_t *om_align_seq(om_align_t *a, uint32_t l_seq, uint8_t const *seq);    /* mm_align_seq, minialign.c:4427; NULL = unmapped */
void om_reg_free(om_reg_t *r);

/* stage taps for the parity tests (valid after om_align_seq / om_stage_* on the same context) */
typedef struct { uint32_t upos, rid, vpos, lid; } om_seed_t;
uint64_t om_stage_seed(om_align_t *a, uint32_t l_seq, uint8_t const *seq, uint64_t iter, om_seed_t const **seeds);   /* mm_seed(i), minialign.c:3500 (i = 0..iter run in order) */
uint64_t om_stage_chain(om_align_t *a, uint64_t const **roots);       /* mm_chain, minialign.c:3702; roots[i] = plen | lid << 32 */
void om_counters(om_align_t const *a, uint64_t out[8]);               /* work counters: fills, vectors, ... */

/* SAM (minialign.c:5096-5426); default tag set (none) */
void om_sam_header(FILE *fp, om_opt_t const *o, om_seq_t const *ref, uint32_t n_ref);
void om_sam_record(FILE *fp, om_seq_t const *ref, om_seq_t const *q, om_reg_t const *reg);
/* with the optional tags, read group, qualities and -P of o (minialign.c:5204-5426) */
void om_sam_record_opt(FILE *fp, om_opt_t const *o, om_seq_t const *ref, om_seq_t const *q, om_reg_t const *reg);
/* the record of a read in the format o->format selects (SAM, or MAF / BLAST6 / PAF: minialign.c:5427-5625; those print nothing for unmapped reads) */
void om_print_record(FILE *fp, om_opt_t const *o, om_seq_t const *ref, om_seq_t const *q, om_reg_t const *reg);

/* whole program: `minialign -x<preset> ref.fa reads.fa > out` (minialign.c:6365-6447) */
int om_main(char const *preset, char const *ref_fn, char const *query_fn, FILE *out, char const *arg_line, double *map_seconds, uint64_t *bases);
int om_main_opt(om_opt_t const *o, char const *ref_fn, char const *query_fn, FILE *out, double *map_seconds, uint64_t *bases);
int om_main_files(om_opt_t const *o, char const *const *files, int nf, FILE *out, double *map_seconds, uint64_t *bases);

#ifdef __cplusplus
}
#endif
#endif
