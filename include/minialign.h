/*
 * include/minialign.h -- C-ABI of the MI355X-native mapper (libminialign_amd.so): the `mm_*` entry points.
 *
 * In the reference every mapper function is `static` inside minialign.c; the contract below exports the same
 * call sequence main_align (minialign.c:6365-6447) performs, with the same names and argument meaning:
 *
 *   mm_opt_init / mm_opt_parse   <- mm_opt_init, mm_opt_parse_argv, presets   (minialign.c:6138, 5771, 5846)
 *   mm_idx_gen / mm_idx_destroy  <- mm_idx_gen, mm_idx_destroy                 (minialign.c:2951, 2703)
 *   mm_idx_dump / mm_idx_load    <- mm_idx_dump, mm_idx_load                   (minialign.c:3070, 3136)
 *   mm_align_init / _destroy     <- mm_align_init, mm_align_destroy            (minialign.c:4671, 4650)
 *   mm_align_file                <- mm_align_file + mm_print_sam_*             (minialign.c:4725, 5096-5426)
 *   mm_main                      <- main                                       (minialign.c:6451)
 *
 * What runs where: sketch + index lookup + seed expansion, seed sort + chaining and the banded extension
 * (fill / max search / traceback) run as HIP kernels on gfx950 (minialign_amd/csrc/mm_device.hpp,
 * gaba_device.hpp); FASTA parsing, index construction, post-map (prune / supplementary / MAPQ) and SAM
 * formatting are host C++.  No CPU fallback exists for the device stages: without a HIP device the calls
 * fail (NULL / non-zero) and say so on stderr.
 */
#ifndef MINIALIGN_AMD_MINIALIGN_H
#define MINIALIGN_AMD_MINIALIGN_H
#include <stdint.h>
#include <stdio.h>
#include "gaba.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_opt_s mm_opt_t;
typedef struct mm_idx_s mm_idx_t;
typedef struct mm_align_s mm_align_t;

/* options: defaults of minialign.c:6141-6162; mm_opt_parse applies "-x preset" strings and the single-letter options of
 * minialign.c:5990-6099 in argv order, `-k15` or `-k 15`: sketch and index (-k -w -B -f -L, -c [names] for circular references), scores (-a -b -e -p -q -r -Y), mapping
 * (-s -m -W -G), output (-T tags, -R read group, -Q qualities, -P), -d (index file to write), -t -v -1 -2 (accepted; they size the
 * reference's host pipeline), with the reference's range checks and mm_opt_check_sanity (minialign.c:6097).  Returns 0 on success, nonzero
 * on anything the reference rejects.  -X (all versus all), -A and -C follow the reference, where only -X changes what is mapped. */
mm_opt_t *mm_opt_init(void);
int mm_opt_parse(mm_opt_t *o, int argc, char const *const *argv, char const **files, int max_files, int *n_files);
void mm_opt_destroy(mm_opt_t *o);

/* index over a FASTA file (host build; value-list order and occurrence thresholds as the reference) */
mm_idx_t *mm_idx_gen(mm_opt_t const *o, char const *ref_fasta);
void mm_idx_destroy(mm_idx_t *mi);
/* index files (mm_idx_dump / mm_idx_load, minialign.c:3070, 3136; `-d idx.mai` and a reference argument ending in .mai).  The layout is this
 * library's own (the reference's is a memory image of its tables and changes between its releases); a file may hold several blocks.
 * mm_idx_dump returns 0 on success; mm_idx_load returns the next block or NULL, with *at_eof telling a clean end of file from a damaged block. */
int mm_idx_dump(mm_idx_t const *mi, FILE *fp);
mm_idx_t *mm_idx_load(FILE *fp, int *at_eof);
uint32_t mm_idx_n_seq(mm_idx_t const *mi);
/* test entry: the 2-bit + N-mask reference a device-built index holds in HBM against the host's conversion of the same text; returns the number of differing bases (-1: nothing to compare) */
int64_t mm_idx_ref_check(mm_idx_t const *mi);
uint32_t mm_idx_occ(mm_idx_t const *mi, uint32_t i);
uint32_t mm_idx_max_len(mm_idx_t const *mi);          /* length of the longest reference sequence */
/* mm_idx_get (minialign.c:2728) on the host copy: writes up to max values (pos | rid << 32), returns the count */
uint32_t mm_idx_get(mm_idx_t const *mi, uint64_t minier, uint64_t *out, uint32_t max);
/* mm_sketch (minialign.c:2410) on the host: minimizer stream words (hash << 8 | strand << 7 | index inside the block of w, minialign.c:2402) of a sequence given
 * one byte per base (0..3, 4 = N), and -- when pos != NULL -- the decoded k-mer start positions (minialign.c:2831-2835); returns the count, writes at most max */
uint32_t mm_sketch(uint8_t const *seq, uint32_t len, uint32_t w, uint32_t k, uint64_t *words, uint32_t *pos, uint32_t max);

/* device context: uploads the reference (2-bit + N mask) and the flattened index, builds the DP constants.
 * Where the reference takes `-t N` worker threads between one source and one drain thread (minialign.c:1013-1048, 4565-4645, 4729), this context takes GPUs: it spans
 * every visible device from the current one on (HIP_VISIBLE_DEVICES chooses them; MM_DEVICES=n in the environment takes n), with a replica of the index in each
 * device's HBM, and the streaming entries (mm_align_file, mm_map_text, mm_map_file, mm_map_packed, mm_map_reads; the command line) deal their batches to device x
 * lane and write in input order -- one process, no collective.  The per-batch entries below (mm_align_batch, mm_batch_*) run on the first device.
 * mm_align_devices: how many devices the context spans. */
mm_align_t *mm_align_init(mm_opt_t const *o, mm_idx_t const *mi);
void mm_align_destroy(mm_align_t *a);
int mm_align_devices(mm_align_t const *a);
/* maps every read of a FASTA / FASTQ file and prints SAM records (no header) to `out`; returns 0 on success */
int mm_align_file(mm_align_t *a, char const *reads_fn, FILE *out);
void mm_print_sam_header(mm_align_t const *a, FILE *out, char const *arg_line);

/* in-memory batch entry (used by bench.py and the parity tests): reads are given as one byte per base (0..4),
 * concatenated, with per-read lengths; SAM text for the batch is appended to *sam (malloc'd / realloc'd by the
 * library, free with free()).  names may be NULL (then "r<i>"). */
int mm_align_batch(mm_align_t *a, uint8_t const *bases, uint32_t const *lens, char const *const *names, uint32_t n_reads,
	char **sam, uint64_t *sam_len);

/* structured results, as mm_align_seq hands them to its caller (minialign.c:4427; mm_aln_t :3260, mm_reg_t :3264): aln[0 .. n_uniq) are the primary and
 * supplementary alignments, aln[n_uniq .. n_all) the secondary ones; mapq is the 16x fixed-point value (>> 4 for SAM); each mm_aln_t is directly followed by
 * its gaba_alignment_t (gaba.h).  regs[i] = NULL for an unmapped read; every non-NULL entry is one block to be released with mm_reg_free. */
typedef struct { uint32_t aid, mapq; gaba_alignment_t a[]; } mm_aln_t;
typedef struct mm_reg_s { uint32_t n_all, n_uniq; mm_aln_t const *aln[]; } mm_reg_t;
int mm_align_batch_regs(mm_align_t *a, uint8_t const *bases, uint32_t const *lens, uint32_t n_reads, mm_reg_t **regs);
void mm_reg_free(mm_reg_t *r);
/* mm_align_seq (minialign.c:4427) for one read (a one-read batch; qid / lmm as in the reference's argument list, unused); NULL = unmapped; mm_reg_free releases it */
mm_reg_t const *mm_align_seq(mm_align_t *a, uint32_t l_seq, uint8_t const *seq, uint32_t qid, void *lmm);
/* the state reads share through the reference's thread buffer (self->rlen, minialign.c:3864: the length of the reference the previous read loaded last): callers
 * that split one read set over several contexts pass it from the end of one part to the start of the next (minialign_amd/multi.py) */
uint32_t mm_align_get_carry(mm_align_t const *a);
void mm_align_set_carry(mm_align_t *a, uint32_t rlen);

/* the same batch in three phases, so that callers can keep inputs resident in HBM and time the hot path alone:
 * upload (parse-free H2D of 2-bit packed reads) -> run (K1 sketch/lookup/expand, K2 sort/chain, K3 extend, in rounds;
 * results stay in HBM) -> finish (D2H, post-map, SAM text appended to *sam).  mm_batch_run may be repeated. */
typedef struct mm_reads_s mm_reads_t;
typedef struct mm_batch_s mm_batch_t;
mm_reads_t *mm_reads_load(char const *fn);
/* the same, keeping the text of the file: batches cut from such a set are 2-bit packed on the device from that text (the table of minialign.c:223-229 applied to every
 * byte of a record's sequence lines but '\n'), which is what the command-line program does; mm_pack_check compares that with the host packing over the whole set and
 * returns the number of differing arena words (0 = identical, -1 = no text / failure) */
mm_reads_t *mm_reads_load_text(char const *fn);
int64_t mm_pack_check(mm_align_t *a, mm_reads_t const *r);
/* the same set packed on the device and brought back: one code byte per base (0..3, 4 = N) read after read into codes, bases per read into lens; returns the number of
 * reads, -1 on failure.  For checkers with a reader of their own (tests/test_pack_gpu.py compares with the oracle's). */
int64_t mm_pack_fetch(mm_align_t *a, mm_reads_t const *r, uint8_t *codes, uint64_t cap, uint32_t *lens, uint32_t max_reads);
/* part `part` of `n_parts` of a read file (one rank's shard of a set): a plain FASTA file is cut by bytes where a '>' starts a line and only that stretch is read;
 * anything else (gzip, FASTQ, stdin) is read whole and the part keeps its share of the records.  The parts, in order, are the file. */
mm_reads_t *mm_reads_load_part(char const *fn, uint32_t part, uint32_t n_parts);
/* ... with the reader's options of a parsed command line (-L, -Q, -T CO), as the text entries take them from the context */
mm_reads_t *mm_reads_load_part_opt(mm_opt_t const *o, char const *fn, uint32_t part, uint32_t n_parts);
void mm_reads_free(mm_reads_t *r);
/* the records of a file as the DEVICE reader finds them (mm_device.hpp K0r: record scanning; K0: base conversion + packing), nothing mapped: names / comments / qualities
 * read off the text at the offsets the scan gave, base codes brought back from the packed arena.  *host_scanned = records that went through the host's sequential FASTQ
 * reader (any FASTQ shape other than four lines per record).  NULL when the file cannot be read or is rejected.  For checkers (tests/test_reader_gpu.py: the oracle's reader). */
mm_reads_t *mm_reads_scan(mm_align_t *a, char const *fn, int keep_qual, int keep_comment, uint64_t *host_scanned);
uint32_t mm_reads_codes(mm_reads_t const *r, uint32_t i, uint8_t *out, uint32_t cap);          /* base codes (0..3, 4 = N) of read i; returns its length */
char const *mm_reads_qual(mm_reads_t const *r, uint32_t i);          /* "" when not kept */
char const *mm_reads_comment(mm_reads_t const *r, uint32_t i);       /* NULL when the record has none / not kept */
int mm_reads_append(mm_reads_t *r, char const *fn);          /* another file behind the reads already loaded; 0 on success */
char const *mm_reads_name(mm_reads_t const *r, uint32_t i);
uint32_t mm_reads_count(mm_reads_t const *r);
uint64_t mm_reads_bases(mm_reads_t const *r, uint32_t first, uint32_t n);
mm_batch_t *mm_batch_upload(mm_align_t *a, mm_reads_t const *r, uint32_t first, uint32_t n);
int mm_batch_run(mm_align_t *a, mm_batch_t *b);
/* two batches in flight: upload to lane 0 / 1 (each lane owns streams and pools, the index is shared), start with _run_async, join with
 * _wait (0 on success).  The launch tail and the latency-bound stages of one batch are then filled by the other batch's work. */
mm_batch_t *mm_batch_upload_lane(mm_align_t *a, mm_reads_t const *r, uint32_t first, uint32_t n, int lane);
int mm_batch_run_async(mm_align_t *a, mm_batch_t *b);
int mm_batch_wait(mm_align_t *a, mm_batch_t *b);
int mm_batch_finish(mm_align_t *a, mm_batch_t *b, char **sam, uint64_t *sam_len);
void mm_batch_free(mm_batch_t *b);
/* stage taps (tests): runs sketch + lookup + expansion and the first round's sort + chain over the batch and stops; then, for one read, the number of minimizers,
 * its seed array as mm_seed leaves it (minialign.c:3500: sorted, sentinel last; 4 words per seed: upos, rid, vpos, lid = INT32_MAX) and its chain roots as
 * mm_chain leaves them (minialign.c:3702: plen | lid << 32, longest first).  0 on success; the batch can be run normally afterwards. */
int mm_batch_tap(mm_align_t *a, mm_batch_t *b, uint32_t read, uint32_t *n_min, uint32_t *seeds, uint32_t seeds_cap, uint32_t *n_seeds, uint64_t *roots, uint32_t roots_cap, uint32_t *n_roots);
/* the minimizer stream words of that read (minialign.c:2402) as K1 computed them; after mm_batch_tap on the same batch.  Returns the count (writes at most cap), -1 on error */
int64_t mm_batch_tap_sketch(mm_align_t *a, mm_batch_t *b, uint32_t read, uint64_t *words, uint32_t cap);
int mm_set_device(int dev);

/* the streaming form main_align uses (minialign.c:6413-6436 with the pipeline of mm_align_file, :4725): the batches of a read set go through `lanes` lanes of
 * the device context, several in flight, with pack / H2D in front and D2H / post-map / text behind overlapped on host threads, and the text is handed to `sink`
 * in input order (a nonzero return from the sink stops the run).  mm_batch_pack prepares a batch ahead of time (2-bit packing on the host, no device work), so a
 * caller can time the map phase from packed reads in host memory to text in host memory; mm_map_reads packs on the fly.  lanes <= 0: the default (4).
 * Returns 0 on success.  The carried reference length (mm_align_get_carry) enters at the first batch and is left at its value after the last read. */
typedef int (*mm_sam_sink_t)(void *opaque, uint32_t batch, char const *text, uint64_t len);
mm_batch_t *mm_batch_pack(mm_reads_t const *r, uint32_t first, uint32_t n);
uint32_t mm_batch_reads(mm_batch_t const *b);
uint32_t mm_batch_pack_all(mm_reads_t const *r, uint32_t first, uint32_t n, mm_batch_t **out, uint32_t max);   /* all batches of a span; returns how many there are */
/* a read set split over several contexts: what another carried reference length at the start of the stream mapped last would change -- 0 nothing, 1 read
 * *first_affected decides differently (re-map from there), 2 undecided within the recorded head (re-map the part); mm_carry_after(i): the value behind read i */
int mm_carry_check(mm_align_t const *a, uint32_t truth, uint32_t *first_affected);
uint32_t mm_carry_after(mm_align_t const *a, uint32_t i);          /* UINT32_MAX when read i lies beyond the recorded head (4 096 reads; fewer when a batch had to be split) */
/* where, in the text the last stream handed to its sink, the records of read i begin (for a stream of n <= 4 096 reads, i = n gives the end of the text); UINT64_MAX
 * beyond what was recorded.  A caller that replaces the records of reads [i, j) -- the window re-map of minialign_amd/multi.py -- cuts the text at these offsets,
 * whatever the output format and the read names are. */
uint64_t mm_head_offset(mm_align_t const *a, uint32_t i);
/* for a stream over a text (mm_map_text / mm_map_file): where the record of read i begins in that text, so that a window of reads can be mapped again as a slice of it */
uint64_t mm_head_text_offset(mm_align_t const *a, uint32_t i);
uint32_t mm_head_count(mm_align_t const *a);          /* reads of the last stream whose head was recorded (at most 4 096; 0 = the stream had no read) */
int mm_map_packed(mm_align_t *a, mm_batch_t *const *batches, uint32_t n_batches, int lanes, mm_sam_sink_t sink, void *opaque);
int mm_map_reads(mm_align_t *a, mm_reads_t const *r, uint32_t first, uint32_t n, int lanes, mm_sam_sink_t sink, void *opaque);
/* the whole input path on the device: the FASTA / FASTQ text of a read set in host memory (mm_map_text) or a file (mm_map_file: plain files are mapped, gzip / stdin are
 * read) goes to HBM as it is, records are found there (bseq_read_fasta's scanning, minialign.c:1996-2090), bases converted and packed there, and the batches stream through
 * the lanes as the reader cuts them; this is what mm_align_file and the command-line program run.  -L, -Q and -T CO of the options apply.  0 on success. */
int mm_map_text(mm_align_t *a, char const *text, uint64_t len, int lanes, mm_sam_sink_t sink, void *opaque);
int mm_map_file(mm_align_t *a, char const *fn, int lanes, mm_sam_sink_t sink, void *opaque);

/* timing / work counters of everything run since the last reset */
typedef struct {
	double k1_ms, k2_ms, k3_ms;             /* summed kernel times (HIP events on the launch stream) */
	uint64_t k1_launches, k2_launches, k3_launches;
	uint64_t reads, bases, minimizers, seeds, fills, vectors, blocks, traces, trace_steps, reruns;
	double host_post_ms, host_sam_ms, wall_ms;
	/* wave-cycles (s_memtime ticks summed over all waves of the extension kernel): DP fill, max search, traceback, whole wave */
	uint64_t k3_cycles_fill, k3_cycles_leaf, k3_cycles_trace, k3_cycles_total;
	uint64_t k3_cycles_next;                /* ... of which in the next-seed search of the extension driver (mm_search_load_next) */
	uint64_t k3_cycles_max, k3_waves;       /* lifetime of the longest-living wave (summed over launches), persistent waves per launch */
	/* the same for the first-round sort + chain kernel, and the number of reads whose seed array did not fit the LDS */
	uint64_t k2_cycles_sort, k2_cycles_chain, k2_cycles_total, k2_reads_hbm;
	uint64_t pool_grows, batch_splits;          /* batches run again with larger device pools (a pool or a per-read cap overflowed) / batches mapped in halves because no pool size held them */
	uint64_t pool_regrows;                      /* batches whose sketch launch was repeated because they asked for more seed / rescue / root entries than the run had seen (pools sized to the demand) */
	uint64_t text_bytes; double reader_ms;      /* the text readers of the streams so far: bytes brought to HBM, time of the uploader threads (summed over devices) */
	uint64_t d2h_bytes, cigar_bytes_device;     /* result bytes brought back from the devices (result pools + either path words or the CIGAR text made on the device, K4); bytes of that text */
	uint64_t k3_aborts;                         /* times the watchdog called extension launches off because one did not end (the batches then ran again in the safe mode) */
} mm_stats_t;
void mm_stats(mm_align_t *a, mm_stats_t *out, int reset);
/* test entry: the CIGAR run lengths of path bits [ppos, ppos + len) of the path that starts at pool[path_word] (header words { plen, 0x40000000 } in front, gaba.h:217) by the
 * code the device kernel runs (K4, csrc/mm_cigar.hpp), on the host: what gaba_dp_print_cigar_reverse prints (gaba_parse.h:168-221); returns the characters (out NULL: counts) */
uint64_t mm_cigar_walk(uint32_t const *pool, uint64_t path_word, uint64_t ppos, uint64_t len, char *out);

/* the command-line program: `minialign [-x preset] [opts] ref.fa reads.{fa,fq} > out.sam` */
int mm_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
