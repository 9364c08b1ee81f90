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

#include "mm_reader.hpp"          /* K0r, K0 */
#include "mm_sketch.hpp"          /* K1 */
#include "mm_sort_chain.hpp"      /* K2s, K2p, K2w, K2a */
#include "mm_extend.hpp"          /* K3 */

} /* namespace mm */
