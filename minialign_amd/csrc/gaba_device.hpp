/*
 * gaba_device.hpp -- CDNA4 (gfx950) device-side adaptive banded Smith-Waterman-Gotoh extension.
 *
 * One wavefront = one band: lane l holds cell l of the current anti-diagonal (W = 64 fills the wave,
 * W = 32 / 16 use the low lanes).  Difference vectors dh/dv/de/df, the running delta / drop and the
 * 16-bit middle delta live in VGPRs; the two lane shifts of the recurrence are single DPP moves
 * (wave_shr:1 / wave_shl:1); the band-steering accumulator, X-drop test and all section bookkeeping
 * run on the scalar unit.  Traceback masks are accumulated per lane over the 32 vectors of a block
 * (one bit per vector) and written as one coalesced 1 KiB store per block -- a transposed layout of the
 * reference's per-vector lane masks, so no ballots are needed in the hot loop.
 *
 * Replaces (reference, behaviour only): gaba.c:735-2203 (fill), :2604-2817 (max search),
 * :2820-3407 (trace), and the arch/x86_64_{sse41,avx2} vector shim it is written on.  Results
 * (fill max/status/positions, max position, path bits, segments, gap counts, identity) are
 * bit-identical to the reference; the int8 wrap / saturation points are called out inline.
 *
 * Integer add/max/compare and lane shifts only: no MFMA (this is not a contraction).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gaba {

constexpr int BLK = 32;
constexpr int MIN_BULK_BLOCKS = 32;              /* gaba.c:186 */
constexpr int INIT_FETCH_POS = -1;               /* gaba.c:194-195 */
enum : int { CONT = 0, ZERO = 0x01, TERM = 0x80, STAT_MASK = 0x81, HEAD = 0x20, MERGE = 0x40, ROOT = 0x60 };   /* gaba.c:678-690 */
enum : uint32_t { UPDATE_A = 0x000f, UPDATE_B = 0x00f0, STATUS_TERM = 0x8000 };                            /* gaba.h:45-51 */
enum : int { MODEL_AFFINE = 1, MODEL_COMBINED = 2 };
constexpr uint32_t NIL = 0xffffffffu;

/* ---- sequence arenas: 2-bit packed bases (16 per u32) + 1-bit N mask (32 per u32) ---- */
struct SeqArena {
	const uint32_t *pk;
	const uint32_t *nm;
};
/* a section: gaba_section_t (gaba.h:151-155) with the mirrored-pointer trick replaced by a flag */
struct Sec {
	uint32_t id, len;
	uint64_t off;        /* first base of the section in its arena */
	uint32_t arena;      /* 0: a-side arena (reference), 1: b-side arena (reads), 2: 96 x N tail (minialign.c:4512-4519) */
	uint32_t rev;        /* 1: mirrored (reverse-complement) */
};

/* ---- per-job state in HBM (offsets are relative to the wave's slab) ---- */
struct Fill {            /* gaba_fill_s, gaba.h:169-178 */
	uint32_t aid, bid, ascnt, bscnt;
	uint64_t apos, bpos;
	int64_t max;
	uint32_t status;
	uint32_t reserved[5];
};
struct Tail {            /* gaba_joint_tail_s, gaba.c:351-366 */
	uint8_t ch[64];
	int8_t xd[64];
	int16_t md[64];
	int16_t mdrop; uint16_t istat; uint32_t pridx;
	uint32_t ridx[2], adv[2];
	uint32_t tail;       /* previous tail (slab offset) or NIL */
	uint32_t last;       /* _last_block(tail) */
	int32_t W; uint32_t _pad;
	Sec sec[2];          /* the sections this fill ran on (replaces atptr / btptr) */
	Fill f;
};
static_assert(sizeof(Tail) == 256 + 8 + 16 + 16 + 48 + 64, "tail layout");
struct BlkMisc {         /* tail of gaba_block_s, gaba.c:311-314; `link` overlays max_mask for head blocks (gaba.c:321) */
	int8_t acc, xstat, acnt, bcnt;
	uint32_t dir_mask;
	union { uint64_t max_mask; uint32_t link; };
};
struct Blk {             /* gaba_block_s, gaba.c:308-315: 1296 B */
	uint32_t m[4][64];   /* h, v, e, f; lane-major, bit (31 - k) = vector k of the block */
	uint32_t diff[64];   /* dh | dv << 8 | de << 16 | df << 24 per lane */
	BlkMisc s;
};
static_assert(sizeof(Blk) == 1296, "block layout");

struct PosPair { uint32_t aid, bid, apos, bpos; uint64_t plen; };        /* gaba.h:183-188 */
struct Segment { uint32_t aid, bid, apos, bpos, alen, blen; uint64_t ppos; };   /* gaba.h:193-200 */

/* scoring constants (uniform; from gaba_init, gaba.c:3644-3680, 3811-3830) */
struct Consts {
	int32_t model;
	uint32_t sb[4];      /* 16 x int8 substitution scores (+2(gi+ge) bias) */
	int32_t adjh, adjv, ofsh, ofsv, gfh, gfv;
	int32_t tx;
	int32_t gi, ge, gfa, gfb;
	double imx, xmx;
	/* single-v_perm score lookup (see step()): 1 = sb[a | 2] (the score against a b side N) is one value for all a: every block; 2 = it is not: every block that sees
	 * no N on the b side (fill_block) */
	int32_t fast_score;
	int32_t score_n;     /* sb[a | 2] (selector 4 reads its byte 0) */
	uint32_t arow[5];    /* arow[a] = { sb[a | 0], sb[a | 4], sb[a | 8], sb[a | 12] } for a = 0..3, N */
};
constexpr uint32_t ROOT_STRIDE = sizeof(Blk) + sizeof(Tail);            /* [blk][tail] x {64, 32, 16} at the head of each slab */
constexpr uint32_t SLAB_HEAD = 3 * ROOT_STRIDE;

/* ---- wave primitives ---- */
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int sext8(int x) { return __builtin_amdgcn_sbfe(x, 0, 8); }
__device__ __forceinline__ int sext16(int x) { return __builtin_amdgcn_sbfe(x, 0, 16); }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int rdfirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rdfirst64(uint64_t v)
{
	return ((uint64_t)(uint32_t)rdfirst((int)(v >> 32)) << 32) | (uint32_t)rdfirst((int)v);
}
/* zero-filling variants: bound_ctrl makes the lane without a source read 0, no `old` operand to set up */
__device__ __forceinline__ int shift_up0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
__device__ __forceinline__ int shift_dn0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }
/* lane i <- lane i - 1, lane 0 <- fill  (_bsl_n, v64i8.h:152) */
__device__ __forceinline__ int shift_up(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
/* lane i <- lane i + 1, lane 63 <- fill (_bsr_n, v64i8.h:164) */
__device__ __forceinline__ int shift_dn(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }

/* ---- sequence fetch (gaba.c:846-1119): value of the k-th base of a section as seen by the band ---- */
__device__ __forceinline__ uint32_t fetch_code(const SeqArena *ar, const Sec &s, uint32_t i)
{
	if(s.arena == 2) { return 4; }
	/* past the end of the section: an extension may start a few bases beyond a sequence (a seed at the wrap of a circular reference; minialign.c:3823-3827
	 * only pulls the position back by k).  The reference then reads the zero bytes that terminate / pad its copy of the sequence: A, or T through the
	 * complement of a mirrored section. */
	if(i >= s.len) { return s.rev ? 3u : 0u; }
	uint64_t p = s.off + (s.rev ? (uint64_t)(s.len - 1 - i) : (uint64_t)i);
	/* constant indices only: `ar` is a two-element local array of the caller, a variable index would keep it in scratch memory and
	 * put a dependent scratch load in front of every sequence fetch (two HBM-latency round trips per block instead of one) */
	const uint32_t *pk = s.arena ? ar[1].pk : ar[0].pk, *nm = s.arena ? ar[1].nm : ar[0].nm;
	uint32_t c = (pk[p >> 4] >> (2 * (p & 15))) & 3;
	uint32_t n = (nm[p >> 5] >> (p & 31)) & 1;
	c = s.rev ? 3 - c : c;                      /* comp_mask_a / compshift_mask_b, gaba.c:852-866 */
	return n ? 4 : c;
}
__device__ __forceinline__ uint32_t enc_b(uint32_t c) { return c == 4 ? 2 : c << 2; }    /* shift_mask_b, gaba.c:856-859 */

/* ---- the band, in registers ---- */
struct Band {
	int dh, dv, de, df;      /* int8 semantics, sign-extended */
	int delta, drop;         /* int8: wrapping add / saturating sub (gaba.c:1649-1650) */
	int ach, bch;            /* sequence windows: lane 0 = newest a base, lane W-1 = newest b base (gaba.c:806-809) */
	uint32_t mh, mv, me, mf; /* per-lane traceback bits of the current block, newest vector in bit 0 */
};

/* uniform per-fill work area (gaba_reader_work_s, gaba.c:407-432) */
struct Work {
	int W;
	uint32_t rlim[2], id[2];
	Sec sec[2];
	uint32_t pridx; int32_t ofsd;
	uint32_t rem[2], sridx[2];
	uint32_t tail;           /* previous tail */
	uint32_t acnt, bcnt;     /* consumed in the current block */
	uint32_t la_cnt;         /* valid look-ahead lengths are implied by the caller */
	uint32_t dmask; int32_t dacc;
	uint32_t nblk;           /* blocks written so far in this fill, head included */
	uint32_t blk0;           /* slab offset of the head block of this fill */
};

#ifdef GABA_TRACE_PROF
__device__ unsigned long long g_trace_prof[16];
#define TPROF_T0() const unsigned long long tp0_ = __builtin_amdgcn_s_memtime()
#define TPROF_ADD(_i) { tp_cy[(_i)] += __builtin_amdgcn_s_memtime() - tp0_; tp_n[(_i)]++; }
#define TPROF_DECL() unsigned long long tp_cy[4] = { 0, 0, 0, 0 }; unsigned long long tp_n[4] = { 0, 0, 0, 0 }
#define TPROF_FLUSH() { if(x.lane == 0) { for(int i_ = 0; i_ < 4; i_++) { atomicAdd(&g_trace_prof[i_], tp_cy[i_]); atomicAdd(&g_trace_prof[8 + i_], tp_n[i_]); } } }
#else
#define TPROF_T0()
#define TPROF_ADD(_i)
#define TPROF_DECL()
#define TPROF_FLUSH()
#endif
struct Ctx {                 /* everything a device routine needs */
	Consts c;                /* by value: lives in SGPRs, never re-loaded inside the DP loops */
	const SeqArena *ar;
	uint8_t *slab;           /* this wave's arena */
	uint32_t top, cap;       /* bump pointer / capacity (bytes) */
	int lane;
	int err;                 /* sticky: 1 = slab exhausted */
	bool no_trace;           /* the fills of this call sequence will only be searched for their maximum, never walked back: skip the traceback masks */
	uint32_t n_vec, n_blk, n_tr;   /* work counters (uniform): DP vectors, blocks stored, traceback steps */
};

__device__ __forceinline__ Blk *blk_at(Ctx &x, uint32_t off) { return (Blk *)(x.slab + off); }
__device__ __forceinline__ Tail *tail_at(Ctx &x, uint32_t off) { return (Tail *)(x.slab + off); }
__device__ __forceinline__ uint32_t root_blk(int bw_idx) { return (uint32_t)bw_idx * ROOT_STRIDE; }
__device__ __forceinline__ uint32_t root_tail(int bw_idx) { return (uint32_t)bw_idx * ROOT_STRIDE + (uint32_t)sizeof(Blk); }

__device__ __forceinline__ uint32_t slab_alloc(Ctx &x, uint32_t bytes)
{
	uint32_t off = x.top;
	bytes = (bytes + 15u) & ~15u;
	if(off + bytes > x.cap) { x.err = 1; return SLAB_HEAD; }      /* keep running inside bounds; the job is reported as failed */
	x.top = off + bytes;
	return off;
}

/*
 * One anti-diagonal (gaba.c:1576-1699) as a single hand-scheduled block per direction: lane shift of the window and of the
 * two diff vectors that move (DPP wave_shr / wave_shl), score lookup (_shuf_n(sb, a | b), gaba.c:1605), the recurrence
 * (gaba.c:1576-1640), the delta / drop update (gaba.c:1647-1655); the direction accumulator (_dir_update, gaba.c:761) is
 * fed from the returned t.  The loop is bound by VALU issue (4 cycles per wave64 integer instruction), so the block is
 * written to the instruction:
 *
 *  - every int8 quantity is kept sign-extended in a 32-bit lane; additions whose result is compared later use the SDWA
 *    form `dst_sel:BYTE_0 dst_unused:UNUSED_SEXT` (add + wrap to int8 in one instruction).  The wrap is required: band
 *    edge lanes do run into it (oracle/ora_gaba.c:og_wrap_events counts such events on the test corpus);
 *  - delta and drop live in the top byte of their lane (value << 24) while a block is filled: a plain 32-bit add is then
 *    the wrapping int8 add and v_sub_i32 + clamp is the saturating int8 subtract (_subs_n), and t is produced in the same
 *    position by an SDWA `dst_sel:BYTE_3` subtract;
 *  - score lookup, general form: the 16-entry byte table sits in four VGPRs, two v_perm_b32 + a select on bit 3.  Fast form
 *    (Consts.fast_score; blocks that see a `b` side N take the general form unless the table has one score for that case): the `a` window carries, per
 *    lane, the four scores of its base against b = A, C, G, T as one dword, the `b` window carries a v_perm selector, and
 *    the lookup is a single v_perm_b32;
 *  - compares write SGPR pairs, the mask algebra of the COMBINED model runs on the scalar unit, and each of the four
 *    traceback bit columns takes its new bit with one add-with-carry (m = m + m + bit);
 *  - the shifted copies of the two moving vectors go to scratch registers and every result is written to its home register,
 *    so both directions leave the same registers live (no copies at the join) and no register-to-register move is needed;
 *  - the schedule keeps the gfx950 wait-state rules by construction: >= 2 instructions between a VALU write of an SGPR /
 *    VCC and its VALU read, >= 1 between an SDWA write and its consumer, >= 2 between a VALU write and a DPP read of the
 *    same VGPR (the shifted registers are last written >= 5 instructions before the block ends), >= 1 before a
 *    v_readlane of a fresh VGPR.
 */
#define GABA_SX         " dst_sel:BYTE_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"
#define GABA_S24        " dst_sel:BYTE_3 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define GABA_SHR        " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define GABA_SHL        " wave_shl:1 row_mask:0xf bank_mask:0xf"
/* right: x0 = shifted dh, x1 = shifted df; down: x0 = shifted dv, x1 = shifted de */
#define GABA_PRE_RIGHT \
	"v_readlane_b32 %[nb], %[look], %[ai]\n\t" \
	"v_mov_b32_dpp %[ach], %[ach]" GABA_SHR "\n\t" \
	"v_mov_b32_dpp %[x0], %[dh]" GABA_SHR " bound_ctrl:0\n\t" \
	"v_mov_b32_dpp %[x1], %[df]" GABA_SHR " bound_ctrl:0\n\t" \
	"v_writelane_b32 %[ach], %[nb], 0\n\t"
#define GABA_PRE_DOWN_WIDE \
	"v_readlane_b32 %[nb], %[look], %[bi]\n\t" \
	"v_mov_b32_dpp %[bch], %[bch]" GABA_SHL "\n\t" \
	"v_mov_b32_dpp %[x0], %[dv]" GABA_SHL " bound_ctrl:0\n\t" \
	"v_mov_b32_dpp %[x1], %[de]" GABA_SHL " bound_ctrl:0\n\t" \
	"v_writelane_b32 %[bch], %[nb], 63\n\t"
#define GABA_PRE_DOWN_NARROW            /* the top lane (W - 1) of a narrow band takes the fill values */ \
	"s_mov_b32 m0, %[wm1]\n\t" \
	"v_readlane_b32 %[nb], %[look], %[bi]\n\t" \
	"v_mov_b32_dpp %[bch], %[bch]" GABA_SHL "\n\t" \
	"v_mov_b32_dpp %[x0], %[dv]" GABA_SHL " bound_ctrl:0\n\t" \
	"v_mov_b32_dpp %[x1], %[de]" GABA_SHL " bound_ctrl:0\n\t" \
	"v_writelane_b32 %[bch], %[nb], m0\n\t" \
	"v_writelane_b32 %[x0], 0, m0\n\t" \
	"v_writelane_b32 %[x1], 0, m0\n\t"
/* register names of the four inputs of the recurrence per direction */
#define GABA_R_DH "%[x0]"
#define GABA_R_DV "%[dv]"
#define GABA_R_DE "%[de]"
#define GABA_R_DF "%[x1]"
#define GABA_D_DH "%[dh]"
#define GABA_D_DV "%[x0]"
#define GABA_D_DE "%[x1]"
#define GABA_D_DF "%[df]"

/* heads: everything up to t = max(...) (score s, biased gap candidates dea / dfa, and dfh / dfv for the COMBINED model) */
#define GABA_HEAD_COMBINED(DH, DV, DE, DF) \
	"v_or_b32 %[s], %[ach], %[bch]\n\t" \
	"v_add_u32 %[dfh], %[gfh], " DV "\n\t" \
	"v_and_b32 %[t1], 7, %[s]\n\t" \
	"v_and_b32 %[s], 8, %[s]\n\t" \
	"v_sub_u32 %[dfv], %[gfv], " DH "\n\t" \
	"v_cmp_eq_u32 vcc, 0, %[s]\n\t" \
	"v_perm_b32 %[t2], %[sb1], %[sb0], %[t1]\n\t" \
	"v_perm_b32 %[t1], %[sb3], %[sb2], %[t1]\n\t" \
	"v_max3_i32 %[t], " DE ", " DF ", %[dfh]\n\t" \
	"v_cndmask_b32_sdwa %[s], %[t1], %[t2], vcc" GABA_S24 \
	"v_add_u32 %[dea], %[adjh], " DE "\n\t" \
	"v_add_u32 %[dfa], %[adjv], " DF "\n\t" \
	"v_max3_i32 %[t], %[t], %[s], %[dfv]\n\t"
#define GABA_HEAD_COMBINED_FAST(DH, DV, DE, DF) \
	"v_add_u32 %[dfh], %[gfh], " DV "\n\t" \
	"v_sub_u32 %[dfv], %[gfv], " DH "\n\t" \
	"v_perm_b32 %[s], %[cN], %[ach], %[bch]\n\t" \
	"v_add_u32 %[dea], %[adjh], " DE "\n\t" \
	"v_max3_i32 %[t], " DE ", " DF ", %[dfh]\n\t" \
	"v_add_u32 %[dfa], %[adjv], " DF "\n\t" \
	"v_max3_i32 %[t], %[t], %[s], %[dfv]\n\t"
#define GABA_HEAD_AFFINE(DH, DV, DE, DF) \
	"v_or_b32 %[s], %[ach], %[bch]\n\t" \
	"v_and_b32 %[t1], 7, %[s]\n\t" \
	"v_and_b32 %[s], 8, %[s]\n\t" \
	"v_add_u32 %[dea], %[adjh], " DE "\n\t" \
	"v_cmp_eq_u32 vcc, 0, %[s]\n\t" \
	"v_perm_b32 %[t2], %[sb1], %[sb0], %[t1]\n\t" \
	"v_perm_b32 %[t1], %[sb3], %[sb2], %[t1]\n\t" \
	"v_add_u32 %[dfa], %[adjv], " DF "\n\t" \
	"v_cndmask_b32_sdwa %[s], %[t1], %[t2], vcc" GABA_S24 \
	"s_nop 0\n\t" \
	"v_max3_i32 %[t], " DE ", " DF ", %[s]\n\t"
#define GABA_HEAD_AFFINE_FAST(DH, DV, DE, DF) \
	"v_perm_b32 %[s], %[cN], %[ach], %[bch]\n\t" \
	"v_add_u32 %[dea], %[adjh], " DE "\n\t" \
	"v_add_u32 %[dfa], %[adjv], " DF "\n\t" \
	"v_max3_i32 %[t], " DE ", " DF ", %[s]\n\t"
/* cores: mask bits and the four new vectors.  UPD writes the new dh / dv (order per direction: the home register that is
 * still an input goes last); TT produces t from the new dh (right) or dv (down), _fill_update_delta gaba.c:1647 */
#define GABA_CORE_COMBINED(DH, DV, DE, DF, UPD, TT) \
	"v_cmp_eq_u32 %[A], %[t], %[dfh]\n\t" \
	"v_cmp_eq_u32 %[B], %[t], " DE "\n\t" \
	"v_cmp_eq_u32 %[C], %[t], %[dfv]\n\t" \
	"v_cmp_eq_u32 %[D], %[t], " DF "\n\t" \
	"v_max_i32 %[de], %[dea], %[t]\n\t" \
	"v_max_i32 %[df], %[dfa], %[t]\n\t" \
	"s_andn2_b64 vcc, %[B], %[A]\n\t"                /* gh & ~gfh */ \
	"s_or_b64 %[A], %[A], %[B]\n\t"                  /* h */ \
	"s_andn2_b64 %[B], %[D], %[C]\n\t"               /* gv & ~gfv */ \
	"s_or_b64 %[C], %[C], %[D]\n\t"                  /* v */ \
	"s_mov_b64 %[D], vcc\n\t" \
	"v_addc_co_u32 %[mh], vcc, %[mh], %[mh], %[A]\n\t" \
	"v_addc_co_u32 %[mv], vcc, %[mv], %[mv], %[C]\n\t" \
	"v_cmp_ge_i32 %[A], %[t], %[dea]\n\t"            /* max(de', t) == t */ \
	"v_cmp_ge_i32 %[C], %[t], %[dfa]\n\t" \
	"v_add_u32 %[de], %[de], " DH "\n\t" \
	"v_sub_u32 %[df], %[df], " DV "\n\t" \
	UPD \
	"s_or_b64 %[A], %[A], %[D]\n\t"                  /* e */ \
	"s_or_b64 %[C], %[C], %[B]\n\t"                  /* f */ \
	TT \
	"v_addc_co_u32 %[me], vcc, %[me], %[me], %[A]\n\t" \
	"v_addc_co_u32 %[mf], vcc, %[mf], %[mf], %[C]\n\t" \
	"v_add_u32 %[delta], %[delta], %[t]\n\t" \
	"v_sub_i32 %[drop], %[drop], %[t] clamp\n\t"
#define GABA_CORE_AFFINE(DH, DV, DE, DF, UPD, TT) \
	"v_cmp_eq_u32 %[A], %[t], " DE "\n\t" \
	"v_cmp_eq_u32 %[C], %[t], " DF "\n\t" \
	"v_max_i32 %[de], %[dea], %[t]\n\t" \
	"v_max_i32 %[df], %[dfa], %[t]\n\t" \
	"v_addc_co_u32 %[mh], vcc, %[mh], %[mh], %[A]\n\t" \
	"v_addc_co_u32 %[mv], vcc, %[mv], %[mv], %[C]\n\t" \
	"v_cmp_ge_i32 %[A], %[t], %[dea]\n\t" \
	"v_cmp_ge_i32 %[C], %[t], %[dfa]\n\t" \
	"v_add_u32 %[de], %[de], " DH "\n\t" \
	"v_sub_u32 %[df], %[df], " DV "\n\t" \
	UPD \
	TT \
	"v_addc_co_u32 %[me], vcc, %[me], %[me], %[A]\n\t" \
	"v_addc_co_u32 %[mf], vcc, %[mf], %[mf], %[C]\n\t" \
	"v_add_u32 %[delta], %[delta], %[t]\n\t" \
	"v_sub_i32 %[drop], %[drop], %[t] clamp\n\t"
/* the same without the traceback bit columns (a fill that is only searched for its maximum: minialign's downward pass) */
#define GABA_CORE_NOTRACE(DH, DV, DE, DF, UPD, TT) \
	"v_max_i32 %[de], %[dea], %[t]\n\t" \
	"v_max_i32 %[df], %[dfa], %[t]\n\t" \
	"v_add_u32 %[de], %[de], " DH "\n\t" \
	"v_sub_u32 %[df], %[df], " DV "\n\t" \
	UPD \
	TT \
	"v_add_u32 %[delta], %[delta], %[t]\n\t" \
	"v_sub_i32 %[drop], %[drop], %[t] clamp\n\t"
#define GABA_UPD_RIGHT  "v_sub_u32 %[dh], %[dv], %[t]\n\t" "v_add_u32 %[dv], %[x0], %[t]\n\t"
#define GABA_UPD_DOWN   "v_add_u32 %[dv], %[dh], %[t]\n\t" "v_sub_u32 %[dh], %[x0], %[t]\n\t"
#define GABA_TT_RIGHT   "v_sub_u32 %[t], %[ofsh], %[dh]\n\t"
#define GABA_TT_DOWN    "v_add_u32 %[t], %[ofsv], %[dv]\n\t"

/* loop-invariant operands of the step, pinned in registers by the caller */
struct StepK {
	uint32_t sb0, sb1, sb2, sb3;     /* score table words (VGPR), general lookup */
	int cN;                          /* fast lookup: the score against a b side N in byte 0 (SGPR) */
	int wm1;                         /* W - 1 (SGPR) */
	int gfh, gfv, adjh, adjv, ofsh, ofsv;   /* VGPR, value << 24: plain VGPR-only adds issue at full rate, a scalar operand halves it */
};
__device__ __forceinline__ StepK step_consts(const Consts &c, int W)
{
	StepK k;
	/* the asm below wants the table in VGPRs (v_perm_b32 takes one scalar operand at most) */
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.sb0) : "s"(rdfirst((int)c.sb[0])));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.sb1) : "s"(rdfirst((int)c.sb[1])));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.sb2) : "s"(rdfirst((int)c.sb[2])));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.sb3) : "s"(rdfirst((int)c.sb[3])));
	k.cN = rdfirst((int)c.score_n);
	k.wm1 = rdfirst(W - 1);
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.gfh) : "s"(rdfirst(c.gfh) << 24));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.gfv) : "s"(rdfirst(c.gfv) << 24));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.adjh) : "s"(rdfirst(c.adjh) << 24));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.adjv) : "s"(rdfirst(c.adjv) << 24));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.ofsh) : "s"(rdfirst(c.ofsh) << 24));
	asm volatile("v_mov_b32 %0, %1" : "=v"(k.ofsv) : "s"(rdfirst(c.ofsv) << 24));
	return k;
}

#define GABA_STEP_OPERANDS \
	: [ach] "+v"(b.ach), [bch] "+v"(b.bch), [dh] "+v"(b.dh), [dv] "+v"(b.dv), [de] "+v"(b.de), [df] "+v"(b.df), \
	  [delta] "+v"(b.delta), [drop] "+v"(b.drop), [mh] "+v"(b.mh), [mv] "+v"(b.mv), [me] "+v"(b.me), [mf] "+v"(b.mf), \
	  [s] "=&v"(s), [t] "=&v"(t), [t1] "=&v"(t1), [t2] "=&v"(t2), [dfh] "=&v"(dfh), [dfv] "=&v"(dfv), [dea] "=&v"(dea), \
	  [dfa] "=&v"(dfa), [x0] "=&v"(x0), [x1] "=&v"(x1), [A] "=&s"(A), [B] "=&s"(B), [C] "=&s"(C), [D] "=&s"(D), [nb] "=&s"(nb) \
	: [look] "v"(look), [ai] "s"(ai), [bi] "s"(bi), [down] "s"(down), [sb0] "v"(k.sb0), [sb1] "v"(k.sb1), [sb2] "v"(k.sb2), [sb3] "v"(k.sb3), [wm1] "s"(k.wm1), \
	  [gfh] "v"(k.gfh), [gfv] "v"(k.gfv), [adjh] "v"(k.adjh), [adjv] "v"(k.adjv), [ofsh] "v"(k.ofsh), [ofsv] "v"(k.ofsv), [cN] "s"(k.cN) \
	: "vcc", "scc"

/* both directions live in one asm statement with a scalar branch inside, so that the compiler sees a single in-place
 * update of the band registers (two statements on an if / else make it copy all twelve at the join) */
#define GABA_STEP(pre_down, HEAD, CORE) \
	"s_cmp_lg_u32 %[down], 0\n\t" \
	"s_cbranch_scc1 .Lgaba_down_%=\n\t" \
	GABA_PRE_RIGHT HEAD(GABA_R_DH, GABA_R_DV, GABA_R_DE, GABA_R_DF) CORE(GABA_R_DH, GABA_R_DV, GABA_R_DE, GABA_R_DF, GABA_UPD_RIGHT, GABA_TT_RIGHT) \
	"s_branch .Lgaba_end_%=\n" \
	".Lgaba_down_%=:\n\t" \
	pre_down HEAD(GABA_D_DH, GABA_D_DV, GABA_D_DE, GABA_D_DF) CORE(GABA_D_DH, GABA_D_DV, GABA_D_DE, GABA_D_DF, GABA_UPD_DOWN, GABA_TT_DOWN) \
	".Lgaba_end_%=:\n\t"

/*
 * WIDE: the band spans all 64 lanes (no top-lane patching).  FAST: single-v_perm score lookup (the windows then hold score
 * rows / selectors, see fill_block_t).  down: wave-uniform direction.  look: the look-ahead (lanes 0..31 next a entries,
 * 32..63 next b entries); ai / bi: the lane of the entry coming into the window for a right / down step (the v_readlane
 * sits three instructions ahead of the v_writelane that consumes its SGPR).  b.delta / b.drop are in the << 24 domain.
 * Returns t << 24 (the per-lane score increment), final >= 4 instructions before the block ends, so a v_readlane may follow.
 */
template<int MODEL, bool WIDE, bool FAST, bool TRACE = true>
__device__ __forceinline__ int step(const StepK &k, Band &b, int look, int down, int ai, int bi)
{
	int nb, s, t, t1, t2, dfh, dfv, dea, dfa, x0, x1; uint64_t A, B, C, D;
	#define GABA_EMIT(_pre, _head, _core) asm volatile(GABA_STEP(_pre, _head, _core) GABA_STEP_OPERANDS)
	if(MODEL == MODEL_COMBINED) {
		if(TRACE) {
			if(FAST) { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_COMBINED_FAST, GABA_CORE_COMBINED); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_COMBINED_FAST, GABA_CORE_COMBINED); } }
			else { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_COMBINED, GABA_CORE_COMBINED); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_COMBINED, GABA_CORE_COMBINED); } }
		} else {
			if(FAST) { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_COMBINED_FAST, GABA_CORE_NOTRACE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_COMBINED_FAST, GABA_CORE_NOTRACE); } }
			else { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_COMBINED, GABA_CORE_NOTRACE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_COMBINED, GABA_CORE_NOTRACE); } }
		}
	} else {
		if(TRACE) {
			if(FAST) { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_AFFINE_FAST, GABA_CORE_AFFINE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_AFFINE_FAST, GABA_CORE_AFFINE); } }
			else { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_AFFINE, GABA_CORE_AFFINE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_AFFINE, GABA_CORE_AFFINE); } }
		} else {
			if(FAST) { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_AFFINE_FAST, GABA_CORE_NOTRACE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_AFFINE_FAST, GABA_CORE_NOTRACE); } }
			else { if(WIDE) { GABA_EMIT(GABA_PRE_DOWN_WIDE, GABA_HEAD_AFFINE, GABA_CORE_NOTRACE); } else { GABA_EMIT(GABA_PRE_DOWN_NARROW, GABA_HEAD_AFFINE, GABA_CORE_NOTRACE); } }
		}
	}
	#undef GABA_EMIT
	return t;
}
/* window encodings of the fast lookup.  a side: the score row of a base code (0..3, 4 = N); b side: the v_perm selector of an
 * encoded b base (0, 4, 8, 12, 2 = N): byte 3 picks the row byte (or Consts.score_n), bytes 0..2 select zero: the score comes out << 24 */
__device__ __forceinline__ int fast_arow(const Consts &c, int code)
{
	return code == 0 ? (int)c.arow[0] : (code == 1 ? (int)c.arow[1] : (code == 2 ? (int)c.arow[2] : (code == 3 ? (int)c.arow[3] : (int)c.arow[4])));
}
__device__ __forceinline__ int fast_bsel(int enc) { return ((enc == 2 ? 4 : (enc >> 2)) << 24) | 0x000c0c0c; }     /* the score lands in byte 3, zeros below */
__device__ __forceinline__ int fast_bdec(int sel) { int v = (sel >> 24) & 7; return v == 4 ? 2 : (v << 2); }

/* ---- fill state shared by the block routines ---- */
struct FillState {
	Band b;
	int xd;                  /* w.r.xd[lane] */
	int md;                  /* w.r.md[lane], int16 */
	int look;                /* look-ahead bases: lanes 0..31 = next a bases, lanes 32..63 = next b bases (b already enc_b'd) */
};

/* _fill_load_context (gaba.c:1527-1552): diff vectors come from the previous block */
__device__ __forceinline__ void load_context(Ctx &x, Work &w, FillState &f, uint32_t prev_off)
{
	const Blk *p = blk_at(x, prev_off);
	uint32_t d = p->diff[x.lane];
	/* every int8 of the band sits in the top byte of its lane while a block is worked on (see step()) */
	f.b.dh = (int)(d << 24); f.b.dv = (int)((d << 16) & 0xff000000u); f.b.de = (int)((d << 8) & 0xff000000u); f.b.df = (int)(d & 0xff000000u);
	f.b.delta = 0; f.b.drop = (int)((uint32_t)f.xd << 24);
	f.b.mh = f.b.mv = f.b.me = f.b.mf = 0;
	w.dmask = (uint32_t)rdfirst(0); w.dacc = rdfirst((int)p->s.acc);
	w.acnt = (uint32_t)rdfirst(0); w.bcnt = (uint32_t)rdfirst(0);
}

/* fill_fetch_core (gaba.c:1125-1144): the windows are already in registers; load the look-ahead */
__device__ __forceinline__ void fetch_look(Ctx &x, Work &w, FillState &f, uint32_t alen, uint32_t blen)
{
	int l = x.lane;
	bool isb = l >= 32;
	uint32_t k = (uint32_t)(l & 31);
	const Sec &s = w.sec[isb ? 1 : 0];
	uint32_t len = isb ? blen : alen;
	uint32_t pos = s.len - w.rem[isb ? 1 : 0] + k;                 /* tptr - rem + k */
	uint32_t c = 0;
	if(k < len) { c = fetch_code(x.ar, s, pos); c = isb ? enc_b(c) : c; }
	f.look = (int)c;
}

/* _fill_store_context (gaba.c:1734-1778) */
__device__ __forceinline__ void store_context(Ctx &x, Work &w, FillState &f, uint32_t blk_off, uint32_t cnt, bool trace = true)
{
	const Consts &c = x.c;
	Blk *p = blk_at(x, blk_off);
	int l = x.lane, W = w.W;
	Band &b = f.b;
	/* left-align the per-lane bit columns: vector k of the block -> bit (31 - k) */
	uint32_t sh = cnt == 0 ? 0 : (uint32_t)(BLK - cnt);
	bool act = l < W;
	if(trace) {              /* 1 KiB of the block; left unwritten when nobody will walk it */
		p->m[0][l] = act ? b.mh << sh : 0; p->m[1][l] = act ? b.mv << sh : 0;
		p->m[2][l] = act ? b.me << sh : 0; p->m[3][l] = act ? b.mf << sh : 0;
	}
	p->diff[l] = ((uint32_t)b.dh >> 24) | (((uint32_t)b.dv >> 16) & 0xff00u) | (((uint32_t)b.de >> 8) & 0xff0000u) | ((uint32_t)b.df & 0xff000000u);
	b.delta >>= 24; b.drop >>= 24;               /* back to plain sign-extended int8 */

	int drop_c = rdlane(b.drop, W / 2), cofs = rdlane(b.delta, W / 2);
	int xstat = (c.tx - drop_c) & TERM;
	int prev_drop = f.xd;
	bool upd = act && (sext8(b.drop + b.delta) > prev_drop);
	uint64_t max_mask = __ballot(upd);
	if(l == 0) {
		p->s.acc = (int8_t)w.dacc; p->s.xstat = (int8_t)xstat;
		p->s.acnt = (int8_t)w.acnt; p->s.bcnt = (int8_t)w.bcnt;
		p->s.dir_mask = w.dmask;
		p->s.max_mask = max_mask;
	}
	w.ofsd += cofs; w.rem[0] -= w.acnt; w.rem[1] -= w.bcnt;
	/* middle delta with overflow / underflow rescue (int16, gaba.c:1752-1761) */
	int md = f.md;
	md = sext16(md + b.delta);
	int ov = sext8(~sext8(b.drop + b.delta) & (b.drop & b.delta));
	md = sext16(md + (0x0100 & ov));
	int uv = sext8(min(127, max(-128, b.delta - 0x40)) | b.drop);
	md = sext16(md + (0x0100 & uv));
	md = sext16(md - (cofs + 0x0100));
	f.md = md; f.xd = b.drop;
}

/* one block of up to BLK vectors.  bounded = per-vector sequence-end tests (fill_cap_seq_bounded, gaba.c:1925-1975).
 * Returns the number of vectors filled. */
template<int MODEL, bool WIDE, bool FAST, bool bounded, bool TRACE>
__device__ __forceinline__ uint32_t fill_block_t(Ctx &x, Work &w, FillState &f, uint32_t prev_off, uint32_t blk_off, bool cont)
{
	const Consts &c = x.c;
	const StepK sk = step_consts(c, w.W);          /* (the look-ahead of the block is in f.look: fill_block fetched it) */
	if(cont) {
		/* the previous block was filled by this wave a moment ago and its diff vectors are still in the registers: what
		 * _fill_load_context (gaba.c:1527) would read back is what is here (acc went through its int8 slot) -- no HBM round trip */
		f.b.delta = 0; f.b.drop = (int)((uint32_t)f.xd << 24);
		f.b.mh = f.b.mv = f.b.me = f.b.mf = 0;
		w.dmask = 0; w.dacc = (int)(int8_t)w.dacc; w.acnt = 0; w.bcnt = 0;
	} else { load_context(x, w, f, prev_off); }
	const int64_t arem = (uint32_t)rdfirst((int)w.rem[0]), brem = (uint32_t)rdfirst((int)w.rem[1]), prem = (uint32_t)rdfirst((int)w.pridx);
	int dacc = rdfirst(w.dacc);
	const int ach_in = f.b.ach, look_in = f.look;
	int look = f.look;
	if(FAST) {
		/* windows and look-ahead switch to the encodings of the single-v_perm lookup */
		f.b.ach = fast_arow(c, ach_in); f.b.bch = fast_bsel(f.b.bch);
		look = x.lane < 32 ? fast_arow(c, look_in) : fast_bsel(look_in);
	}
	uint32_t bi = 32, dmask = 0;                                  /* bi = 32 + bcnt; acnt = k - bcnt */
	uint32_t k = 0;
	for(; k < BLK; k++) {
		const uint32_t down = (uint32_t)dacc >> 31;                  /* _dir_fetch, gaba.c:753 */
		const uint32_t ai = k + 32 - bi;
		if(bounded) {                                               /* _fill_cap_test_idx, gaba.c:1800-1809 */
			int64_t ta = arem - (int64_t)(ai + 1 - down), tb = brem - (int64_t)(bi - 32 + down);
			if((ta | tb | (ta + tb + prem)) < 0) { break; }
		}
		dmask = (dmask << 1) + down;
		const int t = step<MODEL, WIDE, FAST, TRACE>(sk, f.b, look, (int)down, (int)ai, (int)bi);
		bi += down;
		dacc += (rdlane(t, 0) >> 24) - (rdlane(t, sk.wm1) >> 24);   /* _dir_update, gaba.c:761 (t sits in the top byte) */
	}
	w.dacc = dacc; w.bcnt = bi - 32; w.acnt = k - w.bcnt; w.dmask = dmask;
	if(FAST) {
		/* back to base codes: the a window is the old one moved up by acnt lanes with the consumed look-ahead below it */
		const int acnt = (int)w.acnt, l = x.lane;
		const int from_old = __builtin_amdgcn_ds_bpermute(((l - acnt) & 63) << 2, ach_in);
		const int from_look = __builtin_amdgcn_ds_bpermute(((acnt - 1 - l) & 63) << 2, look_in);
		f.b.ach = l < acnt ? from_look : from_old;
		f.b.bch = fast_bdec(f.b.bch);
	}
	w.pridx -= k;
	x.n_vec += k; x.n_blk += 1;
	if(k != 0 && k != BLK) { w.dmask <<= (BLK - k); }              /* _dir_adjust_remainder, gaba.c:769 */
	store_context(x, w, f, blk_off, k, TRACE);
	return k;
}

template<bool bounded>
__device__ __forceinline__ uint32_t fill_block(Ctx &x, Work &w, FillState &f, uint32_t prev_off, uint32_t blk_off, bool cont)
{
	/* the gap model and the lookup form are fixed per context and the band width per fill: pick the specialised
	 * 32-vector loop once per block */
	/* the root windows hold two special codes (a = 0x0c, b = 0x03: gaba.c:3739 phantom block) that the row / selector
	 * encoding cannot express: blocks that still see them take the general lookup (the first one or two after a root) */
	uint32_t alen = BLK, blen = BLK;
	if(bounded) { alen = min(w.rem[0], (uint32_t)BLK); blen = min(w.rem[1], (uint32_t)BLK); }
	fetch_look(x, w, f, alen, blen);
	bool special = __ballot(x.lane < w.W && (f.b.ach > 4 || ((f.b.bch & 3) != 0 && f.b.bch != 2))) != 0;
	/* a score table whose entries against a b side N differ by a (Consts.fast_score == 2: the ONT presets' 4 x 4 matrices): the single-v_perm lookup has ONE score for
	 * that case, so a block that sees an N on the b side -- in the window, or among the 32 bases it may take in -- goes the general way; reads carry Ns rarely */
	if(x.c.fast_score == 2) { special = special || __ballot((x.lane < w.W && f.b.bch == 2) || (x.lane >= 32 && f.look == 2)) != 0; }
	const bool wide = w.W == 64, fast = x.c.fast_score != 0 && !special;
	#define GABA_PICK2(_m, _tr) \
		( wide ? (fast ? fill_block_t<_m, true, true, bounded, _tr>(x, w, f, prev_off, blk_off, cont) : fill_block_t<_m, true, false, bounded, _tr>(x, w, f, prev_off, blk_off, cont)) \
		       : (fast ? fill_block_t<_m, false, true, bounded, _tr>(x, w, f, prev_off, blk_off, cont) : fill_block_t<_m, false, false, bounded, _tr>(x, w, f, prev_off, blk_off, cont)) )
	#define GABA_PICK(_m) ( x.no_trace ? GABA_PICK2(_m, false) : GABA_PICK2(_m, true) )
	if(x.c.model == MODEL_COMBINED) { return GABA_PICK(MODEL_COMBINED); }
	return GABA_PICK(MODEL_AFFINE);
	#undef GABA_PICK2
	#undef GABA_PICK
}

/* ---- section / tail plumbing ---- */
/* fill_load_section (gaba.c:1269-1308); breakpoint masks are always zero in minialign's call pattern */
__device__ __forceinline__ void load_section(Ctx &x, Work &w, uint32_t tail_off, const Sec &a, const Sec &b, uint32_t pridx)
{
	const Tail *t = tail_at(x, tail_off);
	for(int k = 0; k < 2; k++) {
		const Sec &s = k ? b : a;
		uint32_t tr = (uint32_t)rdfirst((int)t->ridx[k]);
		uint32_t ridx = tr == 0 ? s.len : tr;
		w.rlim[k] = 0; w.id[k] = s.id; w.sec[k] = s;
		w.rem[k] = ridx; w.sridx[k] = ridx;
	}
	w.pridx = pridx; w.ofsd = 0; w.tail = tail_off;
}

/* fill_load_vectors (gaba.c:1376-1399) + fill_create_phantom (gaba.c:1315-1333) */
__device__ __forceinline__ void load_vectors(Ctx &x, Work &w, FillState &f, uint32_t tail_off)
{
	const Tail *t = tail_at(x, tail_off);
	int l = x.lane;
	w.W = rdfirst(t->W);
	int ch = t->ch[l];
	f.b.ach = ch & 0x0f; f.b.bch = (ch >> 4) & 0x0f;
	f.xd = t->xd[l]; f.md = t->md[l];
	uint32_t prev = (uint32_t)rdfirst((int)t->last);
	/* head ("phantom") block: copies diff / acc, marks HEAD, links to the previous block */
	uint32_t off = slab_alloc(x, sizeof(Blk));
	Blk *h = blk_at(x, off); const Blk *p = blk_at(x, prev);
	h->diff[l] = p->diff[l];
	if(l == 0) {
		h->s.acc = p->s.acc; h->s.xstat = (int8_t)((p->s.xstat & ROOT) | HEAD);
		h->s.acnt = 0; h->s.bcnt = 0; h->s.dir_mask = 0; h->s.max_mask = 0; h->s.link = prev;
	}
	w.blk0 = off; w.nblk = 1;
}

/* fill_init_fetch (gaba.c:1168-1210): slide the windows by the prefetched bases without computing vectors.
 * Returns bpos after the fetch. */
__device__ __forceinline__ int64_t init_fetch(Ctx &x, Work &w, FillState &f, int64_t apos, int64_t bpos)
{
	int32_t irem[2] = { (int32_t)(INIT_FETCH_POS - (int32_t)apos), (int32_t)(INIT_FETCH_POS - (int32_t)bpos) };
	int32_t srem[2] = { (int32_t)w.rem[0], (int32_t)w.rem[1] };
	int32_t len[2];
	len[0] = min(min(irem[0], srem[0]), (srem[1] - irem[1]) + (1 + irem[0]));
	len[1] = min(min(irem[1], srem[1]), (srem[0] - irem[0]) + (0 + irem[1]));
	fetch_look(x, w, f, (uint32_t)len[0], (uint32_t)len[1]);
	int W = w.W; bool lane_top = x.lane == W - 1;
	for(int k = 0; k < len[0]; k++) { f.b.ach = shift_up(f.b.ach, rdlane(f.look, k)); }
	for(int k = 0; k < len[1]; k++) {
		int nb = rdlane(f.look, 32 + k);
		int v = shift_dn(f.b.bch, nb);
		f.b.bch = (W != 64 && lane_top) ? nb : v;
	}
	Blk *h = blk_at(x, w.blk0);
	if(x.lane == 0) { h->s.acnt = (int8_t)len[0]; h->s.bcnt = (int8_t)len[1]; }
	w.rem[0] = (uint32_t)(srem[0] - len[0]); w.rem[1] = (uint32_t)(srem[1] - len[1]);
	/* the head block's (acnt, bcnt) are consumed by the first real block's fetch (gaba.c:1827): already applied here */
	return bpos + len[1];
}

/* fill_create_tail (gaba.c:1406-1499) */
__device__ __forceinline__ uint32_t create_tail(Ctx &x, Work &w, FillState &f, uint32_t last_blk, uint32_t last_cnt_nonzero, int xstat)
{
	uint32_t off = slab_alloc(x, sizeof(Tail));
	Tail *t = tail_at(x, off);
	const Tail *prev = tail_at(x, w.tail);
	int l = x.lane, W = w.W;
	/* fill_save_vectors: windows after the last consumed bases are exactly the register windows */
	t->ch[l] = (uint8_t)(f.b.ach | (f.b.bch << 4));
	t->xd[l] = (int8_t)f.xd; t->md[l] = (int16_t)f.md;
	int v = (l < W) ? sext16(f.md + f.xd) : -32768;
	for(int o = 32; o > 0; o >>= 1) { v = max(v, __shfl_xor(v, o)); }           /* _hmax_w */
	int mdrop = rdfirst(v);
	uint32_t ridx[2], adv[2];
	for(int k = 0; k < 2; k++) { ridx[k] = w.rem[k] + w.rlim[k]; adv[k] = w.sridx[k] - ridx[k]; }
	int64_t pmax = (int64_t)rdfirst64((uint64_t)prev->f.max); int pmdrop = rdfirst((int)prev->mdrop);
	if(l == 0) {
		t->mdrop = (int16_t)mdrop; t->istat = 0; t->pridx = w.pridx;
		t->ridx[0] = ridx[0]; t->ridx[1] = ridx[1]; t->adv[0] = adv[0]; t->adv[1] = adv[1];
		t->tail = w.tail; t->last = last_blk; t->W = W; t->_pad = 0;
		t->sec[0] = w.sec[0]; t->sec[1] = w.sec[1];
		t->f.aid = w.id[0]; t->f.bid = w.id[1];
		t->f.ascnt = prev->f.ascnt + (ridx[0] == 0); t->f.bscnt = prev->f.bscnt + (ridx[1] == 0);
		t->f.apos = prev->f.apos + (uint64_t)(int64_t)(int32_t)adv[0];
		t->f.bpos = prev->f.bpos + (uint64_t)(int64_t)(int32_t)adv[1];
		t->f.max = (pmax - pmdrop) + w.ofsd + mdrop;
		t->f.status = (((uint32_t)xstat & (TERM | CONT)) << 8) | (ridx[0] == 0 ? UPDATE_A : 0) | (ridx[1] == 0 ? UPDATE_B : 0);
		for(int k = 0; k < 5; k++) { t->f.reserved[k] = 0; }
	}
	(void)last_cnt_nonzero;
	return off;
}

/* fill_section_seq_bounded / fill_seq_bounded (gaba.c:2027-2099): bulk blocks while >= 32 bases remain on both
 * sides, then a per-vector bounded cap.  Returns the tail offset. */
__device__ __forceinline__ uint32_t fill_body(Ctx &x, Work &w, FillState &f, bool run_blocks)
{
	uint32_t last = w.blk0;                        /* last block written */
	bool cont = false;                             /* the diff vectors of `last` are still in registers */
	int xstat = (int)(int8_t)rdfirst((int)blk_at(x, w.blk0)->s.xstat);
	uint32_t head_cnt_nz = 0;
	if(!run_blocks) {
		/* still in the init-fetch state: the tail follows the head directly (gaba.c:2145-2147) */
		const Blk *h = blk_at(x, w.blk0);
		head_cnt_nz = (uint32_t)((rdfirst((int)h->s.acnt) | rdfirst((int)h->s.bcnt)) != 0);
		uint32_t lb = head_cnt_nz ? w.blk0 : (uint32_t)rdfirst((int)h->s.link);
		return create_tail(x, w, f, lb, head_cnt_nz, xstat);
	}
	/* bulk blocks while >= 32 bases remain on both sides (the reference's bulk / bounded-bulk variants only differ in where
	 * bounds are tested) ... */
	while((xstat & STAT_MASK) == CONT && !x.err) {
		bool can_bulk = w.rem[0] >= (uint32_t)BLK && w.rem[1] >= (uint32_t)BLK && w.pridx >= (uint32_t)BLK;
		if(!can_bulk) { break; }
		uint32_t off = slab_alloc(x, sizeof(Blk));
		if(x.err) { break; }
		fill_block<false>(x, w, f, last, off, cont);
		last = off; w.nblk++; cont = true;
		xstat = (int)(int8_t)((x.c.tx - rdlane(f.xd, w.W / 2)) & TERM);
	}
	/* ... then the per-vector bounded cap (fill_cap_seq_bounded, gaba.c:1925-1975) */
	if((xstat & STAT_MASK) == CONT) {
		while(xstat >= 0 && !x.err) {
			uint32_t off = slab_alloc(x, sizeof(Blk));
			if(x.err) { break; }
			uint32_t k = fill_block<true>(x, w, f, last, off, cont);
			cont = true;
			xstat = (int)(int8_t)((x.c.tx - rdlane(f.xd, w.W / 2)) & TERM);
			if(k != 0) { last = off; w.nblk++; } else { x.top = off; }   /* squash the empty block (gaba.c:1492) */
			if(k != BLK) { break; }
		}
	}
	return create_tail(x, w, f, last, 1, xstat);
}

/* gaba_dp_fill_root (gaba.c:2110-2154) when prev_tail == NIL, gaba_dp_fill (gaba.c:2161-2203) otherwise: one body so that
 * callers that chain fills keep a single inlined copy of the block loop */
__device__ __forceinline__ uint32_t dp_fill_any(Ctx &x, uint32_t prev_tail, int bw_idx, const Sec &a, uint32_t apos, const Sec &b, uint32_t bpos, uint32_t pridx)
{
	Work w; FillState f;
	const bool is_root = prev_tail == NIL;
	uint32_t src = prev_tail, vec_src = prev_tail;
	if(is_root) {
		uint32_t rt = root_tail(bw_idx);
		const Tail *root = tail_at(x, rt);
		/* fill_create_bridge (gaba.c:1339-1370) */
		uint32_t bo = slab_alloc(x, sizeof(Tail));
		Tail *br = tail_at(x, bo);
		int l = x.lane;
		br->ch[l] = root->ch[l]; br->xd[l] = root->xd[l]; br->md[l] = root->md[l];
		if(l == 0) {
			br->mdrop = root->mdrop; br->istat = (uint16_t)(root->istat | 1); br->pridx = root->pridx;
			br->ridx[0] = a.len - apos; br->ridx[1] = b.len - bpos; br->adv[0] = apos; br->adv[1] = bpos;
			br->tail = rt; br->last = NIL; br->W = root->W; br->_pad = 0;
			br->sec[0] = a; br->sec[1] = b;
			br->f = root->f; br->f.aid = a.id; br->f.bid = b.id;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		src = bo; vec_src = rt;
	}
	const Tail *t = tail_at(x, vec_src);
	uint32_t pr = pridx != 0 ? pridx : (is_root ? 0xffffffffu : (uint32_t)rdfirst((int)t->pridx));
	load_section(x, w, src, a, b, pr);
	load_vectors(x, w, f, vec_src);
	int64_t tap = (int64_t)rdfirst64(t->f.apos), tbp = (int64_t)rdfirst64(t->f.bpos);
	bool run = true;
	if(is_root || tbp < INIT_FETCH_POS) { run = !(init_fetch(x, w, f, tap, tbp) < INIT_FETCH_POS); }
	return fill_body(x, w, f, run);
}
__device__ __forceinline__ uint32_t dp_fill_root(Ctx &x, int bw_idx, const Sec &a, uint32_t apos, const Sec &b, uint32_t bpos, uint32_t pridx)
{
	return dp_fill_any(x, NIL, bw_idx, a, apos, b, bpos, pridx);
}
__device__ __forceinline__ uint32_t dp_fill(Ctx &x, uint32_t prev_tail, const Sec &a, const Sec &b, uint32_t pridx)
{
	return dp_fill_any(x, prev_tail, 0, a, 0, b, 0, pridx);
}

/* mm_extend_core (minialign.c:4075-4112): fill_root, then continue into the tail sections until X-drop or until a side that
 * was already updated asks again; returns the tail with the largest max.  The a-side tail of a circular reference is the
 * reference section itself (minialign.c:3753), else both are the run of N. */
__device__ __forceinline__ uint32_t extend_core(Ctx &x, int bw_idx, Sec ca, uint32_t apos, Sec cb, uint32_t bpos, const Sec &atail, const Sec &btail, int64_t &mmax_out, uint32_t &n_fill)
{
	uint32_t f = NIL, m = NIL, flag = STATUS_TERM; int64_t mmax = 0;
	while(true) {
		f = dp_fill_any(x, f, bw_idx, ca, apos, cb, bpos, 0); n_fill++;
		const Tail *t = tail_at(x, f);
		uint32_t st = (uint32_t)rdfirst((int)t->f.status);
		int64_t fm = (int64_t)rdfirst64((uint64_t)t->f.max);
		if(m == NIL || fm > mmax) { m = f; mmax = fm; }
		if((flag & st) != 0 || x.err) { break; }
		if(st & UPDATE_A) { ca = atail; }
		if(st & UPDATE_B) { cb = btail; }
		flag |= st & (UPDATE_A | UPDATE_B);
	}
	mmax_out = mmax;
	return m;
}

/* ---- max search (gaba.c:2604-2817) ---- */
struct Leaf {
	uint32_t blk;            /* block holding the max cell */
	uint32_t p, q;
	int32_t gidx[2], sgidx[2];
	uint32_t ofs[2], id[2];
	uint32_t tl[2];
	uint32_t state;
	uint32_t icnt[2], ecnt[2], fcnt[2];
};

/* previous block of `off` inside a fill is simply off - sizeof(Blk) (blocks of one fill are contiguous) */
__device__ __forceinline__ uint32_t skip_heads(Ctx &x, uint32_t off)
{
	while(off != NIL && (rdfirst((int)blk_at(x, off)->s.xstat) & HEAD)) { off = (uint32_t)rdfirst((int)blk_at(x, off)->s.link); }
	return off;
}

/* leaf_search (gaba.c:2708-2770): returns plen, fills lf.{blk,p,q,gidx,sgidx} */
__device__ __forceinline__ uint64_t leaf_search(Ctx &x, uint32_t tail_off, Leaf &lf)
{
	const Consts &c = x.c;
	const Tail *t = tail_at(x, tail_off);
	int l = x.lane, W = rdfirst(t->W);
	bool act = l < W;
	/* leaf_load_max_mask (gaba.c:2609-2631) */
	int mdrop = rdfirst((int)t->mdrop);
	uint64_t max_mask = __ballot(act && (sext16((int)t->md[l] + (int)t->xd[l]) == mdrop));
	int32_t ridx[2] = { rdfirst((int)t->ridx[0]), rdfirst((int)t->ridx[1]) };
	uint32_t b = (uint32_t)rdfirst((int)t->last) + (uint32_t)sizeof(Blk);
	while(true) {
		b -= (uint32_t)sizeof(Blk);
		int xs = rdfirst((int)blk_at(x, b)->s.xstat);
		if((xs & ROOT) == ROOT) { return 0; }
		b = skip_heads(x, b);
		const Blk *pb = blk_at(x, b);
		ridx[0] += rdfirst((int)pb->s.acnt); ridx[1] += rdfirst((int)pb->s.bcnt);
		uint64_t mm = rdfirst64(pb->s.max_mask);
		if((max_mask & ~mm) == 0) { break; }
		max_mask &= ~mm;
	}
	/* fill_restore_fetch (gaba.c:1217-1264): rebuild the sequence windows at the head of block b */
	const Tail *pt = tail_at(x, (uint32_t)rdfirst((int)t->tail));
	Work w; FillState f;
	w.W = W;
	int32_t ofs[2], len[2], cridx[2];
	for(int k = 0; k < 2; k++) {
		int32_t sridx = (int32_t)(rdfirst((int)t->ridx[k]) + rdfirst((int)t->adv[k]));
		int32_t dridx = ridx[k] + W;
		cridx[k] = min(dridx, sridx);
		ofs[k] = dridx - cridx[k];
		len[k] = min(cridx[k], W + BLK - ofs[k]);
		w.sec[k] = t->sec[k];
		w.sec[k].id = (uint32_t)rdfirst((int)w.sec[k].id); w.sec[k].len = (uint32_t)rdfirst((int)w.sec[k].len);
		w.sec[k].off = rdfirst64(w.sec[k].off); w.sec[k].arena = (uint32_t)rdfirst((int)w.sec[k].arena); w.sec[k].rev = (uint32_t)rdfirst((int)w.sec[k].rev);
	}
	{
		/* a window: lanes >= W - ofs come from the previous tail's window, the rest from the stream (t = W - ofs - 1 - lane) */
		int pch = pt->ch[(l - (W - ofs[0])) & 63] & 0x0f;
		int tt = W - ofs[0] - 1 - l;
		uint32_t base = w.sec[0].len - (uint32_t)cridx[0];
		int sv = (act && tt >= 0 && tt < len[0]) ? (int)fetch_code(x.ar, w.sec[0], base + (uint32_t)tt) : 0;
		f.b.ach = (l >= W - ofs[0]) ? pch : sv;
		/* b window: lanes < ofs come from the previous tail's top lanes, the rest from the stream (t = lane - ofs) */
		int pcb = (pt->ch[(W - ofs[1] + l) & 63] >> 4) & 0x0f;
		int tb = l - ofs[1];
		uint32_t bbase = w.sec[1].len - (uint32_t)cridx[1];
		int bv = (act && tb >= 0 && tb < len[1]) ? (int)enc_b(fetch_code(x.ar, w.sec[1], bbase + (uint32_t)tb)) : 0;
		f.b.bch = (l < ofs[1]) ? pcb : bv;
		/* look-ahead: a stream t = W - ofs + k, b stream t = W - ofs + k */
		bool isb = l >= 32; int k = l & 31;
		int ts = W - ofs[isb] + k;
		uint32_t sb_ = isb ? bbase : base;
		int lv = (ts < len[isb]) ? (int)fetch_code(x.ar, w.sec[isb], sb_ + (uint32_t)ts) : 0;
		f.look = isb ? (int)enc_b((uint32_t)lv) : lv;
		if(ts >= len[isb]) { f.look = 0; }
	}
	/* leaf_detect_pos (gaba.c:2663-2700): refill the block, recording cell-wise update masks */
	const Blk *pb = blk_at(x, b);
	int cnt = rdfirst((int)pb->s.acnt) + rdfirst((int)pb->s.bcnt);
	f.xd = 0; f.md = 0;
	load_context(x, w, f, b - (uint32_t)sizeof(Blk));
	int mx = f.b.delta;
	uint32_t upd = 0;          /* per-lane: bit (k) = updated at vector k */
	const StepK sk = step_consts(c, W);
	int dacc = rdfirst(w.dacc);
	for(int k = 0; k < cnt; k++) {
		const bool down = dacc < 0;
		const int sd = rdfirst((int)down), sa = rdfirst((int)w.acnt), sb_ = rdfirst(32 + (int)w.bcnt);
		const int tv = c.model == MODEL_COMBINED ? step<MODEL_COMBINED, false, false>(sk, f.b, f.look, sd, sa, sb_)
			: step<MODEL_AFFINE, false, false>(sk, f.b, f.look, sd, sa, sb_);
		w.acnt += down ? 0 : 1; w.bcnt += down ? 1 : 0;
		dacc += (rdlane(tv, 0) >> 24) - (rdlane(tv, sk.wm1) >> 24);
		upd |= (uint32_t)(act && f.b.delta > mx) << k;
		mx = max(mx, f.b.delta);
	}
	/* leaf_search_pos (gaba.c:2636-2652) */
	int mi = cnt;
	uint64_t mcur = 0;
	while(mi > 0) {
		mi--;
		mcur = __ballot((upd >> mi) & 1);
		if((max_mask & ~mcur) == 0) { break; }
		max_mask &= ~mcur;
	}
	uint64_t hit = mcur & max_mask;
	lf.p = (uint32_t)mi;
	lf.q = hit == 0 ? 64u : (uint32_t)__builtin_ctzll(hit);
	lf.blk = b;
	int32_t fcnt = (int32_t)lf.p + 1;
	uint32_t dir_mask = (uint32_t)rdfirst((int)pb->s.dir_mask) >> (BLK - fcnt);
	int32_t pc = __builtin_popcount(dir_mask);
	ridx[0] -= (fcnt - pc) - (1 + (int32_t)lf.q);
	ridx[1] -= pc - (W - (int32_t)lf.q);
	int32_t tr[2] = { rdfirst((int)t->ridx[0]), rdfirst((int)t->ridx[1]) };
	for(int k = 0; k < 2; k++) { lf.gidx[k] = 1 - ridx[k] + tr[k]; lf.sgidx[k] = lf.gidx[k]; }
	int32_t rem0 = ridx[0] - tr[0], rem1 = ridx[1] - tr[1];
	uint64_t plen = rdfirst64(t->f.apos) + rdfirst64(t->f.bpos) + 2ull + (uint64_t)W - (uint64_t)(int64_t)rem1 - (uint64_t)(int64_t)rem0;
	return plen;
}

/* gaba_dp_search_max (gaba.c:2776-2817), second half: convert the leaf's grid indices to section ids / positions */
__device__ __forceinline__ PosPair search_max_walk(Ctx &x, uint32_t tail_off, const Leaf &lf, uint64_t plen)
{
	PosPair pos;
	pos.plen = plen;
	const Tail *t = tail_at(x, tail_off);
	int32_t gidx[2] = { lf.gidx[0], lf.gidx[1] }, acc[2] = { 0, 0 };
	uint32_t id[2] = { (uint32_t)rdfirst((int)t->f.aid), (uint32_t)rdfirst((int)t->f.bid) };
	uint32_t cur = tail_off;
	while(true) {
		const Tail *ct = tail_at(x, cur);
		uint32_t prev = (uint32_t)rdfirst((int)ct->tail);
		if(prev == NIL) { break; }
		bool upd[2] = { 1 > gidx[0], 1 > gidx[1] };
		if(!upd[0] && !upd[1]) { break; }
		uint32_t nid[2] = { (uint32_t)rdfirst((int)ct->f.aid), (uint32_t)rdfirst((int)ct->f.bid) };
		acc[0] += rdfirst((int)ct->adv[0]); acc[1] += rdfirst((int)ct->adv[1]);
		cur = prev;
		const Tail *pt = tail_at(x, cur);
		for(int k = 0; k < 2; k++) {
			bool m = upd[k] && (rdfirst((int)pt->ridx[k]) == 0);
			if(m) { gidx[k] += acc[k]; id[k] = nid[k]; acc[k] = 0; }
		}
	}
	pos.aid = id[0]; pos.bid = id[1]; pos.apos = (uint32_t)gidx[0]; pos.bpos = (uint32_t)gidx[1];
	return pos;
}
__device__ __forceinline__ PosPair dp_search_max(Ctx &x, uint32_t tail_off, Leaf &lf)
{
	uint64_t plen = leaf_search(x, tail_off, lf);
	return search_max_walk(x, tail_off, lf, plen);
}

/* ---- traceback (gaba.c:2820-3407) ---- */
enum : uint32_t { TS_H = 1, TS_V = 2, TS_S = 4, ts_d = 3, ts_v0 = 2, ts_v1 = 6, ts_h0 = 1, ts_h1 = 5 };

struct Trace {
	int W, model;
	uint32_t blk; int32_t p; uint32_t q, save, dir_mask; bool bulk;
	int32_t gidx[2];
	uint64_t ppos;
	uint32_t *path;
	bool oob;
	/* this lane's mask words of the current block, and the 4 words of lane q (uniform) */
	uint32_t lm[4];
	uint32_t qh, qv, qe, qf; uint64_t qcur; uint32_t blk_loaded;      /* qcur: the q the four words were read for; ~0ull = none (q itself can be 0xffffffff: a walk that left the band at its lower edge) */
	uint32_t pw; uint64_t pw_idx;   /* path word being assembled (uniform) */
};

__device__ __forceinline__ void trace_load_block(Ctx &x, Trace &t)
{
	const Blk *b = blk_at(x, t.blk);
	int l = x.lane;
	t.lm[0] = b->m[0][l]; t.lm[1] = b->m[1][l]; t.lm[2] = b->m[2][l]; t.lm[3] = b->m[3][l];
	t.blk_loaded = t.blk; t.qcur = ~0ull;
}
/* (mask->x.all >> q) & 1 with the x86 shift-count masking of the reference's word size (gaba.c:2931-2951) */
__device__ __forceinline__ void trace_sel_q(Trace &t)
{
	if(t.qcur == (uint64_t)t.q) { return; }
	t.qcur = (uint64_t)t.q;
	uint32_t ql = (t.W == 64) ? (t.q & 63) : (t.q & 31);
	bool dead = ql >= (uint32_t)t.W;
	t.qh = dead ? 0 : (uint32_t)rdlane((int)t.lm[0], (int)ql); t.qv = dead ? 0 : (uint32_t)rdlane((int)t.lm[1], (int)ql);
	t.qe = dead ? 0 : (uint32_t)rdlane((int)t.lm[2], (int)ql); t.qf = dead ? 0 : (uint32_t)rdlane((int)t.lm[3], (int)ql);
}
#define GABA_BIT(_w, _p)   ( ((_w) >> (31 - (_p))) & 1u )
__device__ __forceinline__ bool t_diag_h(Trace &t) { trace_sel_q(t); return GABA_BIT(t.qh, t.p) == 0; }
__device__ __forceinline__ bool t_diag_v(Trace &t) { trace_sel_q(t); return GABA_BIT(t.qv, t.p) == 0; }
__device__ __forceinline__ bool t_gap_h(Trace &t) { trace_sel_q(t); return (t.model == MODEL_COMBINED ? GABA_BIT(~t.qh & t.qe, t.p) : GABA_BIT(t.qe, t.p)) == 0; }
__device__ __forceinline__ bool t_gap_v(Trace &t) { trace_sel_q(t); return (t.model == MODEL_COMBINED ? GABA_BIT(~t.qv & t.qf, t.p) : GABA_BIT(t.qf, t.p)) == 0; }
__device__ __forceinline__ bool t_fgap_h(Trace &t) { trace_sel_q(t); return t.model == MODEL_COMBINED ? GABA_BIT(t.qe, t.p) == 0 : false; }
__device__ __forceinline__ bool t_fgap_v(Trace &t) { trace_sel_q(t); return t.model == MODEL_COMBINED ? GABA_BIT(t.qf, t.p) == 0 : false; }

__device__ __forceinline__ uint32_t trace_head_cnt(int W) { return (uint32_t)(W / BLK + (W == 16)); }   /* gaba.c:3051 */

/* _trace_test_bulk (gaba.c:3035-3046) */
__device__ __forceinline__ bool trace_test_bulk(Ctx &x, Trace &t)
{
	const Blk *b = blk_at(x, t.blk);
	int32_t ga = t.gidx[0] - rdfirst((int)b->s.acnt), gb = t.gidx[1] - rdfirst((int)b->s.bcnt);
	if(!(t.W > ga) && !(t.W > gb)) { t.gidx[0] = ga; t.gidx[1] = gb; return true; }
	return false;
}
/* _trace_reload_block / _trace_reload_tail (gaba.c:3000-3031) */
__device__ __forceinline__ void trace_reload(Ctx &x, Trace &t)
{
	uint32_t b = skip_heads(x, t.blk - (uint32_t)sizeof(Blk));
	if(b == NIL) { t.blk = NIL; t.p = -1; t.dir_mask = 0; return; }
	const Blk *pb = blk_at(x, b);
	int cnt = rdfirst((int)pb->s.acnt) + rdfirst((int)pb->s.bcnt);
	t.p = cnt - 1; t.dir_mask = (uint32_t)rdfirst((int)pb->s.dir_mask) >> (BLK - cnt);
	t.blk = b;
	trace_load_block(x, t);
}
__device__ __forceinline__ void path_flush(Ctx &x, Trace &t)
{
	if(x.lane == 0) { t.path[t.pw_idx] = t.pw; }
}
/* _pop_vector (gaba.c:3114-3122) */
__device__ __forceinline__ bool trace_pop(Ctx &x, Trace &t, int is_v)
{
	if(!t.bulk) { t.gidx[is_v]--; }
	t.ppos--; x.n_tr++;
	if((t.ppos >> 5) != t.pw_idx) { path_flush(x, t); t.pw_idx = t.ppos >> 5; t.pw = 0; }
	t.pw |= (uint32_t)is_v << (t.ppos & 31);
	t.q += (t.dir_mask & 1) - (uint32_t)is_v;
	t.dir_mask >>= 1;
	t.p--;
	if(t.p >= 0) { return false; }
	if(t.bulk) {
		trace_reload(x, t);
		if(!trace_test_bulk(x, t)) {
			if(t.q >= (uint32_t)t.W) { t.oob = true; return true; }
			t.gidx[1] += (int32_t)(t.q - t.save);
			t.gidx[0] += (int32_t)(t.save - t.q);
			t.save = trace_head_cnt(t.W);
			t.bulk = false;
		}
	} else {
		bool prev_head = (rdfirst((int)blk_at(x, t.blk - (uint32_t)sizeof(Blk))->s.xstat) & HEAD) != 0;
		trace_reload(x, t);
		if(!prev_head) {
			t.save--;
			if(t.save >= trace_head_cnt(t.W) && trace_test_bulk(x, t)) { t.save = t.q; t.bulk = true; }
		}
	}
	return false;
}

/*
 * trace_core (gaba.c:3111-3228).  Everything but the four mask columns is wave-uniform: state lives in locals, the mask
 * words of lane q are re-read (4 x v_readlane) only when q moves, and the block reload is one shared site at the top of the
 * dispatch loop.  Same entry points (labels), same tail / bulk mode switches, same exits as the reference.
 */
__device__ __forceinline__ void trace_core(Ctx &x, Trace &t, Leaf &lf)
{
	enum { L_D_HEAD, L_D_MID, L_D_TAIL, L_H_HEAD, L_H_LOOP, L_H_TAIL, L_V_HEAD, L_V_LOOP, L_V_TAIL };
	const int W = t.W; const bool comb = t.model == MODEL_COMBINED;
	const uint32_t head_cnt = trace_head_cnt(W);
	uint32_t blk = lf.blk; int32_t p = (int32_t)lf.p; uint32_t q = lf.q, save = head_cnt;
	bool bulk = false; t.oob = false;
	uint32_t dir = (uint32_t)rdfirst((int)blk_at(x, blk)->s.dir_mask) >> (BLK - (p + 1));
	int32_t g0 = lf.gidx[0], g1 = lf.gidx[1];
	uint32_t ppos = (uint32_t)t.ppos, pw = t.pw;
	uint32_t icnt0 = lf.icnt[0], icnt1 = lf.icnt[1], ecnt0 = lf.ecnt[0], ecnt1 = lf.ecnt[1], fcnt0 = lf.fcnt[0], fcnt1 = lf.fcnt[1];
	uint32_t state = lf.state, n_pop = 0;
	int lbl;
	switch(state) {
		case ts_d:  lbl = L_D_HEAD; break;
		case ts_v0: lbl = L_V_HEAD; break;
		case ts_v1: lbl = L_V_TAIL; break;
		case ts_h0: lbl = L_H_HEAD; break;
		case ts_h1: lbl = L_H_TAIL; break;
		default: return;
	}
	if(t.blk_loaded != blk) { t.blk = blk; trace_load_block(x, t); }
	uint64_t qsel = ~0ull; uint32_t qh = 0, qv = 0, qe = 0, qf = 0;    /* qsel: the q the cached mask words belong to; ~0ull = none (q can be 0xffffffff when the walk has left the band) */
	TPROF_DECL();
	/* one-block look-behind: mask columns and trailer of the block at pf_off (loads are issued here, waited for at first use) */
	uint32_t pf_off = NIL, pf_m0 = 0, pf_m1 = 0, pf_m2 = 0, pf_m3 = 0; uint4 pf_s = make_uint4(0, 0, 0, 0);
	#define TR_PREFETCH(_off) { const Blk *pb_ = blk_at(x, (_off)); pf_m0 = pb_->m[0][x.lane]; pf_m1 = pb_->m[1][x.lane]; pf_m2 = pb_->m[2][x.lane]; pf_m3 = pb_->m[3][x.lane]; \
		pf_s = *(const uint4 *)&pb_->s; pf_off = (_off); }
	if(blk >= 2 * (uint32_t)sizeof(Blk)) { TR_PREFETCH(blk - (uint32_t)sizeof(Blk)); }

	#define TR_BIT(_w)      ( ((_w) >> (31 - p)) & 1u )
	/* _pop_vector (gaba.c:3114-3122) without the reload */
	#define TR_POP(_is_v) { \
		if(!bulk) { if(_is_v) { g1--; } else { g0--; } } \
		if((ppos & 31) == 0) { if(x.lane == 0) { t.path[ppos >> 5] = pw; } pw = 0; } \
		ppos--; n_pop++; \
		if(_is_v) { pw |= 1u << (ppos & 31); } \
		q += (dir & 1) - (uint32_t)(_is_v); dir >>= 1; p--; }

	while(true) {
		if(p < 0) {
			TPROF_T0();
			/* _trace_{bulk,tail}_load_n (gaba.c:3052-3089): step to the previous block, hopping over head blocks.  The block
			 * below the current one was requested when the current one was entered (mask columns + the 16-B trailer in one
			 * go), so the walk normally finds it in registers instead of paying four dependent HBM round trips here. */
			const uint32_t cand = blk - (uint32_t)sizeof(Blk);
			if(pf_off != cand) { TR_PREFETCH(cand); }
			uint32_t w0 = (uint32_t)rdfirst((int)pf_s.x), dm = (uint32_t)rdfirst((int)pf_s.y);
			const bool prev_head = (((int)(int8_t)(w0 >> 8)) & HEAD) != 0;
			uint32_t nb = cand;
			if(prev_head) {
				nb = skip_heads(x, cand);
				if(nb == NIL) { blk = NIL; break; }
				TR_PREFETCH(nb);
				w0 = (uint32_t)rdfirst((int)pf_s.x); dm = (uint32_t)rdfirst((int)pf_s.y);
			}
			const int ac = (int)(int8_t)(w0 >> 16), bc = (int)(int8_t)(w0 >> 24);
			p = ac + bc - 1; dir = dm >> (BLK - (ac + bc));
			blk = nb; t.blk = nb; t.lm[0] = pf_m0; t.lm[1] = pf_m1; t.lm[2] = pf_m2; t.lm[3] = pf_m3; t.blk_loaded = nb; qsel = ~0ull;
			if(nb >= 2 * (uint32_t)sizeof(Blk)) { TR_PREFETCH(nb - (uint32_t)sizeof(Blk)); }
			bool can_bulk = !(W > g0 - ac) && !(W > g1 - bc);             /* _trace_test_bulk (gaba.c:3035-3046) */
			if(bulk) {
				if(can_bulk) { g0 -= ac; g1 -= bc; }
				else { if(q >= (uint32_t)W) { t.oob = true; break; } g1 += (int32_t)(q - save); g0 += (int32_t)(save - q); save = head_cnt; bulk = false; }
			} else if(!prev_head) {
				save--;
				if(save >= head_cnt && can_bulk) { g0 -= ac; g1 -= bc; save = q; bulk = true; }
			}
			TPROF_ADD(0);
		}
		/*
		 * diagonal run, batched.  The cells a run of diagonals visits are fixed by the direction bits alone: the k-th one
		 * is (p - 2k, q + popcount(dir[0, 2k)) - k).  Lane k gathers the h / v mask words of its cell (ds_bpermute) and
		 * two ballots give the first cell where the reference's loop (gaba.c:3133-3160) would leave the diagonal: a set
		 * v bit (tested first, _trace_test_diag_v at the tail of the previous diagonal) or a set h bit.  The run is
		 * capped to the vectors of this block and, in tail mode, to the section indices; whatever stops it is handled
		 * by the step-wise code below.
		 */
		if((lbl == L_D_HEAD || lbl == L_D_TAIL) && p >= 0) {
			/* entered either in front of the h test (head) or in front of the v test of the same cell (tail) */
			const bool from_tail = lbl == L_D_TAIL;
			uint32_t nmax = (uint32_t)(p + 1) >> 1;                        /* whole diagonals left in this block */
			if(!bulk) { nmax = min(nmax, (uint32_t)max(0, min(g0, g1))); }
			TPROF_T0();
			const uint32_t k = (uint32_t)x.lane;
			const uint32_t ktest = min(nmax, (uint32_t)p >> 1);             /* cells 0..ktest lie in the block and are reached */
			const uint32_t qk = q + (uint32_t)__popc(dir & ((1u << ((2 * k) & 31)) - 1u)) - k;
			const uint32_t qlk = (W == 64) ? (qk & 63) : (qk & 31);
			const uint32_t hw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(qlk << 2), (int)t.lm[0]);
			const uint32_t vw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(qlk << 2), (int)t.lm[1]);
			const uint32_t sh = ((uint32_t)(31 - p) + 2 * k) & 31;         /* 31 - (p - 2k) */
			const bool live = k <= ktest && qlk < (uint32_t)W;
			const uint64_t eh = __ballot(live && ((hw >> sh) & 1u));
			const uint64_t ev = __ballot(live && (from_tail || k >= 1) && ((vw >> sh) & 1u));
			const uint32_t e_h = eh ? (uint32_t)__builtin_ctzll(eh) : 64u, e_v = ev ? (uint32_t)__builtin_ctzll(ev) : 64u;
			const uint32_t n = min(min(e_h, e_v), nmax);
			const bool event = (e_h & e_v) != 64u;
			if(event) { lbl = (e_v < 64u && e_v <= e_h) ? L_V_HEAD : L_H_HEAD; }
			else { lbl = (p - (int32_t)(2 * n) >= 0) ? L_D_HEAD : L_D_TAIL; }
			if(n > 0) {
				/* 2n pops: path bits 0,1,0,1,... (gaba.c:3114-3122), i.e. every position of ppos's parity below ppos */
				uint32_t rem = 2 * n; const uint32_t par = (ppos & 1) ? 0xaaaaaaaau : 0x55555555u;
				while(rem) {
					if((ppos & 31) == 0) { if(x.lane == 0) { t.path[ppos >> 5] = pw; } pw = 0; }
					const uint32_t inw = ((ppos - 1) & 31) + 1, take = min(rem, inw), lo = (ppos - take) & 31;
					pw |= ((take == 32 ? 0xffffffffu : ((1u << take) - 1u)) << lo) & par;
					ppos -= take; rem -= take;
				}
				q += (uint32_t)__popc(dir & (n == 16 ? 0xffffffffu : ((1u << (2 * n)) - 1u))) - n;
				dir = n == 16 ? 0u : dir >> (2 * n);
				p -= (int32_t)(2 * n); n_pop += 2 * n;
				if(!bulk) { g0 -= (int32_t)n; g1 -= (int32_t)n; }
				qsel = ~0ull;
			}
			TPROF_ADD(1);
			/* nothing moved and nothing found: the cell in front is clean (h and, from the tail, v are both clear) but no whole
			 * diagonal fits; the step-wise code takes the half step (block boundary) or stops on the section test */
			if(event || n > 0) { continue; }
			lbl = L_D_HEAD;
		}
		/*
		 * gap run, batched the same way.  Entered at the head of a gap (gaba.c:3163-3228 _trace_*_{h,v}_head): the cells a run
		 * of horizontal (vertical) moves visits are again fixed by the direction bits, the j-th one is (p - j, q + popcount(dir[0, j))
		 * [- j]).  Lane j gathers the two mask words its tests need; cell 0 decides between the one-step "fgap" form of the
		 * COMBINED model and an ordinary gap, whose length is the first j >= 1 where the gap-continue test fails.  Capped to
		 * the block and, in tail mode, to the section index; what is left over goes to the step-wise code.
		 */
		if((lbl == L_H_HEAD || lbl == L_V_HEAD) && p >= 0) {
			const bool isv = lbl == L_V_HEAD;
			uint32_t nmax = (uint32_t)(p + 1);
			if(!bulk) { nmax = min(nmax, (uint32_t)max(0, isv ? g1 : g0)); }
			if(nmax > 0) {
				TPROF_T0();
				const uint32_t j = (uint32_t)x.lane;
				const uint32_t lowj = j >= 32 ? 0xffffffffu : ((1u << j) - 1u);
				const uint32_t qj = q + (uint32_t)__popc(dir & lowj) - (isv ? j : 0u);
				const uint32_t qlj = (W == 64) ? (qj & 63) : (qj & 31);
				const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(qlj << 2), (int)(isv ? t.lm[1] : t.lm[0]));   /* h / v */
				const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(qlj << 2), (int)(isv ? t.lm[3] : t.lm[2]));   /* e / f */
				const bool live = j <= (uint32_t)p && qlj < (uint32_t)W;
				const uint32_t sh = ((uint32_t)(31 - p) + j) & 31;
				const uint32_t b0 = live ? (w0 >> sh) & 1u : 0u, b1 = live ? (w1 >> sh) & 1u : 0u;
				const uint64_t m_b1 = __ballot(b1 != 0);
				const uint64_t m_stop = __ballot(j >= 1 && j <= (uint32_t)p && (comb ? (~b0 & b1 & 1u) : b1) != 0);
				uint32_t m;
				if(comb && (m_b1 & 1ull) == 0) {                            /* _trace_test_fgap_{h,v}: a single step, back to the diagonal */
					if(isv) { fcnt1++; } else { fcnt0++; }
					m = 1; lbl = isv ? L_D_TAIL : L_D_HEAD;
				} else {
					if(isv) { icnt1++; } else { icnt0++; }
					const uint32_t first = m_stop ? (uint32_t)__builtin_ctzll(m_stop) : 64u;
					m = min(first, nmax);
					if(isv) { ecnt1 += m; } else { ecnt0 += m; }
					if(first <= nmax) { lbl = isv ? L_D_TAIL : L_D_HEAD; }
					else { lbl = (p - (int32_t)m >= 0) ? (isv ? L_V_LOOP : L_H_LOOP) : (isv ? L_V_TAIL : L_H_TAIL); }
				}
				/* m pops of one kind: path bits all 0 (h) or all 1 (v) */
				uint32_t rem = m;
				while(rem) {
					if((ppos & 31) == 0) { if(x.lane == 0) { t.path[ppos >> 5] = pw; } pw = 0; }
					const uint32_t inw = ((ppos - 1) & 31) + 1, take = min(rem, inw), lo = (ppos - take) & 31;
					if(isv) { pw |= (take == 32 ? 0xffffffffu : ((1u << take) - 1u)) << lo; }
					ppos -= take; rem -= take;
				}
				q += (uint32_t)__popc(dir & (m >= 32 ? 0xffffffffu : ((1u << m) - 1u))) - (isv ? m : 0u);
				dir = m >= 32 ? 0u : dir >> m;
				p -= (int32_t)m; n_pop += m;
				if(!bulk) { if(isv) { g1 -= (int32_t)m; } else { g0 -= (int32_t)m; } }
				qsel = ~0ull;
				TPROF_ADD(2);
				continue;
			}
		}
		TPROF_T0();
		if(qsel != (uint64_t)q) {
			/* (mask >> q) & 1 with the x86 shift-count masking of the reference's word size (gaba.c:2931-2951) */
			qsel = (uint64_t)q; uint32_t ql = (W == 64) ? (q & 63) : (q & 31); bool dead = ql >= (uint32_t)W;
			qh = dead ? 0u : (uint32_t)rdlane((int)t.lm[0], (int)ql); qv = dead ? 0u : (uint32_t)rdlane((int)t.lm[1], (int)ql);
			qe = dead ? 0u : (uint32_t)rdlane((int)t.lm[2], (int)ql); qf = dead ? 0u : (uint32_t)rdlane((int)t.lm[3], (int)ql);
		}
		if(lbl == L_D_HEAD) {
			if(TR_BIT(qh)) { lbl = L_H_HEAD; continue; }
			if(!bulk && (g0 == 0 || g1 == 0)) { state = ts_d; break; }
			TR_POP(0);
			lbl = L_D_MID;
			if(p < 0) { continue; }
			/* fast path: the second half of the diagonal needs no mask word (q may move, the tests come after) */
			TR_POP(1);
			lbl = L_D_TAIL;
		} else if(lbl == L_D_MID) {
			TR_POP(1);
			lbl = L_D_TAIL;
		} else if(lbl == L_D_TAIL) {
			lbl = TR_BIT(qv) ? L_V_HEAD : L_D_HEAD;
		} else if(lbl == L_H_HEAD) {
			if(comb && TR_BIT(qe) == 0) {                               /* _trace_test_fgap_h */
				if(!bulk && g0 == 0) { state = ts_h0; break; }
				fcnt0++;
				TR_POP(0);
				lbl = L_D_HEAD;
			} else { icnt0++; lbl = L_H_LOOP; }
		} else if(lbl == L_H_LOOP) {
			if(!bulk && g0 == 0) { state = ts_h1; break; }
			ecnt0++;
			TR_POP(0);
			lbl = L_H_TAIL;
		} else if(lbl == L_H_TAIL) {
			lbl = ((comb ? TR_BIT(~qh & qe) : TR_BIT(qe)) == 0) ? L_H_LOOP : L_D_HEAD;     /* _trace_test_gap_h */
		} else if(lbl == L_V_HEAD) {
			if(comb && TR_BIT(qf) == 0) {
				if(!bulk && g1 == 0) { state = ts_v0; break; }
				fcnt1++;
				TR_POP(1);
				lbl = L_D_TAIL;
			} else { icnt1++; lbl = L_V_LOOP; }
		} else if(lbl == L_V_LOOP) {
			if(!bulk && g1 == 0) { state = ts_v1; break; }
			ecnt1++;
			TR_POP(1);
			lbl = L_V_TAIL;
		} else { /* L_V_TAIL */
			lbl = ((comb ? TR_BIT(~qv & qf) : TR_BIT(qf)) == 0) ? L_V_LOOP : L_D_TAIL;
		}
		TPROF_ADD(3);
	}
	#undef TR_BIT
	#undef TR_POP
	#undef TR_PREFETCH
	TPROF_FLUSH();
	lf.state = state;
	lf.blk = blk; lf.p = (uint32_t)p; lf.q = q;
	lf.gidx[0] = g0; lf.gidx[1] = g1;
	lf.icnt[0] = icnt0; lf.icnt[1] = icnt1; lf.ecnt[0] = ecnt0; lf.ecnt[1] = ecnt1; lf.fcnt[0] = fcnt0; lf.fcnt[1] = fcnt1;
	t.blk = blk; t.ppos = ppos; t.pw = pw; t.pw_idx = ppos >> 5;
	x.n_tr += n_pop;
}

/* trace_reload_section (gaba.c:2826-2860) */
__device__ __forceinline__ void trace_reload_section(Ctx &x, Leaf &lf, int i)
{
	uint32_t tail = lf.tl[i], prev_tail = tail;
	int32_t gidx = lf.gidx[i];
	while(gidx <= 0) {
		do {
			const Tail *t = tail_at(x, tail);
			gidx += rdfirst((int)t->istat) ? 0 : rdfirst((int)t->adv[i]);
			prev_tail = tail; tail = (uint32_t)rdfirst((int)t->tail);
		} while(rdfirst((int)tail_at(x, tail)->ridx[i]) != 0);
	}
	const Tail *pt = tail_at(x, prev_tail);
	lf.tl[i] = tail;
	lf.id[i] = (uint32_t)rdfirst((int)(i == 0 ? pt->f.aid : pt->f.bid));
	lf.ofs[i] = rdfirst((int)pt->istat) ? (uint32_t)rdfirst((int)pt->adv[i]) : 0u;
	lf.gidx[i] = gidx; lf.sgidx[i] = gidx;
}

struct AlnOut {              /* what gaba_alignment_s carries (gaba.h:205-220) */
	int64_t score; double identity;
	uint32_t agcnt, bgcnt, dcnt, slen, plen;
	int32_t status;          /* 1: ok, -1: path left the band (reference returns NULL, gaba.c:3324) */
};

/*
 * gaba_dp_trace (gaba.c:3372) -> trace_body (gaba.c:3299).  path: (plen + 31) / 32 + 2 words (zeroed here);
 * seg: written in root-first order like aln->seg[]; max_seg bounds the array.
 */
/* phase 1: locate the max cell, return the path length (0 when the fill never left the init-fetch state) */
__device__ __forceinline__ uint64_t dp_trace_begin(Ctx &x, uint32_t tail_off, Leaf &lf)
{
	const Tail *tail = tail_at(x, tail_off);
	int64_t fbpos = (int64_t)rdfirst64(tail->f.bpos);
	return fbpos < INIT_FETCH_POS ? 0 : leaf_search(x, tail_off, lf);
}
/* phase 2: walk back; path must hold (plen + 31) / 32 + 2 words */
__device__ __forceinline__ AlnOut dp_trace_finish(Ctx &x, uint32_t tail_off, Leaf &lf, uint64_t plen, uint32_t *path, Segment *seg, uint32_t max_seg)
{
	const Consts &c = x.c;
	const Tail *tail = tail_at(x, tail_off);
	AlnOut out; out.status = 1;
	uint64_t pn = (plen + 31) / 32 + 2;
	for(uint64_t i = (uint64_t)x.lane; i < pn; i += 64) { path[i] = 0; }
	lf.tl[0] = tail_off; lf.tl[1] = tail_off;
	lf.icnt[0] = lf.icnt[1] = lf.ecnt[0] = lf.ecnt[1] = lf.fcnt[0] = lf.fcnt[1] = 0;
	lf.state = ts_d;
	Trace t;
	t.W = rdfirst(tail->W); t.model = c.model; t.path = path; t.ppos = plen;
	t.blk_loaded = NIL; t.pw_idx = plen >> 5; t.pw = 1u << (plen & 31);            /* sentinel bit (gaba.c:3288) */
	t.blk = NIL; t.p = 0; t.q = 0; t.save = 0; t.dir_mask = 0; t.bulk = false; t.oob = false; t.qcur = ~0ull;
	uint32_t slen = 0;
	while(t.ppos > 0) {
		if(lf.gidx[0] < (int32_t)((lf.state & TS_H) != 0)) { trace_reload_section(x, lf, 0); }
		if(lf.gidx[1] < (int32_t)((lf.state & TS_V) != 0)) { trace_reload_section(x, lf, 1); }
		trace_core(x, t, lf);
		if(lf.q >= (uint32_t)t.W || t.blk == NIL) { out.status = -1; out.plen = (uint32_t)plen; return out; }
		/* trace_push_segment (gaba.c:2865-2897); reversed into root-first order below */
		if(slen < max_seg && x.lane == 0) {
			Segment *s = &seg[slen];
			s->aid = lf.id[0]; s->bid = lf.id[1];
			s->apos = lf.ofs[0] + (uint32_t)lf.gidx[0]; s->bpos = lf.ofs[1] + (uint32_t)lf.gidx[1];
			s->alen = (uint32_t)(lf.sgidx[0] - lf.gidx[0]); s->blen = (uint32_t)(lf.sgidx[1] - lf.gidx[1]);
			s->ppos = t.ppos;
		}
		slen++;
		lf.sgidx[0] = lf.gidx[0]; lf.sgidx[1] = lf.gidx[1];
	}
	if(x.lane == 0) { t.path[t.ppos >> 5] = t.pw; }
	if(slen > max_seg) { x.err = 3; }
	/* reverse the segment array (the reference pushes with seg--) */
	if(x.lane == 0) {
		uint32_t n = min(slen, max_seg);
		for(uint32_t i = 0; i < n / 2; i++) { Segment tmp = seg[i]; seg[i] = seg[n - 1 - i]; seg[n - 1 - i] = tmp; }
	}
	/* identity estimate (gaba.c:3334-3355); _mul_v2i32 = _mm_mul_epi32 multiplies lane 0 only (v2i32.h:106) */
	int32_t gcnt0 = (int32_t)(lf.ecnt[0] + lf.fcnt[0]), gcnt1 = (int32_t)(lf.ecnt[1] + lf.fcnt[1]);
	int64_t p1 = (int64_t)c.gi * (int64_t)(int32_t)lf.icnt[0], p2 = (int64_t)c.ge * (int64_t)(int32_t)lf.ecnt[0], p3 = (int64_t)c.gfa * (int64_t)(int32_t)lf.fcnt[0];
	int32_t g0 = (int32_t)((uint32_t)p1 + (uint32_t)p2 + (uint32_t)p3);
	int32_t g1 = (int32_t)((uint32_t)(p1 >> 32) + (uint32_t)(p2 >> 32) + (uint32_t)(p3 >> 32));
	uint64_t dlen = (plen - (uint64_t)(int64_t)gcnt1 - (uint64_t)(int64_t)gcnt0) >> 1;
	int64_t score = (int64_t)rdfirst64((uint64_t)tail->f.max);
	int64_t dsc = score + g1 + g0;
	out.score = score;
	out.identity = dlen == 0 ? 0.0 : (((double)dsc / (double)dlen) * c.imx - c.xmx);
	out.agcnt = (uint32_t)gcnt0; out.bgcnt = (uint32_t)gcnt1; out.dcnt = (uint32_t)dlen;
	out.slen = slen; out.plen = (uint32_t)plen;
	return out;
}

__device__ __forceinline__ AlnOut dp_trace(Ctx &x, uint32_t tail_off, uint32_t *path, uint64_t path_cap_words, Segment *seg, uint32_t max_seg)
{
	Leaf lf;
	uint64_t plen = dp_trace_begin(x, tail_off, lf);
	if((plen + 31) / 32 + 2 > path_cap_words) { x.err = 2; AlnOut o; o.status = -2; o.plen = (uint32_t)plen; return o; }
	return dp_trace_finish(x, tail_off, lf, plen, path, seg, max_seg);
}

/* gaba_dp_flush (gaba.c:3969): reset the bump pointer behind the root blocks */
__device__ __forceinline__ void dp_flush(Ctx &x) { x.top = SLAB_HEAD; }

} /* namespace gaba */
