/*
 * ora_mm.c -- TEST INFRASTRUCTURE (see ora_mm.h).  Plain-C restatement of the reference mapper
 * (/root/reference/minialign.c 0.6.0-devel) from FASTA in to SAM out, single-threaded:
 *   sketch (minialign.c:2349-2448)       index build / get (:2656-3040)     seed (:3420-3540)
 *   chain (:3547-3725)                   extension driver (:3785-4173)      post-map (:4185-4398)
 *   SAM (:5096-5426)                     ksort radix sorts (ksort.h:84-131) kh_t (:341-683)
 * Quirks of the reference that are part of its observable output are reproduced and marked "QUIRK".
 *   circular references (:2438-2444, 3632-3696, 3753)   optional SAM fields and the MAF / BLAST6 / PAF printers (:5204-5625)
 * Not restated: all-versus-all, BAM / gz input, the pthread pipeline.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <time.h>
#include "ora_mm.h"

#define MAX2(x, y)  ( (x) > (y) ? (x) : (y) )
#define MIN2(x, y)  ( (x) < (y) ? (x) : (y) )
typedef union { uint64_t u64[2]; uint32_t u32[4]; } v4u32_t;
typedef union { uint64_t u64[1]; uint32_t u32[2]; } v2u32_t;

/* double -> uint32 the way gcc/x86-64 does it (cvttsd2si r64, then truncate); out-of-range gives 0x8000000000000000 -> 0 */
static inline uint32_t d2u32(double d) { if(!(d > -9.2e18 && d < 9.2e18)) { return 0; } return (uint32_t)(int64_t)d; }
static inline uint32_t f2u32(float f) { if(!(f > -9.2e18f && f < 9.2e18f)) { return 0; } return (uint32_t)(int64_t)f; }

/* ---- growable arrays ---- */
#define vec_t(type)     struct { uint64_t n, m; type *a; }
#define vec_reserve(type, v, s) { if((v).m < (uint64_t)(s)) { (v).m = MAX2(256, (uint64_t)(s) * 2); (v).a = (type *)realloc((v).a, sizeof(type) * (v).m); } }
#define vec_push(type, v, x)    { vec_reserve(type, v, (v).n + 1); (v).a[(v).n++] = (x); }

/* ---- ksort.h:84-131: in-place MSD radix sort + insertion sort, UNSTABLE; the exact permutation is part of the contract ---- */
#define RS_MIN_SIZE 64
#define RADIX_SORT(name, type_t, keyexpr, keybytes) \
	typedef struct { type_t *b, *e; } rsb_##name##_t; \
	static void rs_ins_##name(type_t *beg, type_t *end) { \
		for(type_t *i = beg + 1; i < end; ++i) { \
			if(keyexpr(*i) < keyexpr(*(i - 1))) { \
				type_t *j, tmp = *i; \
				for(j = i; j > beg && keyexpr(tmp) < keyexpr(*(j - 1)); --j) { *j = *(j - 1); } \
				*j = tmp; \
			} \
		} \
	} \
	static void rs_sort_##name(type_t *beg, type_t *end, int n_bits, int s) { \
		int size = 1 << n_bits, m = size - 1; \
		rsb_##name##_t b[256], *be = b + size, *k; \
		for(k = b; k != be; ++k) { k->b = k->e = beg; } \
		for(type_t *i = beg; i != end; ++i) { ++b[keyexpr(*i) >> s & m].e; } \
		for(k = b + 1; k != be; ++k) { k->e += (k - 1)->e - beg; k->b = (k - 1)->e; } \
		for(k = b; k != be;) { \
			if(k->b != k->e) { \
				rsb_##name##_t *l; \
				if((l = b + (keyexpr(*k->b) >> s & m)) != k) { \
					type_t tmp = *k->b, swap; \
					do { swap = tmp; tmp = *l->b; *l->b++ = swap; l = b + (keyexpr(tmp) >> s & m); } while(l != k); \
					*k->b++ = tmp; \
				} else { ++k->b; } \
			} else { ++k; } \
		} \
		for(b->b = beg, k = b + 1; k != be; ++k) { k->b = (k - 1)->e; } \
		if(s) { \
			s = s > n_bits ? s - n_bits : 0; \
			for(k = b; k != be; ++k) { \
				if(k->e - k->b > RS_MIN_SIZE) { rs_sort_##name(k->b, k->e, n_bits, s); } \
				else if(k->e - k->b > 1) { rs_ins_##name(k->b, k->e); } \
			} \
		} \
	} \
	static void radix_sort_##name(type_t *p, uint64_t l) { \
		if(l <= RS_MIN_SIZE) { rs_ins_##name(p, p + l); } else { rs_sort_##name(p, p + l, 8, keybytes * 8 - 8); } \
	}
#define KEY128(a)   ( (a).u64[0] )
#define KEY64(a)    ( (a).u32[0] )
RADIX_SORT(128x, v4u32_t, KEY128, 8)        /* minialign.c:203-204 */
RADIX_SORT(64x, v2u32_t, KEY64, 4)          /* minialign.c:205-206 */

/* ---- options ---- */
/* one option letter with its argument (the handlers of minialign.c:5990-6099); returns nonzero where the reference's range check fails */
static int opt_preset(om_opt_t *o, char const *preset);
static int opt_one(om_opt_t *o, char c, char const *arg, size_t l)
{
	char buf[256]; if(l > 255) { l = 255; } memcpy(buf, arg, l); buf[l] = 0; arg = buf;
	static uint8_t const base_idx[128] = { ['A'] = 1, ['C'] = 2, ['G'] = 3, ['T'] = 4, ['U'] = 4 };
	switch(c) {
		case 'x': return opt_preset(o, arg);
		case 'k': o->k = (uint32_t)atoi(arg); return !(o->k > 1 && o->k < 32);
		case 'w': o->w = (uint32_t)atoi(arg); return !(o->w > 1 && o->w < 32);
		case 'B': o->b = (uint32_t)atoi(arg); return !(o->b > 1 && o->b < 32);
		case 'f': {
			int bad = 0; o->n_frq = 0;
			for(char *p = buf; *p; ) {
				char *e = p; while(*e && !strchr(",;:/", *e)) { e++; }
				if(e > p) {
					if(o->n_frq >= 7) { return 1; }
					float f = o->frq[o->n_frq] = (float)atof(p);
					if(!(f >= 0.0 && f < 1.0) || (o->n_frq > 0 && !(o->frq[o->n_frq - 1] > f))) { bad = 1; }
					o->n_frq++;
				}
				if(!*e) { break; } p = e + 1;
			}
			return bad || o->n_frq == 0;
		}
		case 'L': o->min_len = (uint32_t)atoi(arg); return !(o->min_len > 0);
		case 'a': { int m = atoi(arg); for(int i = 0; i < 16; i++) { if((i & 3) == (i >> 2)) { o->p.score_matrix[i] = (int8_t)m; } } return !(m > 0 && m < 7); }
		case 'b': { int x = atoi(arg); for(int i = 0; i < 16; i++) { if((i & 3) != (i >> 2)) { o->p.score_matrix[i] = (int8_t)-x; } } return !(x > 0 && x < 7); }
		case 'e': {                      /* mm_opt_mod, minialign.c:6045: "<query base><ref base><delta>", ... */
			for(char *p = buf; *p; ) {
				char *e = p; while(*e && !strchr(",;:/", *e)) { e++; }
				if(e > p) {
					if(e - p < 3 || (p[0] & 0x80) || (p[1] & 0x80) || !base_idx[(int)p[0]] || !base_idx[(int)p[1]]) { return 1; }
					char sv = *e; *e = 0;
					o->p.score_matrix[(base_idx[(int)p[1]] - 1) * 4 + (base_idx[(int)p[0]] - 1)] += (int8_t)atoi(p + 2);
					*e = sv;
				}
				if(!*e) { break; } p = e + 1;
			}
			return 0;
		}
		case 'p': { int gi = atoi(arg); o->p.gi = (int8_t)gi; return !(gi < 32); }
		case 'q': { int ge = atoi(arg); o->p.ge = (int8_t)ge; return !(ge > 0 && ge < 32); }
		case 'r': {
			int g0 = atoi(arg), g1 = g0; char const *cm = arg; while(*cm && !strchr(",;:/", *cm)) { cm++; } if(*cm) { g1 = atoi(cm + 1); }
			o->p.gfa = (int8_t)g0; o->p.gfb = (int8_t)g1;
			return !(g0 >= 0 && g0 < 32 && g1 >= 0 && g1 < 32);
		}
		case 'Y': { int x = atoi(arg); o->p.xdrop = (int8_t)x; return !(x > 10 && x < 128); }
		case 's': o->min_score = (uint32_t)atoi(arg); return !(o->min_score > 0);
		case 'm': o->min_ratio = (float)atof(arg); return !(o->min_ratio > 0.0 && o->min_ratio < 1.0);
		case 'W': o->wlen = (uint32_t)atoi(arg); return 0;
		case 'G': o->glen = (uint32_t)atoi(arg); return 0;
		case 't': case '1': case '2': case 'v': return 0;
		case 'O': {
			static struct { char const *k; uint32_t v; } const t[] = { { "sam", 0 }, { "maf", 1 }, { "blast6", 2 }, { "paf", 5 } };
			for(int i = 0; i < 4; i++) { if(strcmp(arg, t[i].k) == 0) { o->format = t[i].v; return 0; } }
			return 1;
		}
		case 'c': {                      /* mm_opt_circular, minialign.c:5986-5997: no name, `*' or `-' marks every sequence */
			o->circ_set = 1;
			if(l == 0) { return 0; }
			if(l == 1 && (arg[0] == '*' || arg[0] == '-')) { free(o->circ_names); o->circ_names = NULL; return 0; }
			size_t have = o->circ_names ? strlen(o->circ_names) : 0;
			o->circ_names = (char *)realloc(o->circ_names, have + l + 2);
			if(have) { o->circ_names[have++] = ','; }
			memcpy(o->circ_names + have, arg, l); o->circ_names[have + l] = 0;
			return 0;
		}
		case 'X': o->flag |= 0x01; o->ava = 1; return 0;      /* MM_AVA: every file is mapped onto every file (minialign.c:6377); QUIRK: the bit is also the RG tag's */
		case 'A': o->flag |= 0x10; return 0;      /* MM_COMP: no effect on the mapping; QUIRK: the bit is also the AS tag's */
		case 'C': return 0;                       /* base ids: parsed, unused (minialign.c:3768 pins qid to 0) */
		case 'P': o->flag |= 0x08; return 0;
		case 'Q': o->keep_qual = 1; return 0;
		case 'T': {                      /* mm_opt_tags + mm_print_tag2flag, minialign.c:5928, 5631 */
			static char const *const names[] = { "RG", "CO", "NH", "IH", "AS", "XS", "NM", "SA", "MD", "CG", "ID", "SQ" };
			for(char *p = buf; *p; ) {
				char *e = p; while(*e && !strchr(",;:/", *e)) { e++; }
				if(e > p) { if(e - p != 2) { return 1; } for(int i = 0; i < 12; i++) { if(p[0] == names[i][0] && p[1] == names[i][1]) { o->tags |= 1ULL << i; } } }
				if(!*e) { break; } p = e + 1;
			}
			return 0;
		}
		case 'R': {                      /* mm_opt_rg, minialign.c:5890-5921: a backslash turns the next character into a tab */
			free(o->rg_line); free(o->rg_id); o->rg_line = o->rg_id = NULL; o->flag &= ~1ULL;
			char *line = (char *)malloc(l + 1); size_t n = 0;
			for(char const *p = arg; *p; p++) { if(*p == '\\') { p++; line[n++] = '\t'; if(!*p) { break; } } else { line[n++] = *p; } }
			line[n] = 0;
			for(char *p = line; *p; ) {
				char *e = p; while(*e && !strchr("\t\r\n", *e)) { e++; }
				if(e > p && strncmp(p, "ID:", 3) == 0) { o->rg_line = line; o->rg_id = strndup(p, (size_t)(e - p)); o->flag |= 1ULL; break; }
				if(!*e) { break; } p = e + 1;
			}
			if(o->rg_id == NULL) { free(line); return 1; }
			return 0;
		}
		default: return 1;
	}
}
static int opt_apply(om_opt_t *o, char const *s)
{
	int rc = 0;
	while(*s) {
		while(*s == ' ') { s++; }
		if(*s != '-') { break; }
		char c = s[1]; s += 2;
		char const *arg = s;
		while(*s && *s != ' ') { s++; }
		rc |= opt_one(o, c, arg, (size_t)(s - arg));
	}
	return rc;
}
static void opt_defaults(om_opt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->k = 15; o->w = 32; o->b = 14; o->n_frq = 3; o->frq[0] = 0.05f; o->frq[1] = 0.01f; o->frq[2] = 0.001f;
	o->wlen = 7000; o->glen = 7000; o->min_score = 50; o->min_ratio = 0.3f; o->min_len = 1;
	for(int i = 0; i < 16; i++) { o->p.score_matrix[i] = (i & 3) == (i >> 2) ? 1 : -1; }
	o->p.gi = 1; o->p.ge = 1; o->p.gfa = 0; o->p.gfb = 0; o->p.xdrop = 50;
}
/* preset tree, minialign.c:5853-5878: each name applies its line, the next name is looked up among its children (mm_opt_preset, :5880-5889) */
typedef struct preset_s { char const *key, *val; struct preset_s const *kids; } preset_t;
static preset_t const pt_r7[] = { { "1d", "", 0 }, { "2d", "", 0 }, { 0, 0, 0 } };
static preset_t const pt_1[] = { { "1d", "", 0 }, { "1dsq", "-b6 -r4,4", 0 }, { "2d", "-b6 -r4,4", 0 }, { 0, 0, 0 } };
static preset_t const pt_45[] = { { "1", "", pt_1 }, { "1d", "", 0 }, { "1dsq", "-b6 -r4,4", 0 }, { "2d", "-b6 -r4,4", 0 }, { 0, 0, 0 } };
static preset_t const pt_r9[] = { { "4", "-a2", pt_45 }, { "5", "-a2", pt_45 }, { "1d", "", 0 }, { "1dsq", "-b6 -r4,4", 0 }, { "2d", "-b6 -r4,4", 0 }, { 0, 0, 0 } };
static preset_t const pt_ont[] = { { "r7", "-b4", pt_r7 }, { "r9", "", pt_r9 }, { "1d", "-a2", 0 }, { "1dsq", "-a2 -b6 -r4,4", 0 }, { "2d", "-a2 -b6 -r4,4", 0 }, { 0, 0, 0 } };
static preset_t const pt_pacbio[] = { { "clr", "", 0 }, { "ccs", "-b5 -p6 -p2", 0 }, { 0, 0, 0 } };
static preset_t const pt_root[] = { { "pacbio", "-k15 -w10 -a2 -b4 -p4 -q2 -r3,3 -Y50 -s50 -m0.3", pt_pacbio }, { "ont", "-k15 -w10 -a3 -b5 -p6 -q2 -r3,3 -Y50 -s50 -m0.3", pt_ont },
	{ "ava", "-k15 -w5 -a2 -b3 -p0 -q2 -Y50 -s30 -m0.05", 0 }, { 0, 0, 0 } };
static int opt_preset(om_opt_t *o, char const *preset)
{
	preset_t const *c = pt_root; int any = 0;
	for(char const *p = preset; *p; ) {
		char const *e = p; while(*e && *e != '.' && *e != ':') { e++; }
		if(e > p) {
			preset_t const *q = c; while(q && q->key && !(strlen(q->key) == (size_t)(e - p) && strncmp(q->key, p, (size_t)(e - p)) == 0)) { q++; }
			if(!q || !q->key) { return 1; }
			if(opt_apply(o, q->val)) { return 1; }
			c = q->kids; any = 1;
		}
		if(!*e) { break; } p = e + 1;
	}
	return !any;
}
static int opt_check(om_opt_t *o)           /* mm_opt_check_sanity, minialign.c:6097-6112 */
{
	int x = 0; for(int i = 0; i < 16; i++) { if(-(int)o->p.score_matrix[i] > x) { x = -(int)o->p.score_matrix[i]; } }
	int const gfa = o->p.gfa, gfb = o->p.gfb, ge = o->p.ge;
	int rc = 0;
	if(!(gfa == 0 || gfa > ge) || !(gfb == 0 || gfb > ge)) { rc = 1; }
	if((gfa == 0) != (gfb == 0)) { rc = 1; }
	if(!(gfa == 0 || gfb == 0 || gfa + gfb > x)) { rc = 1; }
	if(o->w >= 32) { o->w = (uint32_t)(int)(2.0 / 3.0 * o->k + .499); }   /* minialign.c:6111 */
	return rc;
}
int om_opt_init(om_opt_t *o, char const *preset)
{
	opt_defaults(o);
	int rc = 0;
	if(preset && *preset) { rc = opt_preset(o, preset); }
	return rc | opt_check(o);
}
int om_opt_parse(om_opt_t *o, int argc, char const *const *argv, char const **files, int max_files, int *n_files)
{
	opt_defaults(o);
	int rc = 0, nf = 0;
	for(int i = 1; i < argc; i++) {
		char const *a = argv[i];
		if(a[0] == '-' && a[1]) {
			char const *arg = a + 2;
			if(*arg == 0 && i + 1 < argc && strchr("xkwabpqrYsmtWGfBLe12TRO", a[1])) { arg = argv[++i]; }
			/* options with an optional argument take the next word unless it looks like an option (mm_opt_parse_argv, minialign.c:5786) */
			else if(*arg == 0 && i + 1 < argc && strchr("cvC", a[1]) && (argv[i + 1][0] != '-' || argv[i + 1][1] == 0)) { arg = argv[++i]; }
			rc |= opt_one(o, a[1], arg, strlen(arg));
		} else if(nf < max_files) { files[nf++] = a; }
	}
	if(n_files) { *n_files = nf; }
	return rc | opt_check(o);
}

/* ---- FASTA / FASTQ (bseq_read_fasta, minialign.c:1996-2090; encoding minialign.c:223-229) ---- */
static uint8_t const encaf[16] = { [('A' & 0xf)] = 0, [('C' & 0xf)] = 1, [('G' & 0xf)] = 2, [('T' & 0xf)] = 3, [('U' & 0xf)] = 3, [('N' & 0xf)] = 4 };
om_seqs_t om_read_fasta(char const *fn) { return om_read_fasta_ex(fn, 0, 0); }
/* bseq_read_fasta as a walk over the whole text (minialign.c:1996-2090; the reference refills a buffer, which does not change what is read):
 *   - the file type is the first '>' or '@' among the first four bytes (minialign.c:1784-1792); what stands in front of it is dropped
 *   - name: spaces skipped, then up to the first space or end of line, tabs rewritten to spaces, one trailing CR dropped; a comment exists when
 *     the name ended at a space: spaces skipped, to the end of the line, tabs to spaces, one CR and then the spaces at the end dropped
 *   - bases: every byte of the following lines goes through the low-nibble table -- a CR too (it reads as A) -- until the record delimiter
 *     ('>' for FASTA, '+' for FASTQ) shows up ANYWHERE in a line, or the text ends
 *   - FASTQ: the rest of the '+' line is skipped, then quality lines are taken until their length reaches the number of bases (counted without
 *     a trailing CR when the qualities are kept, with it when they are only skipped), newlines after that are skipped and the next byte must be '@'
 * om_read_error is set when the text is not in this shape (the reference gives up on the whole run: exit 1, no output). */
#define OM_SEQ_MARGIN 4096
int om_read_error = 0;
om_seqs_t om_read_fasta_ex(char const *fn, int keep_qual, int keep_comment)
{
	om_seqs_t r = { 0, 0 };
	om_read_error = 0;
	FILE *fp = fopen(fn, "r");
	if(!fp) { om_read_error = 1; return r; }
	size_t cap = 1 << 20, n = 0; char *d = (char *)malloc(cap + 1); size_t got;
	while((got = fread(d + n, 1, cap - n, fp)) > 0) { n += got; if(n == cap) { cap *= 2; d = (char *)realloc(d, cap + 1); } }
	fclose(fp);
	char const *p = d, *t = d + n;
	char delim = 0;
	for(int i = 0; i < 4 && p < t; i++) { char c = *p; if(c == '>' || c == '@') { delim = c; break; } p++; }
	if(!delim) { free(d); om_read_error = 1; return r; }
	char const dv = delim == '@' ? '+' : delim;
	vec_t(om_seq_t) v = { 0, 0, 0 };
	while(p < t) {
		if(*p++ != delim) { om_read_error = 1; break; }
		om_seq_t s; memset(&s, 0, sizeof(s));
		while(p < t && *p == ' ') { p++; }
		char const *b0 = p; while(p < t && *p != ' ' && *p != '\n') { p++; }
		size_t ln = (size_t)(p - b0); int const has_comment = p < t && *p == ' ';
		if(ln > 0 && b0[ln - 1] == '\r') { ln--; }
		s.l_name = (uint32_t)ln; s.name = strndup(b0, ln);
		for(size_t i = 0; i < ln; i++) { if(s.name[i] == '\t') { s.name[i] = ' '; } }
		if(p < t) { p++; }
		if(has_comment) {
			while(p < t && *p == ' ') { p++; }
			char const *c0 = p; while(p < t && *p != '\n') { p++; }
			size_t cl = (size_t)(p - c0);
			if(p < t) { p++; }
			if(cl > 0 && c0[cl - 1] == '\r') { cl--; }
			while(cl > 0 && c0[cl - 1] == ' ') { cl--; }
			if(keep_comment) { s.comment = strndup(c0, cl); for(size_t i = 0; i < cl; i++) { if(s.comment[i] == '\t') { s.comment[i] = ' '; } } }
		}
		/* bases */
		uint64_t scap = 256; s.seq = (uint8_t *)malloc(scap);
		int at_delim = 0;
		while(p < t) {
			char const *l0 = p; while(p < t && *p != '\n' && *p != dv) { p++; }
			uint64_t ll = (uint64_t)(p - l0);
			if(s.l_seq + ll + 1 > scap) { scap = (s.l_seq + ll + 1) * 2; s.seq = (uint8_t *)realloc(s.seq, scap); }
			for(uint64_t i = 0; i < ll; i++) { s.seq[s.l_seq++] = encaf[l0[i] & 0x0f]; }
			if(p < t && *p == dv) { at_delim = 1; break; }
			if(p < t) { p++; }
		}
		{	/* zero margins on both sides: an extension may start a few bases past the end of a sequence (a seed at the wrap of a circular reference,
			 * minialign.c:3823-3827 only pulls it back by k) and the reference then reads the terminators / margins around its copy of the sequence */
			uint8_t *m = (uint8_t *)calloc(1, (size_t)s.l_seq + 2 * OM_SEQ_MARGIN); memcpy(m + OM_SEQ_MARGIN, s.seq, s.l_seq); free(s.seq); s.seq = m + OM_SEQ_MARGIN;
		}
		if(delim == '@' && at_delim) {
			while(p < t && *p != '\n') { p++; }          /* the '+' line */
			if(p < t) { p++; }
			uint64_t acc = 0, lim = s.l_seq, ql = 0;
			if(keep_qual) { s.qual = (char *)malloc(lim + 2); }
			while(p < t) {
				char const *l0 = p; while(p < t && *p != '\n') { p++; }
				uint64_t ll = (uint64_t)(p - l0);
				if(keep_qual) {
					uint64_t kl = ll; if(kl > 0 && l0[kl - 1] == '\r') { kl--; }
					s.qual = (char *)realloc(s.qual, ql + kl + 2); memcpy(s.qual + ql, l0, kl); ql += kl; acc += kl;
				} else { acc += ll; }
				if(p >= t) { break; }
				if(acc >= lim) { break; }
				p++;
			}
			if(keep_qual) { s.qual[ql] = 0; }
			while(p < t && *p == '\n') { p++; }
		}
		vec_push(om_seq_t, v, s);
	}
	free(d);
	r.a = v.a; r.n = v.n;
	om_seqs_drop_short(&r, 1);       /* -L 1, the default (minialign.c:6145) */
	return r;
}
void om_seqs_drop_short(om_seqs_t *s, uint32_t min_len)       /* sequences shorter than min_len are squashed by the reader (minialign.c:2077) */
{
	uint64_t j = 0;
	for(uint64_t i = 0; i < s->n; i++) { if(s->a[i].l_seq >= min_len) { s->a[j++] = s->a[i]; } else { free(s->a[i].name); free(s->a[i].seq - OM_SEQ_MARGIN); free(s->a[i].qual); free(s->a[i].comment); } }
	s->n = j;
}
void om_seqs_free(om_seqs_t *s)
{
	for(uint64_t i = 0; i < s->n; i++) { free(s->a[i].name); free(s->a[i].seq - OM_SEQ_MARGIN); free(s->a[i].qual); free(s->a[i].comment); }
	free(s->a); s->a = NULL; s->n = 0;
}

/* ---- sketch ---- */
static uint32_t crc_tbl[256]; static int crc_init_done = 0;
static void crc_init(void)
{
	for(uint32_t i = 0; i < 256; i++) { uint32_t c = i; for(int k = 0; k < 8; k++) { c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1; } crc_tbl[i] = c; }
	crc_init_done = 1;
}
static inline uint64_t crc32c_u64(uint64_t crc, uint64_t v)     /* _mm_crc32_u64 */
{
	uint32_t c = (uint32_t)crc;
	for(int i = 0; i < 8; i++) { c = crc_tbl[(c ^ (uint8_t)(v >> (8 * i))) & 0xff] ^ (c >> 8); }
	return (uint64_t)c;
}
#define HASH64(k0, k1, mask)    ( (crc32c_u64((k1), (k1)) ^ (k0)) & (mask) )       /* minialign.c:2353 */

static uint64_t sketch_intl(uint32_t w_, uint32_t k_, uint8_t const *seq, uint32_t len, uint64_t *out, int circular);
uint64_t om_sketch(uint32_t w_, uint32_t k_, uint8_t const *seq, uint32_t len, uint64_t *out) { return sketch_intl(w_, k_, seq, len, out, 0); }
/* with circular != 0 the window runs on over the first min(len, w) bases (mm_sketch_cap, minialign.c:2438-2444, "tail margin for circular sequences") */
static uint64_t sketch_intl(uint32_t w_, uint32_t k_, uint8_t const *seq, uint32_t len, uint64_t *out, int circular)
{
	if(!crc_init_done) { crc_init(); }
	uint64_t r[64]; for(int i = 0; i < 64; i++) { r[i] = UINT64_MAX; }     /* QUIRK: the reference initialises r[0..32) only (minialign.c:2373); w < 16 keeps all reads inside */
	uint64_t const kk = k_ - 1, shift1 = 2 * kk, mask = (1ULL << 2 * k_) - 1, w = w_;
	uint64_t *q = out;
	uint8_t const *p = seq, *t = seq + len;
	uint64_t u = 0, k0 = 0, k1 = 0;
	#define PUSH_KMER() { uint64_t c = *p++; k0 = (k0 << 2 | c) & mask; k1 = (k1 >> 2) | ((3ULL ^ c) << shift1); }
	#define LOOP_CORE(_h) { \
		PUSH_KMER(); \
		uint64_t km = k0 < k1 ? k0 : k1, kx = k0 < k1 ? k1 : k0, m = k0 < k1 ? 0 : 0x80; \
		uint64_t hh = HASH64(km, kx, mask) << 8 | i | m; f = MIN2(f, hh); uint64_t v = MIN2(f, r[i + 1]); \
		if((v == hh) | (v - u)) { *q++ = v; } \
		u = v; (_h) = hh; \
	}
	for(uint64_t i = 0; i < kk && p < t; i++) { PUSH_KMER(); }
	while((int64_t)(t - p) >= (int64_t)w) {
		for(uint64_t i = 0, f = UINT64_MAX; i < w; i++) { uint64_t h; LOOP_CORE(h); r[i] = h; }
		for(uint64_t i = 0, rr = UINT64_MAX; i < w; i++) { rr = MIN2(rr, r[w - i - 1]); r[w - i - 1] = rr; }
	}
	uint64_t l = (uint64_t)(t - p);
	if(l > 0) {
		for(uint64_t i = 0, f = UINT64_MAX; i < l; i++) { uint64_t h; LOOP_CORE(h); r[w + i] = h + w; }
		/* fold the backward-min array, move it to the head, adjust u (minialign.c:2427-2432); read only by the cap that may follow */
		for(uint64_t i = 0, rr = UINT64_MAX; i < w; i++) { rr = MIN2(rr, r[w + l - i - 1]); r[w + l - i - 1] = rr; }
		for(uint64_t i = 0; i < w; i++) { r[i] = r[l + i] - l; }
		u += w - l;
	}
	if(circular) {
		/* mm_sketch_cap: the cap holds i = 0, so up to w more positions, bases taken from the head of the sequence */
		uint64_t lc = MIN2((uint64_t)len, w);
		p = seq; t = seq + lc;
		for(uint64_t i = 0, f = UINT64_MAX; i < w && p < t; i++) { uint64_t h; LOOP_CORE(h); (void)h; }
	}
	#undef PUSH_KMER
	#undef LOOP_CORE
	return (uint64_t)(q - out);
}

/* ---- kh_t: 64 -> 64 ordered linear-probing hash (minialign.c:341-683), literal ---- */
#define KH_SIZE     256
#define KH_THRESH   0.4
#define KH_INIT_VAL UINT64_MAX
typedef struct { uint32_t mask, max, cnt, ub; v4u32_t *a; } kh_t;
static void kh_init_static(kh_t *h, uint64_t size)
{
	size = 0x8000000000000000ULL >> (__builtin_clzll(size - 1) - 1);
	size = MAX2(size, KH_SIZE);
	h->mask = (uint32_t)(size - 1); h->max = (uint32_t)size; h->cnt = 0; h->ub = (uint32_t)(size * KH_THRESH);
	h->a = (v4u32_t *)malloc(sizeof(v4u32_t) * size);
	for(uint64_t i = 0; i < size; i++) { h->a[i].u64[0] = UINT64_MAX; h->a[i].u64[1] = KH_INIT_VAL; }
}
static void kh_clear(kh_t *h)
{
	h->mask = KH_SIZE - 1; h->cnt = 0; h->ub = (uint32_t)(KH_SIZE * KH_THRESH);
	for(uint64_t i = 0; i < KH_SIZE; i++) { h->a[i].u64[0] = UINT64_MAX; h->a[i].u64[1] = KH_INIT_VAL; }
}
typedef struct { uint64_t idx, n; } kh_bidx_t;
static kh_bidx_t kh_allocate(v4u32_t *a, uint64_t k, uint64_t v, uint64_t mask)
{
	#define POLL(_i, _b0, _k1) { \
		int64_t _b = (int64_t)(_b0); \
		while(1) { \
			(_k1) = a[_i].u64[0]; \
			if(_b <= (int64_t)((_k1) & mask) + (int64_t)((_k1) + 2 < 2)) { break; } \
			_b -= (int64_t)(((_i) + 1) & (mask + 1)); \
			(_i) = ((_i) + 1) & mask; \
		} \
	}
	uint64_t i = k & mask, k0 = k, v0 = v, k1;
	POLL(i, i, k1);
	if(k0 == k1) { return (kh_bidx_t){ i, 0 }; }
	uint64_t j = i;
	a[i].u64[0] = k0;
	while(k1 + 2 >= 2) {
		uint64_t v1 = a[i].u64[1];
		a[i].u64[1] = v0;
		k0 = k1; v0 = v1;
		i = (i + 1) & mask;
		POLL(i, k0 & mask, k1);
		a[i].u64[0] = k0;
	}
	a[i].u64[1] = v0;
	return (kh_bidx_t){ j, 1 };
	#undef POLL
}
static void kh_extend(kh_t *h)
{
	uint64_t prev_size = (uint64_t)h->mask + 1, size = 2 * prev_size, mask = size - 1;
	h->mask = (uint32_t)mask; h->ub = (uint32_t)(size * KH_THRESH);
	if(size > h->max) { h->a = (v4u32_t *)realloc(h->a, sizeof(v4u32_t) * size); h->max = (uint32_t)size; }
	for(uint64_t i = 0; i < prev_size; i++) { h->a[i + prev_size].u64[0] = UINT64_MAX; h->a[i + prev_size].u64[1] = KH_INIT_VAL; }
	for(uint64_t i = 0; i < size; i++) {
		uint64_t k = h->a[i].u64[0];
		if(k + 2 < 2 || (k & mask) == i) { continue; }
		uint64_t v = h->a[i].u64[1];
		h->a[i].u64[0] = UINT64_MAX - 1; h->a[i].u64[1] = KH_INIT_VAL;
		kh_allocate(h->a, k, v, mask);
	}
}
static uint64_t *kh_put_ptr(kh_t *h, uint64_t key, uint64_t extend)
{
	if(extend != 0 && h->cnt >= h->ub) { kh_extend(h); }
	kh_bidx_t b = kh_allocate(h->a, key, KH_INIT_VAL, h->mask);
	h->cnt += (uint32_t)b.n;
	return &h->a[b.idx].u64[1];
}

/* ---- index (mm_idx_gen, minialign.c:2767-3040): same key -> value-list map and list order; the 2nd-stage
 *      Robin-Hood probe order is unobservable, so each bucket keeps its (hrem-sorted) array + binary search ---- */
typedef struct { uint64_t hrem; uint32_t pos, rid; } mini_t;     /* mm_mini_t, minialign.c:2661 */
typedef struct {
	uint64_t n_keys;
	uint64_t *key;      /* hrem, ascending */
	uint32_t *start;    /* n_keys + 1 offsets into val */
	uint64_t *val;      /* pos | rid << 32, in post-sort order */
} bkt_t;
struct om_idx_s {
	uint32_t b, w, k, n_occ; uint64_t mask;
	uint32_t occ[16];
	bkt_t *bkt;
	om_seq_t const *s; uint32_t n_seq;
	uint8_t *circular;          /* per sequence, minialign.c:2468 */
};
uint32_t om_idx_occ(om_idx_t const *mi, uint32_t i) { return mi->occ[i]; }

static int cmp_u32(void const *a, void const *b) { uint32_t x = *(uint32_t const *)a, y = *(uint32_t const *)b; return x < y ? -1 : x > y; }

static int name_listed(char const *list, char const *name, uint32_t l_name)
{
	for(char const *p = list; *p; ) {
		char const *e = p; while(*e && !strchr(",;:/", *e)) { e++; }
		if((uint32_t)(e - p) == l_name && memcmp(p, name, l_name) == 0) { return 1; }
		if(!*e) { break; } p = e + 1;
	}
	return 0;
}
om_idx_t *om_idx_build(om_opt_t const *o, om_seq_t const *ref, uint32_t n_ref)
{
	om_idx_t *mi = (om_idx_t *)calloc(1, sizeof(om_idx_t));
	uint32_t b = MIN2(o->k * 2, o->b);
	mi->b = b; mi->w = o->w; mi->k = o->k; mi->n_occ = o->n_frq; mi->mask = (1ULL << b) - 1;
	mi->s = ref; mi->n_seq = n_ref; mi->circular = (uint8_t *)calloc(n_ref + 1, 1);
	uint64_t nb = 1ULL << b;
	typedef vec_t(mini_t) mini_v;
	mini_v *arr = (mini_v *)calloc(nb, sizeof(mini_v));
	/* mm_idx_worker + mm_idx_drain_intl (minialign.c:2790-2860): sketch every sequence, push in reference order */
	for(uint32_t i = 0; i < n_ref; i++) {
		uint64_t *m = (uint64_t *)malloc(sizeof(uint64_t) * (4 * (uint64_t)ref[i].l_seq / o->w + 512));
		/* circular: every sequence when -c came without names, else the named ones (minialign.c:2796, 2963-2964) */
		int circ = o->circ_set && (o->circ_names == NULL || o->circ_names[0] == 0 ? 1 : name_listed(o->circ_names, ref[i].name, ref[i].l_name));
		mi->circular[i] = (uint8_t)circ;
		uint64_t n = sketch_intl(o->w, o->k, ref[i].seq, ref[i].l_seq, m, circ);
		uint64_t w = o->w, base = (uint64_t)-(int64_t)w, v = w;
		for(uint64_t j = 0; j < n; j++) {
			uint64_t u = m[j] & 0x7f, fr = (m[j] >> 7) & 0x01, h = m[j] >> 8;
			base += u <= v ? w : 0; v = u;
			mini_t x = { h >> b, (uint32_t)(base + u), (uint32_t)((i << 1) + fr) };
			vec_push(mini_t, arr[h & mi->mask], x);
		}
		free(m);
	}
	/* mm_idx_count_occ (minialign.c:2867-2900): per-bucket radix sort on hrem, occurrence histogram */
	vec_t(uint32_t) cnt = { 0, 0, 0 };
	for(uint64_t i = 0; i < nb; i++) {
		if(arr[i].n == 0) { continue; }
		radix_sort_128x((v4u32_t *)arr[i].a, arr[i].n);
		uint32_t n = 1;
		for(uint64_t j = 1; j < arr[i].n; j++) {
			if(arr[i].a[j - 1].hrem != arr[i].a[j].hrem) { vec_push(uint32_t, cnt, n); n = 0; }
			n++;
		}
		vec_push(uint32_t, cnt, n);
	}
	/* occ thresholds (minialign.c:2981-2986): k-th smallest count + 1 */
	uint32_t *sorted = (uint32_t *)malloc(sizeof(uint32_t) * (cnt.n + 1));
	memcpy(sorted, cnt.a, sizeof(uint32_t) * cnt.n);
	qsort(sorted, cnt.n, sizeof(uint32_t), cmp_u32);
	for(uint32_t i = 0; i < o->n_frq; i++) {
		if(o->frq[i] <= 0.0) { mi->occ[i] = UINT32_MAX; continue; }
		uint32_t kk = (uint32_t)((1.0 - o->frq[i]) * cnt.n);
		mi->occ[i] = (cnt.n ? sorted[MIN2((uint64_t)kk, cnt.n - 1)] : 0) + 1;
	}
	free(sorted); free(cnt.a);
	/* mm_idx_build_hash (minialign.c:2905-2944): keys with more than occ[n_occ - 1] hits are dropped */
	uint64_t max_cnt = mi->occ[mi->n_occ - 1];
	mi->bkt = (bkt_t *)calloc(nb, sizeof(bkt_t));
	for(uint64_t i = 0; i < nb; i++) {
		uint64_t n = arr[i].n;
		if(n == 0) { continue; }
		bkt_t *bk = &mi->bkt[i];
		bk->key = (uint64_t *)malloc(sizeof(uint64_t) * n); bk->start = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1)); bk->val = (uint64_t *)malloc(sizeof(uint64_t) * n);
		uint64_t nk = 0, nv = 0;
		for(uint64_t j = 0; j < n;) {
			uint64_t e = j + 1; while(e < n && arr[i].a[e].hrem == arr[i].a[j].hrem) { e++; }
			/* QUIRK (minialign.c:2927-2931): when a key exceeds max_cnt the fill cursor `q` is not advanced, so every
			 * later key of the same bucket (in hrem order) is measured from that stale cursor and dropped as well */
			if(e - j > max_cnt) { break; }
			bk->key[nk] = arr[i].a[j].hrem; bk->start[nk] = (uint32_t)nv; nk++;
			for(uint64_t x = j; x < e; x++) { bk->val[nv++] = (uint64_t)arr[i].a[x].pos | ((uint64_t)arr[i].a[x].rid << 32); }
			j = e;
		}
		bk->start[nk] = (uint32_t)nv; bk->n_keys = nk;
		free(arr[i].a);
	}
	free(arr);
	return mi;
}
void om_idx_free(om_idx_t *mi)
{
	if(!mi) { return; }
	for(uint64_t i = 0; i < (1ULL << mi->b); i++) { free(mi->bkt[i].key); free(mi->bkt[i].start); free(mi->bkt[i].val); }
	free(mi->bkt); free(mi->circular); free(mi);
}
uint64_t const *om_idx_get(om_idx_t const *mi, uint64_t minier, uint32_t *n)
{
	bkt_t const *b = &mi->bkt[minier & mi->mask];
	uint64_t key = minier >> mi->b, lo = 0, hi = b->n_keys;
	while(lo < hi) { uint64_t mid = (lo + hi) / 2; if(b->key[mid] < key) { lo = mid + 1; } else { hi = mid; } }
	if(lo >= b->n_keys || b->key[lo] != key) { *n = 0; return NULL; }
	*n = b->start[lo + 1] - b->start[lo];
	return &b->val[b->start[lo]];
}

/* ---- mapper ---- */
typedef struct { uint32_t qs, n; uint64_t const *p; } resc_t;                 /* mm_resc_t, minialign.c:3176 */
typedef struct { uint32_t rsid, rid, lsid, cid; } leaf_t;                       /* mm_leaf_t, minialign.c:3190 */
typedef struct { uint32_t plen, lid; } root_t;                                  /* mm_root_t, minialign.c:3202; aliased by mm_res_t { score, iid } */
typedef struct { uint32_t apos, bpos; } pos_pair_t;
typedef struct {                                                                /* mm_search_t, minialign.c:3218 */
	pos_pair_t cp, tp;
	uint32_t aid, bid;
	uint32_t iid, eid, sid, rev;
	int64_t prem; uint32_t pacc;
	uint32_t crem, srem, narrow;
	uint32_t min_score;
} search_t;
#define MM_CREM 50000
#define MM_SREM 8
#define BIN_N   2           /* mm_bin_t header occupies two pointer slots (minialign.c:3252-3257) */
typedef struct { uint32_t n_aln, plen, lb, ub; } bin_hdr_t;

struct om_align_s {
	om_idx_t const *mi; om_opt_t o;
	uint32_t twlen, tglen; float min_ratio; uint32_t min_score;
	double mcoef, xcoef;
	og_ctx_t *ctx; og_dp_t *dp;
	uint32_t rid, qid, rlen, qlen;
	uint8_t const *rseq, *qseq;
	og_section_t r[2], q[3], t[2];
	og_section_t *rtp;          /* reference-side tail sections: r for a circular reference, else t (minialign.c:3319, 3753) */
	uint8_t tail[128];
	vec_t(resc_t) resc; uint64_t presc;
	vec_t(om_seed_t) seed; uint64_t n_seed;
	vec_t(root_t) root;
	vec_t(v2u32_t) next;
	uint32_t n_res;
	vec_t(uint64_t) bin;            /* slots: bin headers (2 slots) and alignment handles (index + 1 into alns) */
	vec_t(og_alignment_t *) alns;   /* alignment table (the lmm arena of the reference) */
	kh_t pos;
	uint64_t cnt[8];
};
#define OFS(x)          ( (int32_t)0x40000000 - (int32_t)(x) )
#define UD(x, y)        ( ((x) << 1) - (y) )
#define VD(x, y)        ( ((y) << 1) - (x) )
#define U_(x, y)        ( UD(x, y) + OFS(0) )
#define V_(x, y)        ( VD(x, y) + OFS(0) )
#define BARE(x)         ( (x) - OFS(0) )
#define AS(p)           ( (int32_t)((BARE((p)->upos) << 1) + BARE((p)->vpos)) / 3 )
#define BS(p)           ( (int32_t)((BARE((p)->vpos) << 1) + BARE((p)->upos)) / 3 )
#define PS(p)           ( (p)->upos + (p)->vpos )
#define SMASK(x)        ( ((int32_t)(x)) >> 31 )
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
#define KEY(x, y)       ( (uint64_t)(x) ^ ((uint64_t)(x) >> 29) ^ (uint64_t)(y) ^ bswap64((uint64_t)(y)) )   /* minialign.c:3362 */

om_align_t *om_align_init(om_opt_t const *o, om_idx_t const *mi)
{
	om_align_t *a = (om_align_t *)calloc(1, sizeof(om_align_t));
	a->mi = mi; a->o = *o;
	a->twlen = (uint32_t)UD((int32_t)o->wlen, (int32_t)o->wlen); a->tglen = (uint32_t)UD((int32_t)o->glen, (int32_t)o->glen);
	a->min_ratio = o->min_ratio; a->min_score = o->min_score;
	/* QUIRK (minialign.c:4676-4681): both coefficients accumulate score_matrix[0] */
	double mcoef = 0.0, xcoef = 0.0;
	for(uint64_t i = 0; i < 16; i++) { if((i & 0x03) == (i >> 3)) { mcoef += o->p.score_matrix[0]; } else { xcoef += o->p.score_matrix[0]; } }
	a->mcoef = mcoef / 4.0; a->xcoef = xcoef / 12.0;
	a->ctx = og_init(&o->p);
	if(a->ctx == NULL) { free(a); return NULL; }
	a->dp = og_dp_init(a->ctx);
	memset(a->tail, 4, 128);
	a->t[0] = (og_section_t){ 0xfffffffe, 96, a->tail }; a->t[1] = a->t[0];
	kh_init_static(&a->pos, 128);
	return a;
}
void om_align_free(om_align_t *a)
{
	if(!a) { return; }
	og_dp_clean(a->dp); og_clean(a->ctx);
	free(a->resc.a); free(a->seed.a); free(a->root.a); free(a->next.a); free(a->bin.a); free(a->alns.a); free(a->pos.a);
	free(a);
}
void om_counters(om_align_t const *a, uint64_t out[8]) { memcpy(out, a->cnt, sizeof(a->cnt)); }

/* mm_expand, minialign.c:3420-3447 */
static void mm_expand(om_align_t *self, uint32_t n, uint64_t const *r, uint32_t qs)
{
	if(n == 0) { return; }
	vec_reserve(om_seed_t, self->seed, self->seed.n + n);
	for(uint64_t i = 0; i < n; i++) {
		uint32_t rid = (uint32_t)(r[i] >> 32);
		if(rid < self->qid) { continue; }
		uint32_t rs = (uint32_t)r[i];
		uint32_t rmask = -(rid & 0x01);
		int32_t _rs = (int32_t)(rs + (self->mi->k & rmask)), _qs = (int32_t)(qs ^ rmask);
		self->seed.a[self->seed.n++] = (om_seed_t){ .upos = (uint32_t)U_(_rs, _qs), .vpos = (uint32_t)V_(_rs, _qs), .rid = rid >> 1, .lid = INT32_MAX };
	}
}
/* mm_collect_seed, minialign.c:3454-3493 */
static void mm_collect_seed(om_align_t *self)
{
	uint64_t *m = (uint64_t *)malloc(sizeof(uint64_t) * (4 * (uint64_t)self->qlen / self->mi->w + 512));
	uint64_t nm = om_sketch(self->mi->w, self->mi->k, self->qseq, self->qlen, m);
	vec_reserve(resc_t, self->resc, nm + 1);
	uint64_t ns = 0;
	uint32_t max_occ = self->mi->occ[self->mi->n_occ - 1], resc_occ = self->mi->occ[0];
	uint64_t w = self->mi->w, base = (uint64_t)-(int64_t)w, v = w;
	for(uint64_t j = 0; j < nm; j++) {
		uint64_t u = m[j] & 0x7f, fr = (m[j] >> 7) & 0x01, h = m[j] >> 8;
		base += u <= v ? w : 0; v = u;
		uint32_t n;
		uint64_t const *r = om_idx_get(self->mi, h, &n);
		self->cnt[0]++;
		if(n > max_occ) { continue; }
		uint32_t pos = (uint32_t)((base + u + (self->mi->k & -fr)) ^ -fr);
		if(n > resc_occ) { self->resc.a[ns++] = (resc_t){ .p = r, .qs = pos, .n = n }; continue; }
		mm_expand(self, n, r, pos);
	}
	free(m);
	self->resc.n = ns; self->presc = 0; self->root.n = 0;
}
/* mm_seed, minialign.c:3500-3541 */
static uint64_t mm_seed(om_align_t *self, uint64_t cnt)
{
	if(cnt == 0) {
		self->seed.n = 0; self->n_seed = 0;
		mm_collect_seed(self);
	} else {
		if(cnt == 1) { radix_sort_128x((v4u32_t *)self->resc.a, self->resc.n); }     /* key = qs | n << 32 */
		self->seed.n = self->n_seed;
		for(uint64_t i = 0; i < self->seed.n; i++) { self->seed.a[i].lid = INT32_MAX; }
		uint64_t p = self->presc, t = self->resc.n;
		while(p < t && self->resc.a[p].n <= self->mi->occ[cnt]) { mm_expand(self, self->resc.a[p].n, self->resc.a[p].p, self->resc.a[p].qs); p++; }
		self->presc = p;
	}
	self->n_seed = self->seed.n;
	if(self->seed.n == 0) { return 0; }
	om_seed_t sentinel = { .rid = INT32_MAX, .upos = (uint32_t)INT32_MIN, .vpos = (uint32_t)INT32_MIN, .lid = INT32_MAX };
	vec_push(om_seed_t, self->seed, sentinel);
	{	/* analysis hook (tools/k2s_model.c): the seed arrays as they go into the sort, one record { n, n x 16 bytes } per call */
		static FILE *dump = NULL; static int asked = 0;
		if(!asked) { asked = 1; const char *fn = getenv("OM_DUMP_SEEDS"); if(fn) { dump = fopen(fn, "wb"); } }
		if(dump) { uint64_t n = self->seed.n; fwrite(&n, 8, 1, dump); fwrite(self->seed.a, sizeof(om_seed_t), n, dump); fflush(dump); }
	}
	radix_sort_128x((v4u32_t *)self->seed.a, self->seed.n);
	self->cnt[1] += self->seed.n;
	return self->seed.n;
}

/* window arithmetic on (vpos, vpos, rid, upos) vectors, minialign.c:3366-3402, written out per element (signed 32-bit) */
typedef struct { int32_t e[4]; } v4;     /* e[0] = upos, e[1] = rid, e[2] = vpos (vub side), e[3] = vpos (vlb side) */
static inline v4 load_pv(om_seed_t const *p) { return (v4){ { (int32_t)p->upos, (int32_t)p->rid, (int32_t)p->vpos, (int32_t)p->vpos } }; }
static inline v4 window(int32_t len) { return (v4){ { len, 0, len, 0 } }; }       /* _seta(0, len, 0, len): lane0 = len, 1 = 0, 2 = len, 3 = 0 */
static inline v4 add4(v4 a, v4 b) { v4 r; for(int i = 0; i < 4; i++) { r.e[i] = (int32_t)((uint32_t)a.e[i] + (uint32_t)b.e[i]); } return r; }
static inline v4 sub4(v4 a, v4 b) { v4 r; for(int i = 0; i < 4; i++) { r.e[i] = (int32_t)((uint32_t)a.e[i] - (uint32_t)b.e[i]); } return r; }
/* _mask_v4i32(_gt_v4i32(d, u)): 4 bits per lane, lane 0 in the low nibble */
static inline uint32_t inside_mask(v4 u, v4 d) { uint32_t m = 0; for(int i = 0; i < 4; i++) { if(d.e[i] > u.e[i]) { m |= 0xfu << (4 * i); } } return m; }
#define INSIDE_WV(u, d)     ( inside_mask(u, d) == 0xf000 )
#define INSIDE_UUB(u, d)    ( (inside_mask(u, d) & 0xff) == 0x00 )
static inline v4 update_wv(v4 w, v4 f)       /* _update_wv, minialign.c:3380-3388 */
{
	v4 dv = sub4(w, f);
	/* shuffle (3,0,1,2) -> lanes (dv2, dv1, dv0, dv3); keep lanes 0 and 2 */
	v4 sv = { { dv.e[2], 0, dv.e[0], 0 } };
	return sub4(w, sv);
}
static inline int32_t pdiff(v4 w, v4 f) { v4 dv = sub4(w, f); return (int32_t)((uint32_t)dv.e[0] + (uint32_t)dv.e[2]); }

/* mm_chain_seeds, minialign.c:3547-3625 */
static uint64_t mm_chain_seeds(om_align_t *self)
{
	om_seed_t *s = self->seed.a;
	leaf_t *ls = (leaf_t *)self->seed.a;
	root_t *c = self->root.a;
	uint32_t ncid = 0, nlid = (uint32_t)self->n_seed + 1;
	uint32_t nlsid = 0, tsid = (uint32_t)self->n_seed;
	v4 tv = window((int32_t)self->twlen);
	while(nlsid < tsid) {
		uint32_t lid = nlid++;
		ls[lid] = (leaf_t){ .rsid = nlsid, .lsid = nlsid, .rid = s[nlsid].rid, .cid = UINT32_MAX };
		uint32_t plen = PS(&s[nlsid]), scnt = 1;
		uint64_t nrsid = nlsid; nlsid = UINT32_MAX;
		while(1) {
			uint32_t rsid = (uint32_t)nrsid; nrsid = 0;
			v4 wv = add4(load_pv(&s[rsid]), tv);
			for(uint32_t sid = rsid + 1; ; sid++) {
				v4 fv = load_pv(&s[sid]);
				if(!INSIDE_WV(wv, fv)) {
					nlsid = MIN2(nlsid, sid);
					if(INSIDE_UUB(wv, fv)) { continue; }
					break;
				}
				wv = update_wv(wv, fv);
				int64_t di = (int64_t)(((uint64_t)(int64_t)pdiff(wv, fv) << 32) | sid);
				nrsid = (uint64_t)MAX2((int64_t)nrsid, di);
			}
			if(nrsid == 0) { nrsid = rsid; break; }
			if(s[(uint32_t)nrsid].lid != INT32_MAX) { nrsid = (uint32_t)nrsid; break; }
			s[(uint32_t)nrsid].lid = lid; scnt++;
			if(nlsid <= nrsid) { nlsid = UINT32_MAX; }            /* QUIRK: compares against the full 64-bit (pdiff << 32 | sid) value */
		}
		if(nrsid == ls[lid].lsid) { continue; }
		uint32_t cid = UINT32_MAX;
		if(s[nrsid].lid < lid) {
			nrsid = ls[s[nrsid].lid].rsid;
			cid = ls[s[nrsid].lid].cid;
		}
		if(cid == UINT32_MAX) { cid = ncid++; c[cid] = (root_t){ .lid = lid, .plen = (uint32_t)OFS(0) }; }
		ls[lid].cid = cid; ls[lid].rsid = (uint32_t)nrsid;
		plen = (uint32_t)OFS((uint32_t)d2u32((1.0 - 1.0 / (double)scnt) * (double)(uint32_t)(PS(&s[nrsid]) - plen)));
		if(plen < c[cid].plen) { c[cid] = (root_t){ .plen = plen, .lid = lid }; }
	}
	self->root.n = ncid; self->seed.n = nlid;
	return ncid;
}
/* mm_circularize, minialign.c:3632-3696: a chain whose root seed lies within the window of the end of a circular reference is linked to a leaf
 * seed just behind the origin: the far chain is switched off (top bit of plen), its length and root seed pass to the near one */
static void mm_circularize(om_align_t *self)
{
	root_t *c = self->root.a;
	om_seed_t *s = self->seed.a;
	leaf_t *ls = (leaf_t *)self->seed.a;
	uint32_t blid = (uint32_t)self->n_seed + 1, tlid = (uint32_t)self->seed.n;
	v4 tv = window((int32_t)self->twlen);
	for(uint32_t rcid = 0; rcid < self->root.n; rcid++) {
		uint32_t rlid = c[rcid].lid;
		uint32_t rsid = ls[rlid].rsid, rid = ls[rlid].rid;
		int32_t as = AS(&s[rsid]);
		if(self->mi->circular[rid] == 0 || (uint32_t)(self->mi->s[rid].l_seq - (uint32_t)as) > self->twlen) { continue; }
		uint32_t rlen = self->mi->s[rid].l_seq;
		int32_t uofs = (int32_t)((rlen << 1) - 0), vofs = (int32_t)((0 << 1) - rlen);             /* _ud(rlen, 0), _vd(rlen, 0) */
		v4 ov = { { uofs, 0, vofs, vofs } };
		while(blid < tlid && s[ls[blid].lsid].rid < rid) { blid++; }
		uint32_t vub = s[rsid].vpos - (uint32_t)vofs + self->twlen;
		while(blid < tlid && s[ls[blid].lsid].vpos > vub) { blid++; }
		uint64_t llid = UINT64_MAX;
		v4 rv = sub4(add4(load_pv(&s[rsid]), tv), ov);
		for(uint32_t lid = blid; lid < tlid; lid++) {
			v4 lv = load_pv(&s[ls[lid].lsid]);
			if(!INSIDE_WV(rv, lv)) { continue; }
			uint32_t cid = ls[lid].cid;
			if(cid == UINT32_MAX || (c[cid].plen & 0x80000000u)) { continue; }
			llid = MIN2(llid, ((uint64_t)c[cid].plen << 32) | lid);
		}
		if(llid == UINT64_MAX) { continue; }
		uint32_t pd = (uint32_t)(llid >> 32); llid = (uint32_t)llid;
		uint32_t lcid = ls[llid].cid;
		c[lcid].lid = rlid;
		c[lcid].plen |= 0x80000000u;
		s[ls[llid].lsid].lid = ~ls[rlid].rsid;
		c[rcid].plen -= (uint32_t)OFS(pd);
		ls[rlid].rsid = ls[llid].rsid;
	}
}
/* mm_chain, minialign.c:3702-3721 */
static uint64_t mm_chain(om_align_t *self)
{
	vec_reserve(om_seed_t, self->seed, self->seed.n + self->seed.n);
	vec_reserve(root_t, self->root, self->seed.n);
	vec_reserve(v2u32_t, self->next, self->seed.n);
	self->root.n = 0; self->next.n = 0;
	if(mm_chain_seeds(self) == 0) { return 0; }
	mm_circularize(self);
	radix_sort_64x((v2u32_t *)self->root.a, self->root.n);
	self->cnt[2] += self->root.n;
	return self->root.n;
}

/* sections: _sec_fw / _sec_rv, minialign.c:3727-3736 */
static void init_ref(om_align_t *self, uint32_t rid)
{
	om_seq_t const *ref = &self->mi->s[rid];
	self->rid = rid; self->rlen = ref->l_seq; self->rseq = ref->seq;
	self->r[0] = (og_section_t){ rid << 1, ref->l_seq, ref->seq };
	self->r[1] = (og_section_t){ (rid << 1) + 1, ref->l_seq, og_mirror(ref->seq, ref->l_seq) };
	self->rtp = self->mi->circular[rid] ? self->r : self->t;        /* a circular reference continues into itself (minialign.c:3753) */
}
static void init_query(om_align_t *self, uint32_t l_seq, uint8_t const *seq)
{
	self->qid = 0; self->qlen = l_seq; self->qseq = seq;
	self->q[0] = (og_section_t){ 0, l_seq, seq };
	self->q[1] = (og_section_t){ 1, l_seq, og_mirror(seq, l_seq) };
	self->q[2] = self->q[0];
}

static bin_hdr_t *bin_at(om_align_t *self, uint64_t iid) { return (bin_hdr_t *)&self->bin.a[iid]; }
static og_alignment_t *aln_of(om_align_t *self, uint64_t slot) { return self->alns.a[self->bin.a[slot] - 1]; }

/* mm_finish_root, minialign.c:3795-3813 */
static uint64_t mm_finish_root(om_align_t *self, search_t *st)
{
	root_t *r = self->root.a;
	bin_hdr_t *bin = bin_at(self, st->iid);
	if(bin->n_aln == 0 || r[st->eid].plen > (uint32_t)OFS(self->min_score)) {
		self->bin.n = st->iid; self->n_res--; st->crem--;
	} else {
		st->crem = st->crem != 0 ? MM_CREM : 0;
	}
	return st->crem == 0;
}
/* mm_search_load_pos, minialign.c:3818-3834 */
static pos_pair_t mm_search_load_pos(om_align_t *self, om_seed_t const *p, uint32_t *rev)
{
	*rev = BS(p) < 0;
	pos_pair_t cp = { .apos = (uint32_t)AS(p), .bpos = (uint32_t)(BS(p) + (SMASK(BS(p)) & (int32_t)self->qlen)) };
	if(cp.apos >= self->rlen || cp.bpos >= self->qlen) {
		cp.apos -= MIN2(cp.apos, self->mi->k);
		cp.bpos -= MIN2(cp.bpos, self->mi->k);
	}
	return cp;
}
/* mm_search_load_root, minialign.c:3839-3883 */
static uint64_t mm_search_load_root(om_align_t *self, search_t *st, uint32_t cid)
{
	om_seed_t const *s = self->seed.a; leaf_t const *ls = (leaf_t const *)self->seed.a;
	uint32_t lid = self->root.a[cid].lid;
	uint32_t plen = (uint32_t)OFS(self->root.a[cid].plen);
	if(plen * self->mcoef < 2.0 * self->min_score) { return 1; }
	self->next.n = 0;
	/* open result bin: a header with lb = UINT32_MAX */
	vec_reserve(uint64_t, self->bin, self->bin.n + BIN_N);
	uint32_t iid = (uint32_t)self->bin.n;
	/* QUIRK (build-dependent): the source initialises the header with .lb = UINT32_MAX through a type-punned
	 * compound literal (minialign.c:3855); gcc -O3 drops that store (strict aliasing), so in the reference *as built*
	 * every bin starts all-zero -- verified on oracle/_ref (204/204 bins end with lb == 0).  The built binary is the oracle. */
	bin_hdr_t hdr = { .n_aln = 0, .plen = 0, .lb = 0, .ub = 0 };
	memcpy(&self->bin.a[iid], &hdr, sizeof(hdr)); self->bin.n += BIN_N;
	root_t *r = self->root.a;
	uint32_t eid = self->n_res++;
	r[eid] = (root_t){ .plen = (uint32_t)OFS(0), .lid = iid };      /* mm_res_t { score, iid } aliases the root array */
	uint32_t rsid = ls[lid].rsid;
	om_seed_t const *p = &s[rsid];
	/* QUIRK: mm_search_load_pos runs before mm_init_ref, i.e. with the rlen of the previously loaded reference
	 * (0 for the very first chain of the run): the state is carried across chains *and reads* (minialign.c:3864,3873) */
	pos_pair_t cp = mm_search_load_pos(self, p, &st->rev);
	st->cp = cp; st->tp = cp;
	st->aid = p->rid; st->bid = self->qid;
	st->iid = iid; st->eid = eid; st->sid = rsid;
	st->prem = plen; st->pacc = 0;
	st->srem = MM_SREM; st->narrow = 0;
	init_ref(self, st->aid);
	return 0;
}
/* mm_search_load_next, minialign.c:3888-3946 */
static uint64_t mm_search_load_next(om_align_t *self, search_t *st)
{
	if(st->srem == 0) { return 0; }
	st->srem--;
	om_seed_t const *s = self->seed.a;
	v2u32_t *n = self->next.a;
	uint64_t ncnt = self->next.n, ofs = 2 * (uint64_t)self->tglen;
	v4 tv = window((int32_t)self->tglen), ev = window(128);
	int32_t fa = (int32_t)st->cp.apos, fb = (int32_t)(st->cp.bpos - (st->rev ? self->qlen : 0));
	v4 fv = { { (int32_t)U_(fa, fb), (int32_t)st->aid, (int32_t)V_(fa, fb), (int32_t)V_(fa, fb) } };   /* _posv_v4i32 */
	uint64_t plim = ofs - st->pacc;
	if(st->pacc > ofs) { ncnt = 0; }
	for(uint64_t i = 0; i < ncnt; i++) {
		if(n[i].u32[0] >= plim) { ncnt = i; break; }
		n[i].u32[0] += st->pacc;
	}
	uint64_t sid = st->sid;
	for(uint64_t rcnt = 2 * (uint64_t)st->srem; sid > 0 && rcnt > 0; sid--) {
		v4 wv = add4(load_pv(&s[sid - 1]), tv);
		v4 zv = add4(load_pv(&s[sid - 1]), ev);
		if(!INSIDE_UUB(wv, fv)) { break; }
		if(!INSIDE_WV(wv, fv) || INSIDE_WV(zv, fv)) { continue; }
		n[ncnt++] = (v2u32_t){ .u32 = { (uint32_t)pdiff(wv, fv), (uint32_t)(sid - 1) } }; rcnt--;
	}
	st->sid = (uint32_t)sid;
	self->next.n = ncnt;
	if(ncnt == 0) { st->pacc = 0; st->srem = 0; return 0; }
	radix_sort_64x(n, ncnt);
	uint32_t nsid = n[--self->next.n].u32[1];
	st->pacc = (uint32_t)(ofs - n[self->next.n].u32[0]);
	st->cp = mm_search_load_pos(self, &s[nsid], &st->rev);
	return st->srem;
}
/* mm_search_test_dup, minialign.c:3953-3982 */
static uint64_t mm_search_test_dup(om_align_t *self, search_t *st, og_pos_pair_t const *cp)
{
	uint64_t k = KEY((uint64_t)cp->apos | ((uint64_t)cp->bpos << 32), (uint64_t)st->aid | ((uint64_t)st->bid << 32));
	uint64_t *t = kh_put_ptr(&self->pos, k, 1);
	uint64_t prev = *t;
	int32_t pa = MAX2(1, MIN2((int32_t)cp->apos, (int32_t)self->rlen)), pb = MAX2(1, MIN2((int32_t)cp->bpos, (int32_t)self->qlen));
	st->tp.apos = (uint32_t)pa; st->tp.bpos = (uint32_t)pb;
	*t = (uint64_t)st->eid | ((uint64_t)UINT32_MAX << 32);
	if(prev == KH_INIT_VAL) { return 0; }
	uint32_t eid = (uint32_t)*t;                 /* QUIRK: reads the slot *after* overwriting it, so eid == st->eid always */
	root_t *r = self->root.a;
	if(eid != st->eid && cp->plen < bin_at(self, r[eid].lid)->plen) {
		st->srem = 0;
	} else {
		st->narrow = MIN2(st->narrow + 1, 2);
	}
	return 1;
}
/* mm_search_record (+ mm_update_pos), minialign.c:3987-4067 */
static uint64_t mm_search_record(om_align_t *self, search_t *st, og_alignment_t *a)
{
	og_segment_t const *sl = &a->seg[a->slen - 1], *s0 = &a->seg[0];
	uint32_t p[4];
	p[0] = self->rlen - (sl->apos + sl->alen); p[1] = self->qlen - (sl->bpos + sl->blen);
	p[2] = self->rlen - s0->apos;               p[3] = self->qlen - s0->bpos;
	st->cp.apos = p[0]; st->cp.bpos = p[1];
	st->prem -= a->plen; st->pacc = a->plen;

	uint64_t id = (uint64_t)st->aid | ((uint64_t)st->bid << 32);
	uint64_t hk = KEY((uint64_t)p[0] | ((uint64_t)p[1] << 32), id), tk = KEY((uint64_t)p[2] | ((uint64_t)p[3] << 32), id);
	/* QUIRK: h is taken before the second insert, which may shift table entries under it (minialign.c:4027-4029);
	 * both are plain pointers into the same array here too, so the aliasing is reproduced literally */
	uint64_t *h = kh_put_ptr(&self->pos, hk, 1);
	uint64_t *t = kh_put_ptr(&self->pos, tk, 0);
	uint64_t new = (uint32_t)(*h >> 32) == UINT32_MAX;
	vec_push(og_alignment_t *, self->alns, a);
	uint64_t handle = self->alns.n;              /* index + 1 */
	uint32_t nid;
	if(new) { vec_reserve(uint64_t, self->bin, self->bin.n + 1); nid = (uint32_t)self->bin.n; self->bin.a[self->bin.n++] = handle; }
	else { nid = (uint32_t)(*h >> 32); }
	root_t *r = self->root.a;
	bin_hdr_t *bin = bin_at(self, st->iid);
	uint32_t ovl = MAX2(bin->lb, p[1]) - MIN2(bin->ub, p[3]) - p[1] + p[3];
	r[st->eid].plen -= (uint32_t)(a->score + (int64_t)d2u32((double)(uint32_t)(ovl * 2) * a->identity));
	bin->n_aln += (uint32_t)new;
	bin->plen += a->plen;
	bin->lb = MIN2(bin->lb, p[1]);
	bin->ub = MAX2(bin->ub, p[3]);
	og_alignment_t *b0 = aln_of(self, nid);
	if(b0->score > a->score) {
		*t = (uint64_t)st->eid | ((uint64_t)UINT32_MAX << 32);
	} else {
		if(b0 != a) { self->bin.a[nid] = handle; }
		*h = *t = (uint64_t)st->eid | ((uint64_t)nid << 32);
	}
	if(getenv("OM_DEBUG_REC")) { fprintf(stderr, "rec eid %u iid %u aid %u p %u %u %u %u score %ld plen %u new %u n_bin %u kh %u/%u\n", st->eid, st->iid, st->aid, p[0], p[1], p[2], p[3], (long)a->score, (uint32_t)a->plen, (uint32_t)new, (uint32_t)self->bin.n - (uint32_t)new, self->pos.cnt, self->pos.mask); }
	st->srem = MM_SREM; st->narrow = 0;
	{
		float cand = (float)a->score * self->min_ratio, cur = (float)st->min_score;
		st->min_score = f2u32(cur > cand ? cur : cand);
	}
	return (new && st->prem > 0) ? 0 : 1;
}
/* mm_extend_core, minialign.c:4075-4112 */
static og_fill_t const *mm_extend_core(om_align_t *self, int bw_idx, og_section_t const *a, og_section_t const *at, og_section_t const *b, og_section_t const *bt, pos_pair_t s)
{
	if(getenv("OM_DEBUG")) { fprintf(stderr, "fill_root bw(%d) a(%u,%u) apos(%u) b(%u,%u) bpos(%u) rlen(%u) qlen(%u)\n", bw_idx, a->id, a->len, s.apos, b->id, b->len, s.bpos, self->rlen, self->qlen); }
	og_fill_t const *f = og_dp_fill_root(self->dp, bw_idx, a, s.apos, b, s.bpos, 0);
	og_fill_t const *m = f;
	uint32_t flag = OG_TERM;
	self->cnt[3]++;
	while((flag & f->status) == 0) {
		if(f->status & OG_UPDATE_A) { a = at; }
		if(f->status & OG_UPDATE_B) { b = bt; }
		flag |= f->status & (OG_UPDATE_A | OG_UPDATE_B);
		f = og_dp_fill(self->dp, f, a, b, 0);
		m = f->max > m->max ? f : m;
	}
	return m;
}
/* analysis hook (tools/trial_stats.py): OM_DUMP_TRIALS=<file> -> one line per extension trial: read ordinal, round, chain, trial of the chain, ns of the downward pass + maximum
 * search, ns of the upward pass + traceback, outcome (z = maximum 0, d = duplicate, s = score too low / no path, r = recorded, R = recorded and the walk ends) */
static FILE *om_trial_dump = NULL; static uint64_t om_trial_read = 0, om_trial_round = 0;
static inline uint64_t om_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
#define OM_TRIAL(_k, _t, _d, _u, _o) { if(om_trial_dump) { fprintf(om_trial_dump, "%lu\t%lu\t%lu\t%lu\t%lu\t%lu\t%c\n", (unsigned long)om_trial_read, (unsigned long)om_trial_round, (unsigned long)(_k), (unsigned long)(_t), (unsigned long)(_d), (unsigned long)(_u), (_o)); } }
/* mm_extend, minialign.c:4118-4173 */
static uint64_t mm_extend(om_align_t *self)
{
	search_t st; memset(&st, 0, sizeof(st)); st.crem = MM_CREM; st.min_score = self->min_score;
	for(uint64_t k = 0; k < self->root.n; k++) {
		if(mm_search_load_root(self, &st, (uint32_t)k)) { if(getenv("OM_DEBUG")) fprintf(stderr, "chain %lu: plen too short, break\n", k); break; }
		if(getenv("OM_DEBUG")) fprintf(stderr, "chain %lu/%lu: prem %ld aid %u cp(%u,%u) rev %u\n", k, self->root.n, st.prem, st.aid, st.cp.apos, st.cp.bpos, st.rev);
		uint64_t trial = 0;
		for(; st.srem > 0 && st.prem > 0; mm_search_load_next(self, &st), trial++) {
			const uint64_t t0 = om_trial_dump ? om_ns() : 0; uint64_t t1 = 0;
			og_dp_flush(self->dp);
			/* QUIRK (minialign.c:4123): _dp(x) ignores its argument, every call uses dp[st.narrow] */
			og_fill_t const *f = mm_extend_core(self, (int)st.narrow, &self->r[0], self->rtp, &self->q[st.rev], &self->t[0], st.cp);
			if(getenv("OM_DEBUG")) fprintf(stderr, "  down max %ld\n", f->max);
			if(f->max == 0) { if(om_trial_dump) { OM_TRIAL(k, trial, om_ns() - t0, 0, 'z'); } continue; }
			og_pos_pair_t const *pp = og_dp_search_max(self->dp, f);
			if(om_trial_dump) { t1 = om_ns(); }
			if(mm_search_test_dup(self, &st, pp) != 0) { if(getenv("OM_DEBUG")) fprintf(stderr, "  dup\n"); OM_TRIAL(k, trial, t1 - t0, 0, 'd'); continue; }
			pos_pair_t up = { .apos = self->r[0].len - st.tp.apos, .bpos = self->q[0].len - st.tp.bpos };
			f = mm_extend_core(self, (int)st.narrow, &self->r[1], self->rtp + 1, &self->q[1 - st.rev], &self->t[0], up);
			og_alignment_t *a = NULL;
			if(getenv("OM_DEBUG")) fprintf(stderr, "  up max %ld (from %u,%u)\n", f->max, up.apos, up.bpos);
			if(getenv("OM_DEBUG_REC")) { fprintf(stderr, "up eid %u max %ld tp %u %u\n", st.eid, (long)f->max, st.tp.apos, st.tp.bpos); }
			if(f->max < (int64_t)self->min_score || (a = og_dp_trace(self->dp, f)) == NULL) { if(getenv("OM_DEBUG_REC")) { fprintf(stderr, "  rejected (%s)\n", f->max < (int64_t)self->min_score ? "score" : "trace"); } if(om_trial_dump) { OM_TRIAL(k, trial, t1 - t0, om_ns() - t1, 's'); } continue; }
			self->cnt[4]++;
			const uint64_t t2 = om_trial_dump ? om_ns() : 0;
			if(mm_search_record(self, &st, a)) { OM_TRIAL(k, trial, t1 - t0, t2 - t1, 'R'); break; }
			OM_TRIAL(k, trial, t1 - t0, t2 - t1, 'r');
		}
		if(getenv("OM_DEBUG")) fprintf(stderr, "  finish: n_aln %u score %d min_score %u\n", bin_at(self, st.iid)->n_aln, OFS(self->root.a[st.eid].plen), self->min_score);
		if(mm_finish_root(self, &st)) { break; }
	}
	return self->n_res;
}

/* ---- post-map (minialign.c:4175-4398) ---- */
#define MAPQ_DEC    4
#define MAPQ_COEF   ( 1 << MAPQ_DEC )
#define CLIP(x)     MAX2(0, MIN2((uint32_t)d2u32(x), 60 * MAPQ_COEF))
/* mm_prune_regs, minialign.c:4185-4208 */
static uint64_t mm_prune_regs(om_align_t *self)
{
	root_t *res = self->root.a;
	uint64_t q = self->n_res;
	uint32_t min = (uint32_t)OFS(f2u32((float)OFS(res[0].plen) * self->min_ratio));
	while(res[--q].plen > min) { }
	self->n_res = (uint32_t)(q + 1);
	return q + 1;
}
/* mm_collect_supp, minialign.c:4214-4264 */
static uint64_t mm_collect_supp(om_align_t *self, uint32_t n_res, root_t *res)
{
	#define SWAP_RES(x, y)  { root_t _tmp = res[x]; res[x] = res[y]; res[y] = _tmp; }
	uint64_t p, q;
	for(p = 1, q = n_res; p < q; p++) {
		uint64_t max = 0;
		for(uint64_t i = p; i < q; i++) {
			bin_hdr_t *s = bin_at(self, res[i].lid);
			int64_t lb = s->lb, ub = s->ub, span = ub - lb;
			int covered = 0;
			for(uint64_t j = 0; j < p; j++) {
				bin_hdr_t *t = bin_at(self, res[j].lid);
				if((int64_t)t->ub < ub) { lb = MAX2(lb, (int64_t)t->ub); } else { ub = MIN2(ub, (int64_t)t->lb); }
				if(1.2 * (double)(ub - lb) < (double)span) { q--; SWAP_RES(i, q); i--; covered = 1; break; }
			}
			if(covered) { continue; }
			max = MAX2(max, ((uint64_t)(2 * (ub - lb) - span) << 32) | i);
		}
		if(max & 0xffffffff) { SWAP_RES(p, max & 0xffffffff); }
	}
	p = MIN2(p, q);
	#undef SWAP_RES
	return p;
}
/* mm_post_map, minialign.c:4270-4326 */
static uint64_t mm_post_map(om_align_t *self)
{
	root_t *res = self->root.a;
	uint64_t p = mm_collect_supp(self, self->n_res, res);
	int64_t usc = 0, lsc = INT64_MAX, tsc = 0;
	for(uint64_t i = p; i < self->n_res; i++) {
		usc = MAX2(usc, (int64_t)OFS(res[i].plen));
		lsc = MIN2(lsc, (int64_t)OFS(res[i].plen));
		tsc += OFS(res[i].plen);
	}
	lsc = (lsc == INT32_MAX) ? 0 : lsc;          /* QUIRK: compares an INT64_MAX-initialised value with INT32_MAX */
	double tpc = 1.0, x = self->xcoef, mx = self->mcoef + self->xcoef;
	for(uint64_t i = 0; i < p; i++) {
		uint32_t score = (uint32_t)OFS(res[i].plen);
		bin_hdr_t *bin = bin_at(self, res[i].lid);
		double pid = 0.0; uint64_t len = 0;
		for(uint64_t j = 0; j < bin->n_aln; j++) {
			og_alignment_t *al = aln_of(self, res[i].lid + BIN_N + j);
			len += al->plen; pid += (double)al->plen * al->identity;
		}
		pid /= (double)len;
		double ec = 2.0 / (pid * mx - x);
		double ulen = ec * (double)MAX2((int64_t)score - usc, 0), pe = 1.0 / (ulen * ulen + 1);
		bin->plen = CLIP(-10.0 * MAPQ_COEF * log10(pe));
		tpc *= 1.0 - pe;
	}
	double tpe = MIN2(1.0 - tpc, 1.0);
	for(uint64_t i = p; i < self->n_res; i++) {
		bin_hdr_t *bin = bin_at(self, res[i].lid);
		bin->plen = CLIP(-10.0 * MAPQ_COEF * log10(1.0 - tpe * (double)(int64_t)((int64_t)res[i].plen - lsc + 1) / (double)tsc));
	}
	return p;
}

static void tbuf_clear(om_align_t *self)
{
	self->resc.n = 0; self->presc = 0; self->seed.n = 0; self->n_seed = 0; self->root.n = 0; self->next.n = 0;
	self->n_res = 0; self->bin.n = 0; self->alns.n = 0;
	kh_clear(&self->pos);
}

/* mm_align_seq, minialign.c:4427-4474 */
om_reg_t *om_align_seq(om_align_t *self, uint32_t l_seq, uint8_t const *seq)
{
	if(l_seq < self->mi->k || l_seq * self->mcoef < (double)self->min_score) { return NULL; }
	tbuf_clear(self);
	init_query(self, l_seq, seq);
	{ static int asked = 0; if(!asked) { asked = 1; const char *fn = getenv("OM_DUMP_TRIALS"); if(fn) { om_trial_dump = fopen(fn, "w"); } } }
	om_trial_read++;
	for(uint64_t i = 0; i < self->mi->n_occ; i++) {
		om_trial_round = i;
		if(mm_seed(self, i) == 0) { continue; }
		if(mm_chain(self) == 0) { continue; }
		if(mm_extend(self) > 0) { break; }
	}
	if(self->n_res == 0) {
		for(uint64_t i = 0; i < self->alns.n; i++) { og_aln_free(self->alns.a[i]); }
		return NULL;
	}
	radix_sort_64x((v2u32_t *)self->root.a, self->n_res);
	uint32_t n_all = (uint32_t)mm_prune_regs(self);
	uint32_t n_uniq = (uint32_t)mm_post_map(self);
	/* mm_pack_reg, minialign.c:4364-4397 */
	om_reg_t *reg = (om_reg_t *)calloc(1, sizeof(om_reg_t));
	reg->aln = (om_aln_t *)calloc(self->bin.n + 1, sizeof(om_aln_t));
	uint8_t *used = (uint8_t *)calloc(self->alns.n + 1, 1);
	uint32_t np = 0;
	root_t *res = self->root.a;
	for(uint64_t i = 0; i < n_all; i++) {
		bin_hdr_t *bin = bin_at(self, res[i].lid);
		for(uint64_t j = 0; j < bin->n_aln; j++) {
			uint64_t h = self->bin.a[res[i].lid + BIN_N + j];
			reg->aln[np].aid = (uint32_t)i; reg->aln[np].mapq = bin->plen; reg->aln[np].a = self->alns.a[h - 1];
			used[h - 1] = 1; np++;
		}
		if(i == n_uniq - 1) { reg->n_uniq = np; }
	}
	reg->n_all = np;
	for(uint64_t i = 0; i < self->alns.n; i++) { if(!used[i]) { og_aln_free(self->alns.a[i]); } }
	free(used);
	return reg;
}
void om_reg_free(om_reg_t *r)
{
	if(!r) { return; }
	for(uint32_t i = 0; i < r->n_all; i++) { og_aln_free(r->aln[i].a); }
	free(r->aln); free(r);
}

/* stage taps */
uint64_t om_stage_seed(om_align_t *self, uint32_t l_seq, uint8_t const *seq, uint64_t iter, om_seed_t const **seeds)
{
	tbuf_clear(self); init_query(self, l_seq, seq);
	uint64_t n = 0;
	for(uint64_t i = 0; i <= iter; i++) { n = mm_seed(self, i); }
	*seeds = self->seed.a;
	return n;
}
uint64_t om_stage_chain(om_align_t *self, uint64_t const **roots)
{
	uint64_t n = mm_chain(self);
	*roots = (uint64_t const *)self->root.a;
	return n;
}

/* ---- SAM (minialign.c:5096-5426), default tag set ---- */
void om_sam_header(FILE *fp, om_opt_t const *o, om_seq_t const *ref, uint32_t n_ref)
{
	fputs("@HD\tVN:1.0\tSO:unsorted\n", fp);
	for(uint32_t i = 0; i < n_ref; i++) { fprintf(fp, "@SQ\tSN:%.*s\tLN:%u\n", (int)ref[i].l_name, ref[i].name, ref[i].l_seq); }
	if(((o->flag | o->tags) & (1ULL << OM_RG)) && o->rg_line) { fprintf(fp, "%s\n", o->rg_line); }        /* minialign.c:5111 */
	fprintf(fp, "@PG\tID:minialign\tPN:minialign\tVN:%s\tCL:%s\n", "0.6.0-devel", o->arg_line ? o->arg_line : "");
}
static void put_seq(FILE *fp, uint8_t const *s, uint32_t n, int rev)
{
	for(uint32_t i = 0; i < n; i++) {
		uint8_t c = rev ? s[n - 1 - i] : s[i];
		fputc(rev ? "TGCAN\0\0\0\0\0\0\0\0\0\0\0"[c & 15] : "ACGTN\0\0\0\0\0\0\0\0\0\0\0"[c & 15], fp);    /* decaf / decar, minialign.c:231-232 */
	}
}
/* mm_print_sam_mapped_core, minialign.c:5147-5198 */
static void sam_core(FILE *fp, om_seq_t const *r, om_seq_t const *q, og_segment_t const *s, uint32_t const *path, uint32_t flag, uint32_t mapq)
{
	uint32_t rid = s->aid >> 1;
	uint32_t rs = r[rid].l_seq - s->apos - s->alen;
	uint32_t hl = q->l_seq - s->bpos - s->blen, tl = s->bpos;
	uint32_t qs = (flag & 0x900) ? hl : 0;
	uint32_t qe = q->l_seq - ((flag & 0x900) ? tl : 0);
	fprintf(fp, "%.*s\t%u\t%.*s\t%u\t%u\t", (int)q->l_name, q->name, flag | ((~s->bid & 0x01) << 4), (int)r[rid].l_name, r[rid].name, rs + 1, mapq >> MAPQ_DEC);
	if(hl) { fprintf(fp, "%u%c", hl, (flag & 0x900) ? 'H' : 'S'); }
	uint64_t plen = (uint64_t)s->alen + s->blen;
	char *buf = (char *)malloc(plen * 3 + 64);
	og_dump_cigar_reverse(buf, plen * 3 + 64, path, s->ppos, plen);
	fputs(buf, fp); free(buf);
	if(tl) { fprintf(fp, "%u%c", tl, (flag & 0x900) ? 'H' : 'S'); }
	fputs("\t*\t0\t0\t", fp);
	if(s->bid & 0x01) { put_seq(fp, &q->seq[qs], qe - qs, 0); }
	else { put_seq(fp, &q->seq[q->l_seq - qe], qe - qs, 1); }
	fputc('\t', fp);
	if(q->qual && q->qual[0] != '\0') {      /* kept with -Q only (minialign.c:5186-5195, 5964) */
		if(s->bid & 0x01) { fwrite(&q->qual[qs], 1, qe - qs, fp); }
		else { for(uint32_t i = 0; i < qe - qs; i++) { fputc(q->qual[q->l_seq - qe + (qe - qs - 1 - i)], fp); } }
	} else { fputc('*', fp); }
}
/* mm_print_sam_supp, minialign.c:5204-5236: one "rname,pos,strand,CIGAR,mapQ,NM;" entry.  QUIRKS kept: the name is always the first reference
 * sequence's (r->name, not r[rid].name), and the mapping quality is the raw 16x fixed-point value */
static void sam_supp(FILE *fp, om_seq_t const *r, om_seq_t const *q, uint32_t const *path, og_segment_t const *s, uint32_t ed, uint32_t mapq)
{
	uint32_t rid = s->aid >> 1;
	uint32_t rs = r[rid].l_seq - s->apos - s->alen;
	uint32_t hl = q->l_seq - s->bpos - s->blen, tl = s->bpos;
	fprintf(fp, "%.*s,%u,%c,", (int)r[0].l_name, r[0].name, rs + 1, (s->bid & 0x01) ? '+' : '-');
	if(hl != 0) { fprintf(fp, "%uH", hl); }
	uint64_t plen = (uint64_t)s->alen + s->blen;
	char *buf = (char *)malloc(plen * 3 + 64);
	og_dump_cigar_reverse(buf, plen * 3 + 64, path, s->ppos, plen);
	fputs(buf, fp); free(buf);
	if(tl != 0) { fprintf(fp, "%uH", tl); }
	fprintf(fp, ",%u,%u;", mapq, ed);
}
/* mm_print_sam_md, minialign.c:5243-5301: reference bases at mismatches, ^ + bases at deletions, match counts between them */
typedef struct { FILE *fp; uint8_t const *rp, *rb, *qp; int rev; } md_ctx_t;
static void md_step(void *ctx, char op, uint64_t c)
{
	md_ctx_t *m = (md_ctx_t *)ctx;
	if(op == 'D') {
		fprintf(m->fp, "%lu^", (unsigned long)(m->rp - m->rb)); m->rb = m->rp + c;
		for(uint64_t i = 0; i < c; i++) { fputc("ACGTN\0\0\0\0\0\0\0\0\0\0\0"[*m->rp++ & 15], m->fp); }
	} else if(op == 'I') {
		if(m->rev) { m->qp -= c; } else { m->qp += c; }
	} else {
		for(uint64_t t = 0; t < c; t++) {
			uint8_t rc = m->rp[t], qc = m->rev ? (uint8_t)(0x03 ^ m->qp[-1 - (int64_t)t]) : m->qp[t];     /* reverse strand: complement by xor 3, so N (4) never equals N */
			if(rc != qc) { fprintf(m->fp, "%lu%c", (unsigned long)(&m->rp[t] - m->rb), "ACGTN\0\0\0\0\0\0\0\0\0\0\0"[rc & 15]); m->rb = &m->rp[t] + 1; }
		}
		m->rp += c; if(m->rev) { m->qp -= c; } else { m->qp += c; }
	}
}
static void sam_md(FILE *fp, om_seq_t const *r, om_seq_t const *q, uint32_t const *path, og_segment_t const *s)
{
	fputs("\tMD:Z:", fp);
	uint32_t rev = ~s->bid & 0x01, rid = s->aid >> 1;
	md_ctx_t m; m.fp = fp; m.rev = (int)rev;
	m.rp = m.rb = &r[rid].seq[r[rid].l_seq - s->apos - s->alen];
	m.qp = rev ? &q->seq[q->l_seq - s->bpos] : &q->seq[q->l_seq - s->bpos - s->blen];
	og_parse_path_reverse(path, s->ppos, (uint64_t)s->alen + s->blen, md_step, &m);
	fprintf(fp, "%lu", (unsigned long)(m.rp - m.rb));
}
/* mm_print_sam_mapped, minialign.c:5390-5426 (+ _unmapped :5127, _general_tags :5304, _primary_tags :5347) */
void om_sam_record_opt(FILE *fp, om_opt_t const *o, om_seq_t const *ref, om_seq_t const *q, om_reg_t const *reg)
{
	uint64_t const f = o ? (o->flag | o->tags) : 0;         /* one word for flags and tag bits (minialign.c:5677) */
	#define TAG(_x)     ( (f >> (_x)) & 1 )
	if(reg == NULL) {
		fprintf(fp, "%.*s\t4\t*\t0\t0\t*\t*\t0\t0\t", (int)q->l_name, q->name);
		put_seq(fp, q->seq, q->l_seq, 0);
		fputc('\t', fp);
		if(q->qual && q->qual[0] != '\0') { fwrite(q->qual, 1, q->l_seq, fp); } else { fputc('*', fp); }
		if(q->comment) { fprintf(fp, "\tCO:Z:%s", q->comment); }
		fputc('\n', fp);
		return;
	}
	uint64_t n = (f & 0x08) ? reg->n_uniq : reg->n_all;      /* MM_OMIT_REP */
	uint32_t flag = 0;
	for(uint64_t i = 0; i < n; i++) {
		if(i >= reg->n_uniq) { flag = 0x100; }
		om_aln_t const *a = &reg->aln[i];
		for(uint64_t j = a->a->slen; j > 0; j--) {
			sam_core(fp, ref, q, &a->a->seg[j - 1], a->a->path, flag, a->mapq);
			if(TAG(OM_RG)) { fprintf(fp, "\tRG:Z:%s", o->rg_id); }
			if(TAG(OM_NH)) { fprintf(fp, "\tNH:i:%u", reg->n_all); }
			if(TAG(OM_IH)) { fprintf(fp, "\tIH:i:%lu", (unsigned long)i); }
			if(TAG(OM_AS)) { fprintf(fp, "\tAS:i:%ld", (long)a->a->score); }
			if(TAG(OM_NM)) { uint32_t xcnt = (uint32_t)((double)a->a->dcnt * (1.0 - a->a->identity)); fprintf(fp, "\tNM:i:%u", xcnt + a->a->agcnt + a->a->bgcnt); }
			if(TAG(OM_MD)) { sam_md(fp, ref, q, a->a->path, &a->a->seg[j - 1]); }
			if(i == 0 && j == a->a->slen) {
				flag = 0x800;
				uint64_t stop = 0;
				if(TAG(OM_XS)) { fprintf(fp, "\tXS:i:%ld", reg->n_all > 1 ? (long)reg->aln[1].a->score : 0L); }
				if(TAG(OM_SA) && (reg->n_uniq > 1 || reg->aln[0].a->slen > 1)) {
					fputs("\tSA:Z:", fp);
					for(uint64_t x = 0; x < reg->n_uniq; x++) {
						om_aln_t const *b = &reg->aln[x];
						uint32_t ed = (uint32_t)((double)b->a->dcnt * (1.0 - b->a->identity)) + b->a->agcnt + b->a->bgcnt;
						for(uint64_t y = b->a->slen; y > 0; y--) {
							if(x == 0 && y == b->a->slen) { continue; }
							sam_supp(fp, ref, q, b->a->path, &b->a->seg[y - 1], ed, b->mapq);
						}
					}
					stop = 1;
				}
				if(q->comment) { fprintf(fp, "\tCO:Z:%s", q->comment); }
				if(stop) { i = n; j = 1; }                        /* the other records are in the SA tag (minialign.c:5418-5420) */
			}
			fputc('\n', fp);
		}
		flag = 0x800;
	}
	#undef TAG
}
void om_sam_record(FILE *fp, om_seq_t const *ref, om_seq_t const *q, om_reg_t const *reg) { om_sam_record_opt(fp, NULL, ref, q, reg); }

/* ---- the other output formats (minialign.c:5427-5625) ---- */
static void put_fixed(FILE *fp, uint32_t n, int c)          /* _putfi, minialign.c:4812: n with a decimal point in front of its last c digits */
{
	char d[24]; int i = 0;
	while(n || i <= c) { d[i++] = (char)('0' + n % 10); n /= 10; }
	for(int j = i; j > c; j--) { fputc(d[j - 1], fp); }
	fputc('.', fp);
	for(int j = c; j > 0; j--) { fputc(d[j - 1], fp); }
}
static void put_pair(char *b1, char *b2, uint32_t n1, uint32_t n2, int *l)      /* _putpi, minialign.c:4847: two numbers right-aligned to one width */
{
	char d1[16], d2[16]; int i = 0;
	uint32_t m1 = n1, m2 = n2;
	while(m1 | m2) { d1[i] = (char)(m1 % 10); d2[i] = (char)(m2 % 10); m1 /= 10; m2 /= 10; i++; }
	if(i == 0) { d1[0] = d2[0] = 0; i = 1; }
	int z1 = 0, z2 = 0;
	for(int j = i; j > 0; j--) {
		z1 |= d1[j - 1] | (j == 1); z2 |= d2[j - 1] | (j == 1);
		*b1++ = (char)(d1[j - 1] + '0' - (z1 ? 0 : 0x10)); *b2++ = (char)(d2[j - 1] + '0' - (z2 ? 0 : 0x10));
	}
	*l = i;
}
static void maf_core(FILE *fp, om_seq_t const *r, om_seq_t const *q, uint32_t const *path, og_segment_t const *s, int64_t score)     /* mm_print_maf_mapped_core, :5427 */
{
	fprintf(fp, "a score=%u\n", (uint32_t)score);
	uint32_t rid = s->aid >> 1;
	uint32_t const rs = r[rid].l_seq - s->apos - s->alen, qs = q->l_seq - s->bpos - s->blen;
	uint32_t l = (r[rid].l_name > q->l_name ? r[rid].l_name : q->l_name) + 1;
	char n1[3][16], n2[3][16]; int w[3];
	put_pair(n1[0], n2[0], rs, qs, &w[0]); put_pair(n1[1], n2[1], s->alen, s->blen, &w[1]); put_pair(n1[2], n2[2], r[rid].l_seq, q->l_seq, &w[2]);
	uint64_t plen = (uint64_t)s->alen + s->blen;
	char *buf = (char *)malloc(plen + 64);
	fprintf(fp, "s %.*s%*s%.*s %.*s + %.*s ", (int)r[rid].l_name, r[rid].name, (int)(l - r[rid].l_name), "", w[0], n1[0], w[1], n1[1], w[2], n1[2]);
	og_dump_seq_reverse(buf, plen + 64, OG_SEQ_A, path, s->ppos, plen, &r[rid].seq[rs], '-');
	fprintf(fp, "%s\n", buf);
	fprintf(fp, "s %.*s%*s%.*s %.*s %c %.*s ", (int)q->l_name, q->name, (int)(l - q->l_name), "", w[0], n2[0], w[1], n2[1], (s->bid & 0x01) ? '+' : '-', w[2], n2[2]);
	og_dump_seq_reverse(buf, plen + 64, OG_SEQ_B | ((s->bid & 0x01) ? OG_SEQ_FW : OG_SEQ_RV), path, s->ppos, plen, (s->bid & 0x01) ? &q->seq[qs] : &q->seq[q->l_seq - qs], '-');
	fprintf(fp, "%s\n\n", buf);
	free(buf);
}
void om_print_record(FILE *fp, om_opt_t const *o, om_seq_t const *r, om_seq_t const *q, om_reg_t const *reg)
{
	if(o->format == 0) { om_sam_record_opt(fp, o, r, q, reg); return; }
	if(reg == NULL) { return; }
	uint64_t const f = o->flag | o->tags;
	uint64_t const n = (f & 0x08) ? reg->n_uniq : reg->n_all;
	for(uint64_t i = 0; i < n; i++) {
		om_aln_t const *a = &reg->aln[i];
		og_segment_t const *s = &a->a->seg[a->a->slen - 1], *e = &a->a->seg[0];
		uint32_t rid = s->aid >> 1;
		uint32_t dcnt = a->a->dcnt, mcnt = d2u32((double)dcnt * a->a->identity), gcnt = a->a->agcnt + a->a->bgcnt;
		if(o->format == 1) {
			for(uint64_t j = a->a->slen; j > 0; j--) { maf_core(fp, r, q, a->a->path, &a->a->seg[j - 1], a->a->score); }
		} else if(o->format == 2) {          /* mm_print_blast6_mapped, :5497: qname rname idt len #x #gap qs qe rs re e-value bitscore */
			uint32_t rs = (s->bid & 0x01) ? r[rid].l_seq - s->apos - s->alen + 1 : r[rid].l_seq - e->apos;
			uint32_t re = (s->bid & 0x01) ? r[rid].l_seq - e->apos : r[rid].l_seq - s->apos - s->alen + 1;
			uint32_t qs = q->l_seq - s->bpos - s->blen + 1, qe = q->l_seq - e->bpos;
			fprintf(fp, "%.*s\t%.*s\t", (int)q->l_name, q->name, (int)r[rid].l_name, r[rid].name);
			put_fixed(fp, d2u32(1000.0 * a->a->identity), 3);
			fprintf(fp, "\t%u\t%u\t%u\t%u\t%u\t%u\t%u\t", dcnt + gcnt, dcnt - mcnt, gcnt, qs, qe, rs, re);
			double bit = 1.85 * (double)a->a->score - 0.02;
			put_fixed(fp, d2u32(1000.0 * (double)r[rid].l_seq * (double)q->l_seq * pow(2.0, -bit)), 3);
			fprintf(fp, "\t%u\n", d2u32(bit));
		} else {                              /* mm_print_paf_mapped, :5549: qname ql qs qe strand rname rl rs re #match block_len mapq [tags] */
			uint32_t const rs = r[rid].l_seq - s->apos - s->alen, re = r[rid].l_seq - e->apos;
			uint32_t const qs = q->l_seq - s->bpos - s->blen, qe = q->l_seq - e->bpos;
			fprintf(fp, "%.*s\t%u\t%u\t%u\t%c\t%.*s\t%u\t%u\t%u\t%u\t%u\t%u", (int)q->l_name, q->name, q->l_seq, qs, qe, (s->bid & 0x01) ? '+' : '-',
				(int)r[rid].l_name, r[rid].name, r[rid].l_seq, rs, re, mcnt, dcnt + gcnt, a->mapq >> MAPQ_DEC);
			if((f >> OM_AS) & 1) { fprintf(fp, "\tAS:i:%u", (uint32_t)a->a->score); }
			if((f >> OM_ID) & 1) { fputs("\tID:f:", fp); put_fixed(fp, d2u32(a->a->identity * 10000.0), 4); }
			if((f >> OM_NM) & 1) { fprintf(fp, "\tNM:i:%u", (dcnt - mcnt) + gcnt); }
			if((f >> OM_SQ) & 1) { fputs("\tSQ:i:", fp); put_seq(fp, q->seq, q->l_seq, 0); }
			if((f >> OM_CG) & 1) {
				fputs("\tCG:Z:", fp);
				char *buf = (char *)malloc((uint64_t)a->a->plen * 3 + 64);
				og_dump_cigar_reverse(buf, (uint64_t)a->a->plen * 3 + 64, a->a->path, 0, a->a->plen);
				fputs(buf, fp); free(buf);
			}
			fputc('\n', fp);
		}
	}
}

/* ---- whole program ---- */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int om_main(char const *preset, char const *ref_fn, char const *query_fn, FILE *out, char const *arg_line, double *map_seconds, uint64_t *bases)
{
	om_opt_t o;
	if(om_opt_init(&o, preset)) { return 1; }
	o.arg_line = arg_line;
	return om_main_opt(&o, ref_fn, query_fn, out, map_seconds, bases);
}
int om_main_opt(om_opt_t const *op, char const *ref_fn, char const *query_fn, FILE *out, double *map_seconds, uint64_t *bases)
{
	char const *files[2] = { ref_fn, query_fn };
	return om_main_files(op, files, 2, out, map_seconds, bases);
}
/* main_align, minialign.c:6365-6447: the first file is the reference, the others are queries -- or, with -X, every file is indexed in turn and every file
 * mapped onto it (a header per index in SAM) */
int om_main_files(om_opt_t const *op, char const *const *files, int nf, FILE *out, double *map_seconds, uint64_t *bases)
{
	om_opt_t o = *op;
	int const ava = o.ava != 0 && nf > 0;          /* the mapper's own flag word (a.flag); -R sets the same bit in the printer's only */
	int const rt = ava ? nf : 1, qh = ava ? 0 : 1;
	double tmap = 0; uint64_t nb = 0;
	for(int r = 0; r < rt; r++) {
		om_seqs_t ref = om_read_fasta(files[r]); om_seqs_drop_short(&ref, o.min_len);
		if(ref.n == 0) { return 2; }
		om_idx_t *mi = om_idx_build(&o, ref.a, (uint32_t)ref.n);
		om_align_t *al = om_align_init(&o, mi);
		if(al == NULL) { return 3; }
		if(o.format == 0) { om_sam_header(out, &o, ref.a, (uint32_t)ref.n); }          /* only SAM has a header (minialign.c:5666-5671) */
		for(int q = qh; q < nf; q++) {
			om_seqs_t qs = om_read_fasta_ex(files[q], (int)o.keep_qual, (int)(((o.flag | o.tags) >> OM_CO) & 1)); om_seqs_drop_short(&qs, o.min_len);
			/* a query file that is not in shape ends the run with exit 1 (the reference drops the block it was reading: nothing of this file is printed here) */
			if(om_read_error) { om_seqs_free(&qs); om_align_free(al); om_idx_free(mi); om_seqs_free(&ref); return 4; }
			for(uint64_t i = 0; i < qs.n; i++) {
				double t1 = now_s();
				if(getenv("OM_DEBUG_CARRY")) { fprintf(stderr, "carry\t%s\t%u\n", qs.a[i].name, al->rlen); }          /* what this read starts with: the length of the reference the reads before it loaded last (minialign.c:3864) */
				om_reg_t *reg = om_align_seq(al, qs.a[i].l_seq, qs.a[i].seq);
				tmap += now_s() - t1; nb += qs.a[i].l_seq;
				om_print_record(out, &o, ref.a, &qs.a[i], reg);
				om_reg_free(reg);
			}
			om_seqs_free(&qs);
		}
		om_align_free(al); om_idx_free(mi); om_seqs_free(&ref);
	}
	if(map_seconds) { *map_seconds = tmap; }
	if(bases) { *bases = nb; }
	return 0;
}

/* test tap: same record as mm_ref_shim.c:mmref_align */
uint32_t om_align_dump(om_align_t *self, uint8_t const *seq, uint32_t len, int64_t *out, uint32_t max)
{
	om_reg_t *reg = om_align_seq(self, len, seq);
	if(reg == NULL) { return 0; }
	uint32_t k = 0, n = reg->n_all;
	for(uint32_t i = 0; i < reg->n_all && k + 12 <= max; i++) {
		om_aln_t const *a = &reg->aln[i];
		out[k++] = a->aid; out[k++] = a->mapq; out[k++] = a->a->score; out[k++] = a->a->plen; out[k++] = a->a->slen;
		out[k++] = a->a->seg[0].aid; out[k++] = a->a->seg[0].bid; out[k++] = a->a->seg[0].apos; out[k++] = a->a->seg[0].bpos;
		out[k++] = a->a->seg[0].alen; out[k++] = a->a->seg[0].blen; out[k++] = (int64_t)(i < reg->n_uniq);
	}
	om_reg_free(reg);
	return n;
}
