/* k2s_model.c -- which ranges of the seed sort need the literal replay of the reference's unstable permutation?  (CPU analysis, no device code.)
 *
 * The reference sorts a read's seeds with an in-place MSD radix sort (ksort.h:84-131: 8-bit digits from the top, a cycle-leader permutation per level, insertion sort below 65
 * elements) whose order among EQUAL keys is whatever the permutation left, and the chaining that follows depends on that order: the device replays the permutation literally,
 * one dependent LDS round trip after the other (mm_sort_kernel), at 1 200 cycles per seed in the mix.  Claim checked here: elements with equal keys share every digit, so a range
 * (the whole array, or a bucket of a level) that holds no two equal keys ends in ONE possible order; only the ranges on the digit paths of equal-key groups need the walk.
 *
 *   selective sort: (1) any stable sort by key -> S; (2) equal neighbours in S mark tie positions; (3) from the top: replay the level's walk on the range literally, then for
 *   every bucket: more than 64 elements and a tie inside -> recurse; more than 64 and no tie -> take S's slice; 2 .. 64 with a tie -> the reference's insertion sort;
 *   2 .. 64 without -> S's slice.
 *
 * Input: the file OM_DUMP_SEEDS=<file> makes the oracle write (oracle/ora_mm.c mm_seed: one record { u64 n, n x { upos, rid, vpos, lid } } per sort call, the array as it goes in).
 * Output: every array sorted both ways and compared (must be identical), and the work counted: elements walked by the literal sort (summed over the levels that permute) against
 * elements walked by the selective one.   build: gcc -O2 -o tools/k2s_model tools/k2s_model.c */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t upos, rid, vpos, lid; } seed_t;
static inline uint64_t key(const seed_t *p) { return (uint64_t)p->upos | ((uint64_t)p->rid << 32); }

static uint64_t walked_lit, walked_sel, ins_lit, ins_sel;

static void ins_sort(seed_t *beg, seed_t *end)
{
	for(seed_t *i = beg + 1; i < end; ++i) {
		if(key(i) < key(i - 1)) { seed_t *j, tmp = *i; for(j = i; j > beg && key(&tmp) < key(j - 1); --j) { *j = *(j - 1); } *j = tmp; }
	}
}
typedef struct { seed_t *b, *e; } bkt_t;
/* one level of the reference's sort on [beg, end): histogram, bounds, the cycle-leader walk; leaves b[k] = bounds of bucket k */
static int level(seed_t *beg, seed_t *end, int s, bkt_t *b)          /* returns 0 where every element has the same digit (the device steps over such a level) */
{
	bkt_t *be = b + 256, *k;
	for(k = b; k != be; ++k) { k->b = k->e = beg; }
	for(seed_t *i = beg; i != end; ++i) { ++b[key(i) >> s & 255].e; }
	for(k = b + 1; k != be; ++k) { k->e += (k - 1)->e - beg; k->b = (k - 1)->e; }
	for(k = b; k != be;) {
		if(k->b != k->e) {
			bkt_t *l;
			if((l = b + (key(k->b) >> s & 255)) != k) {
				seed_t tmp = *k->b, swap;
				do { swap = tmp; tmp = *l->b; *l->b++ = swap; l = b + (key(&tmp) >> s & 255); } while(l != k);
				*k->b++ = tmp;
			} else { ++k->b; }
		} else { ++k; }
	}
	for(b->b = beg, k = b + 1; k != be; ++k) { k->b = (k - 1)->e; }
	for(k = b; k != be; ++k) { if(k->e - k->b == end - beg) { return 0; } }
	return 1;
}
static void lit_sort(seed_t *beg, seed_t *end, int s)
{
	bkt_t b[256]; if(level(beg, end, s, b)) { walked_lit += (uint64_t)(end - beg); }
	if(s) {
		s = s > 8 ? s - 8 : 0;
		for(int k = 0; k < 256; k++) {
			if(b[k].e - b[k].b > 64) { lit_sort(b[k].b, b[k].e, s); }
			else if(b[k].e - b[k].b > 1) { ins_sort(b[k].b, b[k].e); ins_lit += (uint64_t)(b[k].e - b[k].b); }
		}
	}
}
/* base: the array being sorted; S: the stably sorted copy; tie_pre[i] = number of tie positions in S[0 .. i) */
static void sel_sort(seed_t *base, seed_t *beg, seed_t *end, int s, const seed_t *S, const uint32_t *tie_pre)
{
	bkt_t b[256]; if(level(beg, end, s, b)) { walked_sel += (uint64_t)(end - beg); }
	if(s) {
		s = s > 8 ? s - 8 : 0;
		for(int k = 0; k < 256; k++) {
			const uint64_t n = (uint64_t)(b[k].e - b[k].b); if(n < 2) { continue; }
			const uint64_t lo = (uint64_t)(b[k].b - base), hi = lo + n;
			const int ties = tie_pre[hi] != tie_pre[lo];
			if(!ties) { memcpy(b[k].b, S + lo, n * sizeof(seed_t)); }          /* one possible order: the sorted slice */
			else if(n > 64) { sel_sort(base, b[k].b, b[k].e, s, S, tie_pre); }
			else { ins_sort(b[k].b, b[k].e); ins_sel += n; }
		}
	}
}
static int cmp_stable(const void *x, const void *y)
{
	const seed_t *a = (const seed_t *)x, *b = (const seed_t *)y;
	if(key(a) != key(b)) { return key(a) < key(b) ? -1 : 1; }
	return 0;          /* (qsort need not be stable: equal keys may come out in any order, which is exactly what the claim allows) */
}

int main(int argc, char **argv)
{
	if(argc < 2) { fprintf(stderr, "usage: k2s_model seeds.bin\n"); return 2; }
	FILE *f = fopen(argv[1], "rb"); if(!f) { perror(argv[1]); return 1; }
	uint64_t n, arrays = 0, with_ties = 0, seeds = 0, tie_elems = 0, bad = 0, top_only = 0;
	while(fread(&n, 8, 1, f) == 1) {
		seed_t *a = (seed_t *)malloc((n + 1) * sizeof(seed_t)), *l = (seed_t *)malloc((n + 1) * sizeof(seed_t)), *S = (seed_t *)malloc((n + 1) * sizeof(seed_t));
		uint32_t *tie_pre = (uint32_t *)calloc(n + 2, 4);
		if(fread(a, sizeof(seed_t), n, f) != n) { fprintf(stderr, "short record\n"); return 1; }
		memcpy(l, a, n * sizeof(seed_t)); memcpy(S, a, n * sizeof(seed_t));
		/* the reference's sort */
		if(n <= 64) { ins_sort(l, l + n); } else { lit_sort(l, l + n, 56); }
		/* the selective one */
		qsort(S, n, sizeof(seed_t), cmp_stable);
		uint64_t t = 0;
		for(uint64_t i = 0; i < n; i++) { const int tie = (i > 0 && key(&S[i]) == key(&S[i - 1])) || (i + 1 < n && key(&S[i]) == key(&S[i + 1])); tie_pre[i + 1] = tie_pre[i] + (uint32_t)tie; t += (uint64_t)tie; }
		if(n <= 64) { ins_sort(a, a + n); }
		else if(t == 0) { memcpy(a, S, n * sizeof(seed_t)); }
		else { const uint64_t w0 = walked_sel; sel_sort(a, a, a + n, 56, S, tie_pre); (void)w0; }
		if(memcmp(a, l, n * sizeof(seed_t)) != 0) { bad++; }
		arrays++; seeds += n; with_ties += t != 0; tie_elems += t; (void)top_only;
		free(a); free(l); free(S); free(tie_pre);
	}
	printf("%llu arrays, %llu seeds (%.0f per array); %llu arrays (%.1f %%) hold equal keys, %llu elements (%.2f %%) in such groups\n", (unsigned long long)arrays, (unsigned long long)seeds,
		arrays ? (double)seeds / arrays : 0.0, (unsigned long long)with_ties, arrays ? 100.0 * with_ties / arrays : 0.0, (unsigned long long)tie_elems, seeds ? 100.0 * tie_elems / seeds : 0.0);
	printf("selective sort against the reference's: %llu arrays differ%s\n", (unsigned long long)bad, bad ? "  <-- THE CLAIM IS WRONG" : " (identical everywhere)");
	printf("elements walked by the cycle-leader permutation, summed over levels: literal %llu (%.2f per seed), selective %llu (%.2f per seed): %.1f %% of the literal walk\n",
		(unsigned long long)walked_lit, seeds ? (double)walked_lit / seeds : 0.0, (unsigned long long)walked_sel, seeds ? (double)walked_sel / seeds : 0.0, walked_lit ? 100.0 * walked_sel / walked_lit : 0.0);
	printf("elements insertion-sorted: literal %llu, selective %llu\n", (unsigned long long)ins_lit, (unsigned long long)ins_sel);
	return bad != 0;
}
