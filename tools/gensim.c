/*
 * gensim.c -- seeded synthetic data generator (repo's own): a reference genome with planted
 * interspersed / tandem repeats and N runs, and PBSIM-CLR-like or ONT-like reads sampled from it.
 * Own PRNG (xoshiro256** seeded by splitmix64) so that streams are identical on every box.
 *
 *   gensim genome <seed> <total_len> <n_contigs> <repeat_frac> > ref.fa
 *   gensim genomehard <seed> <total_len> <n_contigs> [repeat_frac = 0.45] > ref.fa
 *       a reference with the repeat structure of a mammalian genome rather than a few planted families: SINE-like and LINE-like families with 10^4 .. 10^6 copies that
 *       diverge 3 - 20 % from their consensus (LINE copies 5'-truncated), two hundred middle-sized families of 10 .. 10^4 copies, segmental duplications of 10 - 200 kb
 *       at 95 - 99.5 % identity, satellite arrays (171-base units in higher-order blocks, tens of kb to megabases), microsatellites, and per contig one long N gap
 *   gensim reads  <seed> ref.fa <depth> <pacbio|ont> [fq] [len_mean len_sd] > reads.fa
 *   gensim reads  <seed> ref.fa <depth> <pacbio|ont> <fa|fq> <len_mean> <len_sd> <part> <n_parts> > reads.part.fa
 *       one of n_parts independent streams (seed + part, depth / n_parts each, read names r<part>.<i>_...): the parts are generated side by side and the
 *       read set is their concatenation in part order -- bench.py builds the 9.2 Gb set of the headline workload this way
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>

static uint64_t s[4];
static uint64_t splitmix(uint64_t *x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static void seed_rng(uint64_t sd) { for(int i = 0; i < 4; i++) s[i] = splitmix(&sd); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rnd(void) { uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
static inline double unif(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t below(uint64_t n) { return (uint64_t)(unif() * (double)n); }
static double gauss(void) { double u = unif(), v = unif(); if(u < 1e-300) u = 1e-300; return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

static void put_fasta(FILE *fp, char const *name, char const *seq, uint64_t len)
{
	fprintf(fp, ">%s\n", name);
	for(uint64_t i = 0; i < len; i += 80) { fwrite(seq + i, 1, len - i < 80 ? len - i : 80, fp); fputc('\n', fp); }
}

static int main_genome(int argc, char **argv)
{
	if(argc < 6) return 1;
	seed_rng(strtoull(argv[2], 0, 0));
	uint64_t total = strtoull(argv[3], 0, 0); int nc = atoi(argv[4]); double rf = atof(argv[5]);
	char *g = malloc(total + 1);
	for(uint64_t i = 0; i < total; i++) g[i] = "ACGT"[rnd() >> 62];
	/* contig boundaries: geometric-ish sizes */
	uint64_t *cs = malloc(sizeof(uint64_t) * (nc + 1)); double *wts = malloc(sizeof(double) * nc), ws = 0;
	for(int i = 0; i < nc; i++) { wts[i] = 0.2 + unif(); ws += wts[i]; }
	cs[0] = 0; for(int i = 0; i < nc; i++) { cs[i + 1] = cs[i] + (uint64_t)(wts[i] / ws * total); } cs[nc] = total;
	/* planted interspersed repeats: families of 300 bp .. 6 kb elements, copies at 80-99 % identity */
	uint64_t planted = 0, target = (uint64_t)(rf * total);
	while(planted < target) {
		uint64_t el = 300 + below(5700); if(el * 4 > total) el = total / 8 + 1;
		uint64_t src = below(total - el); int copies = 2 + (int)below(60);
		for(int c = 0; c < copies && planted < target; c++) {
			uint64_t dst = below(total - el); double id = 0.80 + 0.19 * unif();
			for(uint64_t i = 0; i < el; i++) { char ch = g[src + i]; if(unif() > id) ch = "ACGT"[rnd() >> 62]; g[dst + i] = ch; }
			planted += el;
		}
	}
	/* tandem arrays */
	for(int t = 0; t < (int)(rf * 200) + 1 && total > 10000; t++) {
		uint64_t ul = 2 + below(60), n = 5 + below(80), pos = below(total - ul * n - 1);
		for(uint64_t i = ul; i < ul * n; i++) g[pos + i] = g[pos + (i % ul)];
	}
	/* N runs */
	for(int t = 0; t < (int)(total / 2000000) + 1 && total > 100000; t++) { uint64_t pos = below(total - 2000), l = 10 + below(500); memset(g + pos, 'N', l); }
	for(int i = 0; i < nc; i++) { char name[64]; sprintf(name, "ctg%04d len=%lu", i, (unsigned long)(cs[i + 1] - cs[i])); put_fasta(stdout, name, g + cs[i], cs[i + 1] - cs[i]); }
	return 0;
}

/* one copy of an element (consensus[from, len)) at dst, on a random strand, `div` of its bases substituted and a few short indels (an inserted / dropped base per 1 / div0 bases) */
static char comp(char c);
static void plant(char *g, uint64_t total, uint64_t dst, const char *cons, uint64_t from, uint64_t len, double div)
{
	if(dst + len + 8 > total) return;
	const int rev = (int)(rnd() >> 63); uint64_t o = 0;
	for(uint64_t i = 0; i < len && o < len; i++) {
		char ch = rev ? comp(cons[from + len - 1 - i]) : cons[from + i];
		const double x = unif();
		if(x < div * 0.8) { ch = "ACGT"[rnd() >> 62]; }
		else if(x < div * 0.9) { continue; }                                     /* base dropped */
		else if(x < div) { g[dst + o++] = "ACGT"[rnd() >> 62]; if(o >= len) break; }      /* base inserted in front */
		g[dst + o++] = ch;
	}
}
static int main_genome_hard(int argc, char **argv)
{
	if(argc < 5) return 1;
	seed_rng(strtoull(argv[2], 0, 0));
	const uint64_t total = strtoull(argv[3], 0, 0); const int nc = atoi(argv[4]); const double rf = argc > 5 ? atof(argv[5]) : 0.45;
	char *g = malloc(total + 16);
	for(uint64_t i = 0; i < total; i++) g[i] = "ACGT"[rnd() >> 62];
	uint64_t *cs = malloc(sizeof(uint64_t) * (nc + 1)); double *wts = malloc(sizeof(double) * nc), ws = 0;
	for(int i = 0; i < nc; i++) { wts[i] = 0.2 + unif(); ws += wts[i]; }
	cs[0] = 0; for(int i = 0; i < nc; i++) { cs[i + 1] = cs[i] + (uint64_t)(wts[i] / ws * total); } cs[nc] = total;
	const uint64_t budget = (uint64_t)(rf * total);
	char *cons = malloc(8192 + 16);
	/* SINE-like: a few families of ~300-base elements, each copy 5 - 18 % off its consensus (22 % of the repeat budget: hundreds of thousands of copies in a 3 Gb genome) */
	{
		const int n_fam = 3; uint64_t left = budget * 22 / 100;
		for(int f = 0; f < n_fam; f++) {
			const uint64_t el = 270 + below(60), share = left / (uint64_t)(n_fam - f); const double age = 0.05 + 0.10 * unif();
			for(uint64_t i = 0; i < el; i++) cons[i] = "ACGT"[rnd() >> 62];
			for(uint64_t done = 0; done < share && total > 4 * el; done += el) plant(g, total, below(total - el - 8), cons, 0, el, age + 0.03 * unif());
			left -= share;
		}
	}
	/* LINE-like: 6 kb elements, copies truncated at their 5' end (most of them short), 3 - 20 % off */
	{
		const int n_fam = 4; uint64_t left = budget * 38 / 100;
		for(int f = 0; f < n_fam; f++) {
			const uint64_t el = 5000 + below(1500), share = left / (uint64_t)(n_fam - f); const double age = 0.03 + 0.14 * unif();
			for(uint64_t i = 0; i < el; i++) cons[i] = "ACGT"[rnd() >> 62];
			for(uint64_t done = 0; done < share && total > 4 * el;) { const double u = unif(); uint64_t ln = 300 + (uint64_t)((double)(el - 300) * u * u * u); plant(g, total, below(total - ln - 8), cons, el - ln, ln, age + 0.03 * unif()); done += ln; }
			left -= share;
		}
	}
	/* middle-sized families: 10 .. 10^4 copies (log-uniform) of 500 - 8 000-base elements, one age per family */
	{
		const int n_fam = 200; uint64_t left = budget * 20 / 100;
		for(int f = 0; f < n_fam && left > 0; f++) {
			const uint64_t el = 500 + below(7500); uint64_t copies = (uint64_t)exp(log(10.0) + unif() * (log(10000.0) - log(10.0)));
			const double age = 0.01 + 0.18 * unif();
			if(copies * el > left / (uint64_t)(n_fam - f) * 6) copies = left / (uint64_t)(n_fam - f) * 6 / el;
			if(copies < 2 || total < 8 * el) continue;
			for(uint64_t i = 0; i < el; i++) cons[i] = "ACGT"[rnd() >> 62];
			for(uint64_t c = 0; c < copies; c++) plant(g, total, below(total - el - 8), cons, 0, el, age + 0.02 * unif());
			left -= copies * el > left ? left : copies * el;
		}
	}
	/* segmental duplications: 10 - 200 kb stretches (repeats and all) copied elsewhere at 95 - 99.5 % identity */
	{
		uint64_t left = budget * 8 / 100; const uint64_t lmax = total / 64 < 200000 ? total / 64 : 200000, lmin = lmax / 20 + 1;
		while(left > lmin && lmax > 2000) {
			const uint64_t ln = lmin + below(lmax - lmin), src = below(total - ln), dst = below(total - ln); const double id = 0.95 + 0.045 * unif();
			if(src + ln > dst && dst + ln > src) continue;
			for(uint64_t i = 0; i < ln; i++) { char ch = g[src + i]; if(unif() > id) ch = "ACGT"[rnd() >> 62]; g[dst + i] = ch; }
			left -= ln > left ? left : ln;
		}
	}
	/* satellite arrays: a 171-base unit, 12 variants of it (1 - 3 % apart) as one higher-order block, the block repeated with 0.2 - 1 % noise over tens of kb to megabases */
	{
		uint64_t left = budget * 10 / 100; const uint64_t amax = total / 40 < 3000000 ? total / 40 : 3000000;
		char unit[256], *hor = malloc(12 * 256);
		while(left > 20000 && amax > 20000) {
			const uint64_t ul = 160 + below(24), alen = 20000 + below(amax - 20000), pos = below(total - alen - 1); const double noise = 0.002 + 0.008 * unif();
			for(uint64_t i = 0; i < ul; i++) unit[i] = "ACGT"[rnd() >> 62];
			for(int v = 0; v < 12; v++) for(uint64_t i = 0; i < ul; i++) hor[v * ul + i] = unif() < 0.02 ? "ACGT"[rnd() >> 62] : unit[i];
			for(uint64_t i = 0; i < alen; i++) { char ch = hor[i % (12 * ul)]; if(unif() < noise) ch = "ACGT"[rnd() >> 62]; g[pos + i] = ch; }
			left -= alen > left ? left : alen;
		}
	}
	/* microsatellites */
	for(uint64_t t = 0; t < total / 8000 + 1 && total > 10000; t++) { const uint64_t ul = 1 + below(6), n = 10 + below(50), pos = below(total - ul * n - 1); for(uint64_t i = ul; i < ul * n; i++) g[pos + i] = g[pos + (i % ul)]; }
	/* N: one long gap per contig (half a per cent to two per cent of it), and short runs */
	for(int i = 0; i < nc; i++) { const uint64_t cl = cs[i + 1] - cs[i]; if(cl < 20000) continue; const uint64_t l = cl / 200 + below(cl / 66), pos = cs[i] + cl / 4 + below(cl / 2 - l); memset(g + pos, 'N', l); }
	for(int t = 0; t < (int)(total / 2000000) + 1 && total > 100000; t++) { uint64_t pos = below(total - 2000), l = 10 + below(500); memset(g + pos, 'N', l); }
	for(int i = 0; i < nc; i++) { char name[64]; sprintf(name, "ctg%04d len=%lu", i, (unsigned long)(cs[i + 1] - cs[i])); put_fasta(stdout, name, g + cs[i], cs[i + 1] - cs[i]); }
	return 0;
}

typedef struct { char *name; char *seq; uint64_t len; } ctg_t;
static ctg_t *load_fasta(char const *fn, int *n)
{
	FILE *fp = fopen(fn, "r"); if(!fp) { perror(fn); exit(1); }
	ctg_t *c = NULL; int nc = 0; size_t cap = 0; char *line = NULL; size_t lcap = 0; ssize_t l;
	while((l = getline(&line, &lcap, fp)) > 0) {
		while(l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if(line[0] == '>') { c = realloc(c, sizeof(ctg_t) * (nc + 1)); char *sp = strpbrk(line + 1, " \t"); if(sp) *sp = 0; c[nc].name = strdup(line + 1); c[nc].seq = NULL; c[nc].len = 0; cap = 0; nc++; }
		else if(nc) { ctg_t *t = &c[nc - 1]; if(t->len + l + 1 > cap) { cap = (t->len + l + 1) * 2; t->seq = realloc(t->seq, cap); } memcpy(t->seq + t->len, line, l); t->len += l; }
	}
	fclose(fp); *n = nc; return c;
}
static char comp(char c) { switch(c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

static int main_reads(int argc, char **argv)
{
	if(argc < 6) return 1;
	int part = argc > 10 ? atoi(argv[9]) : -1, n_parts = argc > 10 ? atoi(argv[10]) : 1;
	if(n_parts < 1 || part >= n_parts) return 1;
	seed_rng(strtoull(argv[2], 0, 0) + (part >= 0 ? 0x9e3779b97f4a7c15ULL * (uint64_t)(part + 1) : 0));
	int nc; ctg_t *c = load_fasta(argv[3], &nc);
	double depth = atof(argv[4]); int ont = strcmp(argv[5], "ont") == 0; int fq = argc > 6 && strcmp(argv[6], "fq") == 0;
	double lm = argc > 8 ? atof(argv[7]) : 20000.0, lsd = argc > 8 ? atof(argv[8]) : 2000.0;
	uint64_t total = 0; for(int i = 0; i < nc; i++) total += c[i].len;
	uint64_t want = (uint64_t)(depth * total / n_parts), made = 0, id = 0;
	char *buf = malloc(4 * 1000000 + 16), *qb = malloc(4 * 1000000 + 16);
	while(made < want) {
		double len_d; 
		if(ont) { len_d = exp(log(8000.0) + 0.9 * gauss()); } else { len_d = lm + lsd * gauss(); }
		if(len_d < 1000) len_d = 1000; if(len_d > 900000) len_d = 900000;
		uint64_t len = (uint64_t)len_d;
		/* pick a contig proportional to length */
		uint64_t p = below(total); int ci = 0; while(p >= c[ci].len) { p -= c[ci].len; ci++; }
		if(len > c[ci].len) len = c[ci].len;
		if(p + len > c[ci].len) p = c[ci].len - len;
		double acc = ont ? 0.90 + 0.05 * gauss() : 0.88 + 0.07 * gauss();
		if(acc < 0.6) acc = 0.6; if(acc > 0.99) acc = 0.99;
		double e = 1.0 - acc, ps, pi, pd;
		if(ont) { ps = e * 0.4; pi = e * 0.2; pd = e * 0.4; } else { ps = e * 0.10; pi = e * 0.60; pd = e * 0.30; }
		int rev = rnd() >> 63;
		uint64_t n = 0;
		for(uint64_t i = 0; i < len; i++) {
			char ch = rev ? comp(c[ci].seq[p + len - 1 - i]) : c[ci].seq[p + i];
			double x = unif();
			double pdl = pd; if(ont && i > 0 && n > 0 && buf[n - 1] == ch) pdl = pd * 2.0;   /* homopolymer-biased deletions */
			if(x < pdl) continue;
			if(x < pdl + ps) { char m; do { m = "ACGT"[rnd() >> 62]; } while(m == ch); buf[n++] = m; }
			else { buf[n++] = ch; }
			while(unif() < pi && n < 3900000) { buf[n++] = "ACGT"[rnd() >> 62]; }
		}
		buf[n] = 0;
		char name[160];
		if(part >= 0) sprintf(name, "r%d.%lu_%s_%lu_%lu_%c_%.3f", part, (unsigned long)id, c[ci].name, (unsigned long)p, (unsigned long)(p + len), rev ? '-' : '+', acc);
		else sprintf(name, "r%lu_%s_%lu_%lu_%c_%.3f", (unsigned long)id, c[ci].name, (unsigned long)p, (unsigned long)(p + len), rev ? '-' : '+', acc);
		if(fq) { for(uint64_t i = 0; i < n; i++) qb[i] = (char)('!' + 5 + below(20)); qb[n] = 0; printf("@%s\n%s\n+\n%s\n", name, buf, qb); }
		else { printf(">%s\n%s\n", name, buf); }
		made += n; id++;
	}
	return 0;
}

int main(int argc, char **argv)
{
	if(argc > 1 && strcmp(argv[1], "genome") == 0) return main_genome(argc, argv);
	if(argc > 1 && strcmp(argv[1], "genomehard") == 0) return main_genome_hard(argc, argv);
	if(argc > 1 && strcmp(argv[1], "reads") == 0) return main_reads(argc, argv);
	fprintf(stderr, "usage: gensim genome <seed> <total_len> <n_contigs> <repeat_frac> | gensim reads <seed> ref.fa <depth> <pacbio|ont> [fq|fa] [len_mean len_sd]\n");
	return 1;
}
