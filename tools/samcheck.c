/*
 * samcheck.c -- test tool: size-independent properties of a SAM stream too large to walk in Python (the 13 GB the hg38-size x3 set maps to), at pipe speed.
 *
 *   minialign ... ref.fa reads.fa | tools/samcheck reads.fa[,more.fa,...] N head.sam
 *
 * Reads the names and lengths of the reads from reads.fa (plain FASTA as tools/gensim writes it), then the SAM text from stdin and checks, record by record:
 *   - one primary record (flag without 0x100 / 0x800) per read, in input order; the other records of a read follow its primary record
 *   - an unmapped record (0x4) has RNAME and CIGAR '*'
 *   - a mapped one: the CIGAR parses completely, M/I/S/=/X/H add up to the read's length, M/I/S/=/X to the length of SEQ, and POS .. POS + (M/D/N/=/X) lies
 *     inside the reference sequence the @SQ header announced
 * The records of the first N reads are copied to head.sam (for a byte comparison with the compiled reference at -t1).  Prints one JSON line: counts, the
 * first violation if any, and an order-dependent 2 x 64-bit digest of every record line (two streams of the same set must agree on it).
 * Reads named r<p>.<i>_... (the parts tools/gensim writes side by side) also get a digest and a record count per part p ("parts": [[records, "digest"], ...]), so that
 * the stream of a whole set can be compared with the compiled reference part by part:
 *   cat last_reads_of_part_p-1.fa part_p.fa | oracle/_ref/minialign -t1 ref.mai | tools/samcheck --parts
 * prints only those (no read file, no property checks); the few reads in front carry the state the reference's thread buffer would be in at the start of part p
 * (DESIGN.md 5), their own records count under part p - 1 and are ignored by the caller.
 * Test infrastructure only; nothing in the product uses it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef struct { char *name; uint32_t nlen, len; } rd_t;
typedef struct { char *name; uint32_t nlen; uint64_t len; } sq_t;

static uint64_t mix(uint64_t x) { x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; return x; }
static void line_hash(const char *p, size_t n, uint64_t *a, uint64_t *b)
{
	uint64_t h1 = 0x9e3779b97f4a7c15ull ^ n, h2 = 0xc2b2ae3d27d4eb4full + n, w;
	size_t i = 0;
	for(; i + 8 <= n; i += 8) { memcpy(&w, p + i, 8); h1 = (h1 ^ w) * 0x100000001b3ull; h1 ^= h1 >> 29; h2 = (h2 + w) * 0xff51afd7ed558ccdull; h2 ^= h2 >> 31; }
	w = 0; memcpy(&w, p + i, n - i); h1 = (h1 ^ w) * 0x100000001b3ull; h2 = (h2 + w) * 0xff51afd7ed558ccdull;
	*a = mix(h1); *b = mix(h2);
}

#define MAX_PARTS 64
static uint64_t pd1[MAX_PARTS], pd2[MAX_PARTS], pn[MAX_PARTS];
/* the part of a record by its QNAME (r<p>.<i>_...), -1 for any other name */
static int part_of(const char *q, size_t n)
{
	if(n < 3 || q[0] != 'r') return -1;
	int p = 0; size_t i = 1;
	for(; i < n && q[i] >= '0' && q[i] <= '9' && i < 4; i++) p = p * 10 + (q[i] - '0');
	if(i == 1 || i >= n || q[i] != '.' || p >= MAX_PARTS) return -1;
	return p;
}
static void part_add(const char *line, size_t ln, uint64_t a, uint64_t b)
{
	const char *t = memchr(line, '\t', ln); const int p = part_of(line, t ? (size_t)(t - line) : ln);
	if(p < 0) return;
	pd1[p] = pd1[p] * 0x9e3779b97f4a7c15ull + a; pd2[p] = (pd2[p] ^ b) * 0xff51afd7ed558ccdull + 1; pn[p]++;
}
static void parts_print(FILE *out)
{
	int last = -1; for(int p = 0; p < MAX_PARTS; p++) if(pn[p]) last = p;
	fprintf(out, "\"parts\": [");
	for(int p = 0; p <= last; p++) fprintf(out, "%s[%lu, \"%016lx%016lx\"]", p ? ", " : "", (unsigned long)pn[p], (unsigned long)pd1[p], (unsigned long)pd2[p]);
	fprintf(out, "]");
}
int main(int argc, char **argv)
{
	if(argc >= 2 && !strcmp(argv[1], "--parts")) {
		char *line = NULL; size_t lcap = 0; ssize_t ln; uint64_t n_rec = 0;
		while((ln = getline(&line, &lcap, stdin)) > 0) { if(line[0] == '@') continue; uint64_t a, b; line_hash(line, (size_t)ln, &a, &b); part_add(line, (size_t)ln, a, b); n_rec++; }
		printf("{\"records\": %lu, ", (unsigned long)n_rec); parts_print(stdout); printf("}\n");
		return 0;
	}
	if(argc < 4) { fprintf(stderr, "usage: samcheck reads.fa n_head head.sam < sam\n"); return 2; }
	const uint64_t n_head = strtoull(argv[2], NULL, 10);
	/* reads: names and lengths (argv[1]: one file, or several separated by commas -- the parts of a set in order) */
	size_t cap = 1 << 20, n_rd = 0; rd_t *rd = malloc(cap * sizeof(rd_t));
	char *flist = strdup(argv[1]);
	for(char *fn = strtok(flist, ","); fn != NULL; fn = strtok(NULL, ",")) {
		FILE *fp = fopen(fn, "rb"); if(!fp) { perror(fn); return 2; }
		size_t bcap = 64u << 20; char *buf = malloc(bcap + 1); size_t have = 0, got; int in_name = 0;
		char nm[4096]; size_t nl = 0; int name_done = 0; int at_line_start = 1;
		while((got = fread(buf, 1, bcap, fp)) > 0) {
			(void)have;
			for(size_t i = 0; i < got;) {
				if(at_line_start && buf[i] == '>') {
					if(n_rd == cap) { cap *= 2; rd = realloc(rd, cap * sizeof(rd_t)); }
					rd[n_rd].len = 0; rd[n_rd].name = NULL; n_rd++; in_name = 1; nl = 0; name_done = 0; at_line_start = 0; i++; continue;
				}
				at_line_start = 0;
				if(in_name) {
					for(; i < got && buf[i] != '\n'; i++) { if(!name_done) { if(buf[i] == ' ' || buf[i] == '\t' || buf[i] == '\r') name_done = 1; else if(nl < sizeof(nm) - 1) nm[nl++] = buf[i]; } }
					if(i < got) { in_name = 0; rd[n_rd - 1].name = malloc(nl + 1); memcpy(rd[n_rd - 1].name, nm, nl); rd[n_rd - 1].name[nl] = 0; rd[n_rd - 1].nlen = (uint32_t)nl; at_line_start = 1; i++; }
				} else {
					char *e = memchr(buf + i, '\n', got - i); size_t n = e ? (size_t)(e - (buf + i)) : got - i;
					if(n_rd) rd[n_rd - 1].len += (uint32_t)n;
					i += n; if(e) { at_line_start = 1; i++; }
				}
			}
		}
		free(buf); fclose(fp);
	}
	free(flist);
	FILE *hf = fopen(argv[3], "wb"); if(!hf) { perror(argv[3]); return 2; }
	sq_t *sq = NULL; size_t n_sq = 0, sq_cap = 0;
	char *line = NULL; size_t lcap = 0; ssize_t ln;
	uint64_t n_rec = 0, n_prim = 0, n_mapped = 0, n_unmapped = 0, n_sec = 0, n_supp = 0, bytes = 0, bases_mapped = 0, d1 = 0, d2 = 0;
	int64_t cur = -1; char err[512] = ""; size_t last_sq = 0;
	while((ln = getline(&line, &lcap, stdin)) > 0) {
		if(line[0] == '@') {
			if(!strncmp(line, "@SQ", 3)) {
				char *sn = strstr(line, "SN:"), *l = strstr(line, "LN:");
				if(sn && l) { if(n_sq == sq_cap) { sq_cap = sq_cap ? sq_cap * 2 : 64; sq = realloc(sq, sq_cap * sizeof(sq_t)); } sn += 3; size_t k = strcspn(sn, "\t\n"); sq[n_sq].name = malloc(k + 1); memcpy(sq[n_sq].name, sn, k); sq[n_sq].name[k] = 0; sq[n_sq].nlen = (uint32_t)k; sq[n_sq].len = strtoull(l + 3, NULL, 10); n_sq++; }
			}
			continue;
		}
		bytes += (uint64_t)ln; n_rec++;
		{ uint64_t a, b; line_hash(line, (size_t)ln, &a, &b); d1 = d1 * 0x9e3779b97f4a7c15ull + a; d2 = (d2 ^ b) * 0xff51afd7ed558ccdull + 1; part_add(line, (size_t)ln, a, b); }
		if(err[0]) continue;
		/* fields */
		char *f[11]; int nf = 0; char *p = line; f[nf++] = p;
		while(nf < 11) { char *t = memchr(p, '\t', (size_t)(line + ln - p)); if(!t) break; p = t + 1; f[nf++] = p; }
		if(nf < 11) { snprintf(err, sizeof(err), "record %lu: fewer than 11 fields", (unsigned long)n_rec); continue; }
		const size_t nlen = (size_t)(f[1] - f[0] - 1); const unsigned flag = (unsigned)strtoul(f[1], NULL, 10);
		const size_t rnlen = (size_t)(f[3] - f[2] - 1); const uint64_t pos = strtoull(f[3], NULL, 10);
		const char *cg = f[5]; const size_t cglen = (size_t)(f[6] - f[5] - 1); const size_t sqlen = (size_t)(f[10] - f[9] - 1);
		const int prim = !(flag & (0x100 | 0x800));
		if(prim) {
			cur++; n_prim++;
			if((uint64_t)cur >= n_rd || rd[cur].nlen != nlen || memcmp(rd[cur].name, f[0], nlen)) { snprintf(err, sizeof(err), "record %lu: primary record of `%.*s' where read %ld (`%s') is due", (unsigned long)n_rec, (int)nlen, f[0], (long)cur, (uint64_t)cur < n_rd ? rd[cur].name : "<none>"); continue; }
		} else {
			if(cur < 0 || rd[cur].nlen != nlen || memcmp(rd[cur].name, f[0], nlen)) { snprintf(err, sizeof(err), "record %lu: secondary / supplementary record of `%.*s' not behind its primary", (unsigned long)n_rec, (int)nlen, f[0]); continue; }
			if(flag & 0x100) n_sec++; else n_supp++;
		}
		if((uint64_t)cur < n_head) fwrite(line, 1, (size_t)ln, hf);
		if(flag & 4) {
			n_unmapped++;
			if(!(cglen == 1 && cg[0] == '*' && rnlen == 1 && f[2][0] == '*')) snprintf(err, sizeof(err), "record %lu: unmapped record with RNAME / CIGAR", (unsigned long)n_rec);
			continue;
		}
		uint64_t q_all = 0, q_seq = 0, r_span = 0, num = 0; int bad = 0, digits = 0;
		for(size_t i = 0; i < cglen; i++) {
			const char c = cg[i];
			if(c >= '0' && c <= '9') { num = num * 10 + (uint64_t)(c - '0'); digits++; continue; }
			if(!digits || num == 0) { bad = 1; break; }
			switch(c) { case 'M': case '=': case 'X': q_all += num; q_seq += num; r_span += num; break; case 'I': case 'S': q_all += num; q_seq += num; break; case 'H': q_all += num; break; case 'D': case 'N': r_span += num; break; case 'P': break; default: bad = 1; }
			num = 0; digits = 0;
		}
		if(bad || digits) { snprintf(err, sizeof(err), "record %lu: CIGAR does not parse", (unsigned long)n_rec); continue; }
		if(q_all != rd[cur].len) { snprintf(err, sizeof(err), "record %lu (`%s'): CIGAR covers %lu bases of a read of %u", (unsigned long)n_rec, rd[cur].name, (unsigned long)q_all, rd[cur].len); continue; }
		if(q_seq != sqlen) { snprintf(err, sizeof(err), "record %lu (`%s'): CIGAR asks for %lu bases of SEQ, %lu are there", (unsigned long)n_rec, rd[cur].name, (unsigned long)q_seq, (unsigned long)sqlen); continue; }
		size_t si = n_sq;
		if(last_sq < n_sq && sq[last_sq].nlen == rnlen && !memcmp(sq[last_sq].name, f[2], rnlen)) si = last_sq;
		else for(size_t i = 0; i < n_sq; i++) if(sq[i].nlen == rnlen && !memcmp(sq[i].name, f[2], rnlen)) { si = i; break; }
		if(si == n_sq) { snprintf(err, sizeof(err), "record %lu: RNAME `%.*s' is not in the header", (unsigned long)n_rec, (int)rnlen, f[2]); continue; }
		last_sq = si;
		if(pos < 1 || pos - 1 + r_span > sq[si].len) { snprintf(err, sizeof(err), "record %lu (`%s'): POS %lu + %lu leaves `%s' (%lu)", (unsigned long)n_rec, rd[cur].name, (unsigned long)pos, (unsigned long)r_span, sq[si].name, (unsigned long)sq[si].len); continue; }
		if(prim) { n_mapped++; bases_mapped += rd[cur].len; }
	}
	fclose(hf);
	if(!err[0] && (uint64_t)(cur + 1) != n_rd) snprintf(err, sizeof(err), "%lu primary records for %lu reads", (unsigned long)n_prim, (unsigned long)n_rd);
	for(char *q = err; *q; q++) if(*q == '"' || *q == '\\') *q = '\'';
	printf("{\"reads\": %lu, \"records\": %lu, \"primary\": %lu, \"mapped\": %lu, \"unmapped\": %lu, \"secondary\": %lu, \"supplementary\": %lu, \"bytes\": %lu, \"bases_mapped\": %lu, \"contigs\": %lu, \"digest\": \"%016lx%016lx\", \"error\": \"%s\", ",
		(unsigned long)n_rd, (unsigned long)n_rec, (unsigned long)n_prim, (unsigned long)n_mapped, (unsigned long)n_unmapped, (unsigned long)n_sec, (unsigned long)n_supp, (unsigned long)bytes, (unsigned long)bases_mapped, (unsigned long)n_sq, (unsigned long)d1, (unsigned long)d2, err);
	parts_print(stdout); printf("}\n");
	return err[0] ? 1 : 0;
}
