/*
 * mm_host.hip -- host side of include/minialign.h: options, FASTA/FASTQ reader, index construction,
 * the GPU batch pipeline (K1 sketch/lookup/expand -> K2 sort/chain -> K3 extend, in occurrence-threshold rounds),
 * post-map (prune / supplementary / MAPQ) and the SAM printer.
 *
 * Reference behaviour mirrored here (file:line in /root/reference):
 *   options / presets   minialign.c:5846-5900, 6141-6162      reader        minialign.c:1996-2090, 214-229
 *   index build         minialign.c:2767-3040, ksort.h:84-131  post-map      minialign.c:4185-4398
 *   SAM                 minialign.c:5096-5426, gaba_parse.h:168-263
 * The reference streams 512 KB batches through a pthread queue (minialign.c:4535-4732); here a batch is as
 * many reads as fit the device pools, every device stage is one launch over the whole batch, and batches are
 * drained strictly in input order.
 */
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <deque>
#include <cerrno>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <string>
#include <vector>
#include <thread>
#include <map>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <algorithm>
#include "gaba_host.hpp"
#include "mm_device.hpp"
#include "mm_index.hpp"
#include "mm_cigar.hpp"
#include "../../include/minialign.h"

using namespace mm;

namespace {

double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline uint32_t h_d2u32(double d) { if(!(d > -9.2e18 && d < 9.2e18)) return 0; return (uint32_t)(int64_t)d; }
inline uint32_t h_f2u32(float f) { if(!(f > -9.2e18f && f < 9.2e18f)) return 0; return (uint32_t)(int64_t)f; }
inline int32_t h_ofs(int32_t x) { return (int32_t)0x40000000 - x; }

/* ---------------------------------------------------------------------------------------------
 * the reference's radix sort (ksort.h:84-131): in-place MSD radix on 8-bit digits, cycle-leader
 * permutation (unstable), insertion sort below 65 elements.  Its exact permutation decides the order of
 * hits inside an index value list and of equal-score results, so it is implemented, not substituted.
 * --------------------------------------------------------------------------------------------- */
template<typename T, typename K> struct KRadix {
	static void insertion(T *beg, T *end, K key)
	{
		for(T *i = beg + 1; i < end; ++i) {
			if(key(*i) < key(*(i - 1))) {
				T tmp = *i, *j;
				for(j = i; j > beg && key(tmp) < key(*(j - 1)); --j) *j = *(j - 1);
				*j = tmp;
			}
		}
	}
	static void msd(T *beg, T *end, int s, K key)
	{
		struct B { T *b, *e; } b[256];
		for(int k = 0; k < 256; k++) b[k].b = b[k].e = beg;
		for(T *i = beg; i != end; ++i) ++b[(key(*i) >> s) & 255].e;
		for(int k = 1; k < 256; k++) { b[k].e += b[k - 1].e - beg; b[k].b = b[k - 1].e; }
		for(int k = 0; k < 256;) {
			if(b[k].b != b[k].e) {
				int l = (int)((key(*b[k].b) >> s) & 255);
				if(l != k) {
					T tmp = *b[k].b, swp;
					do { swp = tmp; tmp = *b[l].b; *b[l].b++ = swp; l = (int)((key(tmp) >> s) & 255); } while(l != k);
					*b[k].b++ = tmp;
				} else ++b[k].b;
			} else ++k;
		}
		b[0].b = beg; for(int k = 1; k < 256; k++) b[k].b = b[k - 1].e;
		if(s) {
			int ns = s > 8 ? s - 8 : 0;
			for(int k = 0; k < 256; k++) {
				if(b[k].e - b[k].b > 64) msd(b[k].b, b[k].e, ns, key);
				else if(b[k].e - b[k].b > 1) insertion(b[k].b, b[k].e, key);
			}
		}
	}
	static void sort(T *p, size_t n, int key_bits, K key) { if(n <= 64) insertion(p, p + n, key); else msd(p, p + n, key_bits - 8, key); }
};
struct Mini { uint64_t hrem; uint32_t pos, rid; };
struct ResEnt { uint32_t score, iid; };
inline void sort_minis(Mini *p, size_t n) { auto key = [](const Mini &m) { return m.hrem; }; KRadix<Mini, decltype(key)>::sort(p, n, 64, key); }
inline void sort_res(ResEnt *p, size_t n) { auto key = [](const ResEnt &m) { return m.score; }; KRadix<ResEnt, decltype(key)>::sort(p, n, 32, key); }

/* ---------------------------------------------------------------------------------------------
 * sequences
 * --------------------------------------------------------------------------------------------- */
struct HSeq { std::string name; std::vector<uint8_t> seq; std::string qual, comment; bool has_comment = false; bool circular = false;
	uint64_t t_off = 0; uint32_t t_len = 0; int32_t t_id = -1;
	uint32_t len = 0;          /* a reference sequence whose bases stayed in the text (device build of the index): its length; seq is filled when a printer asks (ref_codes) */
	uint32_t blen() const { return len ? len : (uint32_t)seq.size(); } };          /* where the bases stand in the text of the file (when that is kept): every byte of the extent but '\n' is a base */     /* qual / comment: kept on request only (-Q, -T CO) */

/* run fn(t, nth) on up to `cap` (default 32) host threads (reads / records are independent in every host stage that uses this) */
template<typename F> static void host_parallel(uint32_t want, F fn, uint32_t cap = 32)
{
	const uint32_t nth = std::max<uint32_t>(1, std::min<uint32_t>(want, std::min<uint32_t>(std::max<uint32_t>(1, std::thread::hardware_concurrency()), cap)));
	std::vector<std::thread> th;
	for(uint32_t t = 1; t < nth; t++) th.emplace_back(fn, t, nth);
	fn(0u, nth);
	for(auto &x : th) x.join();
}

/* FASTA / FASTQ text; bases by the low-nibble table of minialign.c:223-229 (anything but ACGTUN -> A).  A record name runs to the first space;
 * a tab does not end it but is rewritten to a space (bseq_read_fasta copies through an escape while testing the raw bytes, minialign.c:1957-1968) */
/* The reader follows bseq_read_fasta (minialign.c:1996-2090) byte for byte, oddities included:
 *   - the file type is the first '>' or '@' among the first four bytes (minialign.c:1784-1792); what stands in front of it is dropped
 *   - name: spaces skipped, then up to the first space or end of line, tabs rewritten to spaces, one trailing CR dropped; a comment exists when the
 *     name ended at a space: spaces skipped, to the end of the line, tabs to spaces, one CR and then the spaces at the end dropped
 *   - bases: every byte of the following lines goes through the low-nibble table -- a CR too (it reads as A) -- until the record delimiter ('>' for
 *     FASTA, '+' for FASTQ) shows up ANYWHERE in a line, or the text ends
 *   - FASTQ: the rest of the '+' line is skipped, then quality lines are taken until their length reaches the number of bases (counted without a
 *     trailing CR when the qualities are kept, with it when they are only skipped), newlines after that are skipped and the next byte must be '@' */
/* header line at p (just behind the delimiter): fills name / comment, returns the first byte of the next line */
static const char *parse_header(const char *p, const char *end, HSeq &r, bool keep_comment)
{
	while(p < end && *p == ' ') p++;
	const char *b0 = p; while(p < end && *p != ' ' && *p != '\n') p++;
	size_t ln = (size_t)(p - b0); const bool has_comment = p < end && *p == ' ';
	if(ln > 0 && b0[ln - 1] == '\r') ln--;
	r.name.assign(b0, ln); for(char &ch : r.name) if(ch == '\t') ch = ' ';
	if(p < end) p++;
	if(has_comment) {
		while(p < end && *p == ' ') p++;
		const char *c0 = (const char *)p, *nl = (const char *)memchr(p, '\n', (size_t)(end - p)); const char *le = nl ? nl : end;
		size_t cl = (size_t)(le - c0);
		p = nl ? nl + 1 : end;
		if(cl > 0 && c0[cl - 1] == '\r') cl--;
		while(cl > 0 && c0[cl - 1] == ' ') cl--;
		if(keep_comment) { r.comment.assign(c0, cl); r.has_comment = true; for(char &ch : r.comment) if(ch == '\t') ch = ' '; }
	}
	return p;
}
/* sequence lines at p up to the delimiter dv anywhere in a line (returns its position) or the end of the text */
static const char *parse_bases(const char *p, const char *end, char dv, const uint8_t *enc, std::vector<uint8_t> &sq, bool &at_delim, const char *base = nullptr, HSeq *rec = nullptr)
{
	at_delim = false;
	if(rec) { rec->t_off = (uint64_t)(p - base); rec->t_len = 0; }
	while(p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p)); const char *le = nl ? nl : end;
		const char *dl = (const char *)memchr(p, dv, (size_t)(le - p)); const char *stop = dl ? dl : le;
		const size_t n = (size_t)(stop - p), o = sq.size();
		if(n) { sq.resize(o + n); uint8_t *d = sq.data() + o; for(size_t i = 0; i < n; i++) d[i] = enc[p[i] & 15]; if(rec) { rec->t_len = (uint32_t)((uint64_t)(stop - base) - rec->t_off); } }
		if(dl) { at_delim = true; return dl; }
		p = nl ? nl + 1 : end;
	}
	return p;
}
/* one FASTA stretch (starts at a '>', holds whole records) */
static void parse_fasta_span(const char *p, const char *end, const uint8_t *enc, std::vector<HSeq> &out, bool keep_comment, const char *base = nullptr)
{
	while(p < end) {
		p++;                                         /* the '>' */
		out.emplace_back(); HSeq &r = out.back();
		p = parse_header(p, end, r, keep_comment);
		bool at; p = parse_bases(p, end, '>', enc, r.seq, at, base, base ? &r : nullptr);
	}
}
/* FASTQ, in sequence; false when a record does not start with '@' where one must */
static bool parse_fastq(const char *p, const char *end, const uint8_t *enc, std::vector<HSeq> &out, bool keep_qual, bool keep_comment, const char *base = nullptr)
{
	while(p < end) {
		if(*p++ != '@') return false;
		out.emplace_back(); HSeq &r = out.back();
		p = parse_header(p, end, r, keep_comment);
		bool at; p = parse_bases(p, end, '+', enc, r.seq, at, base, base ? &r : nullptr);
		if(!at) break;
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p)); p = nl ? nl + 1 : end;      /* the '+' line */
		uint64_t acc = 0; const uint64_t lim = r.seq.size();
		while(p < end) {
			nl = (const char *)memchr(p, '\n', (size_t)(end - p)); const char *le = nl ? nl : end;
			size_t ll = (size_t)(le - p);
			if(keep_qual) { size_t kl = ll; if(kl > 0 && p[kl - 1] == '\r') kl--; r.qual.append(p, kl); acc += kl; } else acc += ll;
			p = le;
			if(p >= end || acc >= lim) break;
			p++;
		}
		while(p < end && *p == '\n') p++;
	}
	return true;
}
/* part `part` of `n_parts` of a plain FASTA file, cut where a '>' starts a line (such a '>' always starts a record, see parse_fasta_span): only the bytes of the part are
 * read.  false when the file is not plain FASTA (gzip, FASTQ, stdin: the caller reads it whole and takes its share of the records) or cannot be read */
static bool read_fasta_part(const char *fn, uint32_t part, uint32_t n_parts, std::vector<char> &data)
{
	if(strcmp(fn, "-") == 0) return false;
	FILE *fp = fopen(fn, "rb");
	if(!fp) return false;
	bool ok = false;
	do {
		uint8_t head[4]; const size_t hn = fread(head, 1, 4, fp);
		bool fasta = false; for(size_t i = 0; i < hn; i++) { if(head[i] == '>') { fasta = true; break; } if(head[i] == '@' || (i == 0 && head[i] == 0x1f)) break; }
		if(!fasta || fseek(fp, 0, SEEK_END) != 0) break;
		const int64_t size = ftell(fp); if(size < 0) break;
		/* boundary b -> the first line start at or behind it that holds a '>' (0 stays 0, the end stays the end) */
		auto cut = [&](int64_t b) -> int64_t {
			if(b <= 0) return 0;
			if(b >= size) return size;
			std::vector<char> buf(1 << 20); int64_t at = b - 1;          /* the byte in front decides whether b itself starts a line */
			while(at < size) {
				if(fseek(fp, (long)at, SEEK_SET) != 0) return -1;
				const size_t got = fread(buf.data(), 1, buf.size(), fp); if(got == 0) return -1;
				for(size_t i = 0; i + 1 < got; i++) if(buf[i] == '\n' && buf[i + 1] == '>') return at + (int64_t)i + 1;
				at += (int64_t)got - 1;                                    /* the last byte is looked at again as the first of the next piece */
				if(got < 2) break;
			}
			return size;
		};
		const int64_t lo = cut(size * (int64_t)part / n_parts), hi = cut(size * (int64_t)(part + 1) / n_parts);
		if(lo < 0 || hi < 0) break;
		data.resize((size_t)(hi - lo));
		if(hi > lo && (fseek(fp, (long)lo, SEEK_SET) != 0 || fread(data.data(), 1, data.size(), fp) != data.size())) break;
		ok = true;
	} while(0);
	fclose(fp);
	return ok;
}
bool read_seq_file(const char *fn, std::vector<HSeq> &out, uint32_t min_len = 1, bool keep_qual = false, bool keep_comment = false, std::vector<char> *keep_text = nullptr, int32_t text_id = -1, uint32_t part = 0, uint32_t n_parts = 1)
{
	std::vector<char> data;
	const bool ranged = n_parts > 1 && read_fasta_part(fn, part, n_parts, data);
	if(ranged && data.empty()) { if(keep_text) keep_text->clear(); return true; }          /* a part without a record */
	const size_t out0 = out.size();
	FILE *fp = ranged ? NULL : (strcmp(fn, "-") == 0 ? stdin : fopen(fn, "rb"));
	if(!fp && !ranged) return false;
	uint8_t enc[16] = { 0 };
	enc['A' & 15] = 0; enc['C' & 15] = 1; enc['G' & 15] = 2; enc['T' & 15] = 3; enc['U' & 15] = 3; enc['N' & 15] = 4;
	/* the whole file in memory first (the reads are kept in memory anyway) */
	if(!ranged) {
		size_t cap = 1 << 22, len = 0; 
		if(fp != stdin && fseek(fp, 0, SEEK_END) == 0) { long sz = ftell(fp); if(sz > 0) cap = (size_t)sz + 1; rewind(fp); }
		data.resize(cap);
		size_t got;
		while((got = fread(data.data() + len, 1, data.size() - len, fp)) > 0) { len += got; if(len == data.size()) data.resize(data.size() * 2); }
		data.resize(len);
	}
	if(fp && fp != stdin) fclose(fp);
	/* gzip input is inflated in memory (the reference reads through gzread, minialign.c:1184-1363: plain and gzip text alike, members back to back) */
	if(data.size() >= 2 && (uint8_t)data[0] == 0x1f && (uint8_t)data[1] == 0x8b) {
		std::vector<char> raw; raw.resize(std::max<size_t>(data.size() * 4, 1 << 16));
		z_stream zs; memset(&zs, 0, sizeof(zs));
		if(inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
		zs.next_in = (Bytef *)data.data(); size_t in_left = data.size(), out_len = 0;
		bool ok = true;
		while(ok) {
			zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30); const size_t in_before = zs.avail_in;
			if(raw.size() - out_len < (1u << 16)) raw.resize(raw.size() * 2);
			zs.next_out = (Bytef *)raw.data() + out_len; zs.avail_out = (uInt)std::min<size_t>(raw.size() - out_len, 1u << 30); const size_t out_before = zs.avail_out;
			int rc = inflate(&zs, Z_NO_FLUSH);
			in_left -= in_before - zs.avail_in; out_len += out_before - zs.avail_out;
			if(rc == Z_STREAM_END) { if(in_left < 2 || (uint8_t)zs.next_in[0] != 0x1f || (uint8_t)zs.next_in[1] != 0x8b) break; if(inflateReset(&zs) != Z_OK) ok = false; }
			else if(rc != Z_OK && !(rc == Z_BUF_ERROR && zs.avail_out == 0)) ok = false;
			else if(in_left == 0 && zs.avail_out != 0) ok = false;          /* truncated stream */
		}
		inflateEnd(&zs);
		if(!ok) { fprintf(stderr, "[minialign_amd] broken gzip stream in `%s'\n", fn); return false; }
		raw.resize(out_len); data.swap(raw);
	}
	size_t first = 0; char delim = 0;
	for(int i = 0; i < 4 && first < data.size(); i++) { if(data[first] == '>' || data[first] == '@') { delim = data[first]; break; } first++; }
	if(!delim) { fprintf(stderr, "[minialign_amd] `%s' is neither FASTA nor FASTQ\n", fn); return false; }
	if(delim == '>') {
		/* FASTA: a '>' at the beginning of a line always starts a record, so the file splits at such points and the pieces are
		 * parsed on host threads, each into its own list */
		const char *base = data.data(), *end = base + data.size();
		std::vector<std::vector<HSeq>> part;
		std::vector<const char *> cut;
		{
			const uint32_t nth = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(std::max<uint32_t>(1, std::thread::hardware_concurrency()), 32), (uint32_t)(data.size() >> 20) + 1));
			cut.push_back(base + first);
			for(uint32_t t = 1; t < nth; t++) {
				const char *q = base + data.size() * t / nth;
				while(q < end) { const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q)); if(!nl || nl + 1 >= end) { q = end; break; } if(nl[1] == '>') { q = nl + 1; break; } q = nl + 1; }
				if(q > cut.back()) cut.push_back(q);
			}
			cut.push_back(end);
			part.resize(cut.size() - 1);
		}
		host_parallel((uint32_t)part.size(), [&](uint32_t t, uint32_t nth) { for(size_t i = t; i < part.size(); i += nth) parse_fasta_span(cut[i], cut[i + 1], enc, part[i], keep_comment, keep_text ? base : nullptr); });
		size_t tot = 0; for(auto &v : part) tot += v.size();
		out.reserve(out.size() + tot);
		for(auto &v : part) for(auto &r : v) out.emplace_back(std::move(r));
	} else if(!parse_fastq(data.data() + first, data.data() + data.size(), enc, out, keep_qual, keep_comment, keep_text ? data.data() : nullptr)) {
		fprintf(stderr, "[minialign_amd] `%s': broken FASTQ record\n", fn); return false;       /* the reference gives up on the run (exit 1) */
	}
	/* -L (default 1, minialign.c:2077, 6145): records shorter than the limit are dropped, reference and query side alike */
	out.erase(std::remove_if(out.begin() + out0, out.end(), [min_len](const HSeq &s) { return s.seq.size() < min_len; }), out.end());
	/* a part of a file that could not be cut by bytes: its share of the records */
	if(n_parts > 1 && !ranged) {
		const size_t n = out.size() - out0, lo = n * part / n_parts, hi = n * (part + 1) / n_parts;
		out.erase(out.begin() + out0 + hi, out.end()); out.erase(out.begin() + out0, out.begin() + out0 + lo);
	}
	if(keep_text) { for(HSeq &q : out) { if(q.t_id == -1 && (q.t_len != 0 || q.seq.empty())) q.t_id = text_id; } keep_text->swap(data); }          /* (records of this file only: those of earlier files carry their id already) */
	return true;
}

/* `at` is a multiple of 32 for every caller (reads start on 64-base boundaries): whole words are built and stored, no read-modify-write */
void pack_bases(const uint8_t *b, uint64_t n, std::vector<uint32_t> &pk, std::vector<uint32_t> &nm, uint64_t at)
{
	uint32_t *pw = pk.data() + (at >> 4), *nw = nm.data() + (at >> 5);
	uint64_t i = 0;
	for(; i + 32 <= n; i += 32) {
		uint32_t w0 = 0, w1 = 0, m = 0;
		for(int j = 0; j < 16; j++) { uint32_t c = b[i + j]; m |= (uint32_t)(c > 3) << j; w0 |= (c & 3 & (uint32_t)-(int32_t)(c <= 3)) << (2 * j); }
		for(int j = 0; j < 16; j++) { uint32_t c = b[i + 16 + j]; m |= (uint32_t)(c > 3) << (16 + j); w1 |= (c & 3 & (uint32_t)-(int32_t)(c <= 3)) << (2 * j); }
		pw[0] = w0; pw[1] = w1; nw[0] = m; pw += 2; nw += 1;
	}
	for(; i < n; i++) {
		uint64_t p = at + i; uint32_t c = b[i];
		if(c > 3) { nm[p >> 5] |= 1u << (p & 31); c = 0; }
		pk[p >> 4] |= c << (2 * (p & 15));
	}
}

/* ---------------------------------------------------------------------------------------------
 * minimizer sketch on the host (index construction only; reads are sketched on the device).
 * Same emission rule as the device kernel: window minimum over the last w positions of
 * h = hash << 8 | (pos mod w) | strand << 7, emitted when the current position is the minimum or the minimum changed.
 * --------------------------------------------------------------------------------------------- */
uint32_t h_crc32c(uint32_t crc, uint64_t v) { for(int i = 0; i < 64; i++) { uint32_t b = (crc ^ (uint32_t)(v >> i)) & 1u; crc = (crc >> 1) ^ (b ? 0x82f63b78u : 0u); } return crc; }
struct HMin { uint64_t hash; uint32_t pos; uint32_t strand; };
/* minimizers of positions [begin, end) of a sequence.  A stretch that does not start at 0 warms the registers up over the k + 1 + w
 * positions in front of it: an N stops influencing k1 after k + 1 bases, the window holds w positions, u is the previous minimum --
 * so the stretches of a long sequence can be sketched independently and joined in order. */
void sketch_host(const uint8_t *seq, uint32_t len, uint32_t k, uint32_t w, std::vector<HMin> &out, uint32_t begin = 0, uint32_t end = 0xffffffffu)
{
	const uint64_t mask = (1ull << 2 * k) - 1; const int sh = 2 * (k - 1);
	/* minimum of h over the last w positions: a monotone queue (ascending h from head to tail; h carries its position mod w in the low bits, so no two
	 * entries of a window are equal) -- O(1) per base instead of a scan of the window */
	uint64_t qh[64]; uint32_t qp[64]; uint32_t qb = 0, qe = 0;          /* ring of at most w <= 31 live entries */
	uint64_t k0 = 0, k1 = 0, u = 0;
	const uint32_t p0 = begin > k + 1 + w ? begin - (k + 1 + w) : 0;
	if(end > len) end = len;
	for(uint32_t p = p0; p < end; p++) {
		uint64_t c = seq[p];
		k0 = (k0 << 2 | c) & mask; k1 = (k1 >> 2) | ((3ull ^ c) << sh);
		if(p + 1 < k || p < p0 + (p0 ? k : 0)) continue;
		uint64_t km = k0 < k1 ? k0 : k1, kx = k0 < k1 ? k1 : k0, m = k0 < k1 ? 0 : 0x80;
		uint64_t crc = (kx >> 32) ? (uint64_t)h_crc32c((uint32_t)kx, kx) : 0ull;
		uint32_t i = (p - (k - 1)) % w;
		uint64_t h = ((crc ^ km) & mask) << 8 | i | m;
		while(qe != qb && qh[(qe - 1) & 63] > h) qe--;                 /* entries that can never be the minimum again */
		qh[qe & 63] = h; qp[qe & 63] = p; qe++;
		while(qp[qb & 63] + w <= p) qb++;                             /* entries that left the window */
		const uint64_t v = qh[qb & 63];
		if(p >= begin && (v == h || v != u)) {
			uint32_t iv = (uint32_t)(v & 0x7f);
			out.push_back(HMin{ v >> 8, (p - (k - 1)) - ((i + w - iv) % w), (uint32_t)((v >> 7) & 1) });
		}
		u = v;
	}
}

/* a circular reference sequence (-c): the reference runs its window on over the first min(len, w) bases (mm_sketch_cap, minialign.c:2438-2444) from the
 * state its block-wise loop was left in, and decodes positions from the local indices of the stream (minialign.c:2831-2835) -- so the wrap-around
 * minimizers carry the positions that decoder gives them.  Reproduced by walking the sequence the way mm_sketch does (blocks of w with a backward-min
 * array, minialign.c:2410-2435); circular sequences are chromosomes / plasmids, one host thread each. */
void sketch_host_circular(const uint8_t *seq, uint32_t len, uint32_t k_, uint32_t w_, std::vector<HMin> &out)
{
	const uint64_t kk = k_ - 1, shift1 = 2 * kk, mask = (1ull << 2 * k_) - 1, w = w_;
	uint64_t r[64]; for(int i = 0; i < 64; i++) r[i] = ~0ull;
	const uint8_t *p = seq, *t = seq + len;
	uint64_t u = 0, k0 = 0, k1 = 0;
	uint64_t base = (uint64_t)-(int64_t)w, dv = w;                  /* position decoder state */
	auto push = [&](uint64_t v) { uint64_t lu = v & 0x7f; base += lu <= dv ? w : 0; dv = lu; out.push_back(HMin{ v >> 8, (uint32_t)(base + lu), (uint32_t)((v >> 7) & 1) }); };
	auto kmer = [&]() { uint64_t c = *p++; k0 = (k0 << 2 | c) & mask; k1 = (k1 >> 2) | ((3ull ^ c) << shift1); };
	auto core = [&](uint64_t i, uint64_t &f) -> uint64_t {
		kmer();
		uint64_t km = k0 < k1 ? k0 : k1, kx = k0 < k1 ? k1 : k0, m = k0 < k1 ? 0 : 0x80;
		uint64_t crc = (kx >> 32) ? (uint64_t)h_crc32c((uint32_t)kx, kx) : 0ull;
		uint64_t hh = ((crc ^ km) & mask) << 8 | i | m; f = std::min(f, hh); uint64_t v = std::min(f, r[i + 1]);
		if((v == hh) | (v - u)) push(v);
		u = v; return hh;
	};
	for(uint64_t i = 0; i < kk && p < t; i++) kmer();
	while((int64_t)(t - p) >= (int64_t)w) {
		uint64_t f = ~0ull; for(uint64_t i = 0; i < w; i++) r[i] = core(i, f);
		uint64_t rr = ~0ull; for(uint64_t i = 0; i < w; i++) { rr = std::min(rr, r[w - i - 1]); r[w - i - 1] = rr; }
	}
	const uint64_t l = (uint64_t)(t - p);
	if(l > 0) {
		uint64_t f = ~0ull; for(uint64_t i = 0; i < l; i++) r[w + i] = core(i, f) + w;
		uint64_t rr = ~0ull; for(uint64_t i = 0; i < w; i++) { rr = std::min(rr, r[w + l - i - 1]); r[w + l - i - 1] = rr; }
		for(uint64_t i = 0; i < w; i++) r[i] = r[l + i] - l;
		u += w - l;
	}
	p = seq; t = seq + std::min<uint64_t>(len, w);
	{ uint64_t f = ~0ull; for(uint64_t i = 0; i < w && p < t; i++) core(i, f); }
}

} /* anonymous */

#include "host_options.hpp"
#include "host_index.hpp"
/* =============================================================================================
 * device context + batch pipeline
 * ============================================================================================= */

/* What a batch will leave as the carried reference length, as far as its chains tell (the prediction run_rounds hands from read to read inside a batch), posted by the lane
 * that has the batch as soon as its chaining is through, for the lane with the NEXT batch: that one's first read otherwise starts with the value the stream had when the batch
 * was taken -- three or four batches back, the wrong contig's length 24 times out of 25 -- and whenever that flips its `apos >= rlen` test the read is mapped again in a launch
 * of its own behind the check, with every lane behind waiting (12 of the 43 batches of a headline step, 55 - 120 ms each).  A guess like any other: batch_verify_carry checks it. */
struct PredBoard {
	std::mutex mu; std::condition_variable cv; std::unordered_map<uint32_t, uint32_t> out; uint32_t gone_below = 0; std::unordered_map<uint32_t, bool> gone;
	void post(uint32_t k, uint32_t v) { { std::lock_guard<std::mutex> lk(mu); out[k] = v; } cv.notify_all(); }
	void leave(uint32_t k) { { std::lock_guard<std::mutex> lk(mu); gone[k] = true; } cv.notify_all(); }          /* batch k will post nothing (split, failed) */
	/* the value batch k posted; false when it has not within `ms` (or never will): the caller keeps its own guess */
	bool get(uint32_t k, uint32_t *v, int ms) {
		std::unique_lock<std::mutex> lk(mu);
		cv.wait_for(lk, std::chrono::milliseconds(ms), [&]() { return out.count(k) != 0 || gone.count(k) != 0; });
		auto it = out.find(k); if(it == out.end()) return false; *v = it->second; return true;
	}
};
struct mm_align_s {
	mm_opt_s o; const mm_idx_s *mi;
	gaba_t *gctx;
	DevIndex dix; IdxSlot *d_slot = nullptr; uint64_t *d_val = nullptr; uint32_t *d_seq_len = nullptr; uint64_t *d_seq_off = nullptr; uint8_t *d_seq_circ = nullptr;
	gaba_arena_t *ref_ar = nullptr;
	uint32_t twlen, tglen; double mcoef, xcoef;
	hipStream_t stream; hipEvent_t ev0, ev1;
	bool k4_off = false;          /* a text stream of fewer than two batches per lane: K4's walk (5 - 15 ms of dependent steps behind a batch's extension launch) has nothing to hide behind, the host's threads are idle anyway -- the strings are made there (align_text) */
	hipEvent_t k4e = nullptr; bool k4_pending = false;          /* K4 (mm_cigar.hpp) runs on the last side stream behind every extension launch: recorded behind the last one queued */
	hipStream_t k2s[12]; hipEvent_t k2e[16]; bool k2s_ok = false;    /* side streams: the size classes of the sort + chain stage run concurrently */
	uint32_t n_waves = 0;
	uint64_t mem_for_batches = 0;                              /* device memory the lanes' pools may take together (measured when the first text stream starts: batch_cap_bases) */
	uint32_t qlen_hint = 0;                                    /* longest read of the input being mapped, when known (mm_align_file) */
	uint32_t k3_waves = 0; uint64_t slab_stride = 0;      /* extension kernel: persistent waves actually launched and the DP workspace of each */
	/* pools */
	DBuf<uint32_t> q_pk, q_nm; DBuf<ReadIn> d_in; DBuf<ReadState> d_st; DBuf<uint32_t> d_work;
	DBuf<MinRec> min_pool; DBuf<Seed> seed_pool; DBuf<Resc> resc_pool; DBuf<Root> root_pool;
	bool rerun_heavy = false;              /* the re-run at hand has a read that walked many chains the last time (batch_verify_carry -> run_rounds) */
	uint64_t min_over_base = 0, min_over_n = 0;          /* the overflow region of the sketch kernel inside min_pool (ensure_pools) */
	DBuf<uint32_t> rs_scratch; DBuf<uint8_t> slabs; DBuf<KhSlot> kh_pool; DBuf<uint64_t> next_pool;
	DBuf<uint64_t> bin_pool; DBuf<AlnRec> aln_pool; DBuf<gaba::Segment> seg_pool; DBuf<uint32_t> path_pool;
	DBuf<uint32_t> d_k2cnt;                /* work-list cursors of the sort + chain launches */
	DBuf<uint8_t> k2w_scratch;          /* lane-per-read chain sweep: leaf / chain scratch, 16 B per seed found (a third of 16 B per element of the seed pool) */
	DBuf<SpecJob> rq_jobs; DBuf<SpecMemo> rq_memo; DBuf<uint32_t> rq_state;          /* retry jobs of a launch (K3Args.rjobs); rq_state: one word per slot, then the four control words */
	DBuf<SpecJob> spec_jobs; DBuf<SpecMemo> spec_memo; DBuf<uint32_t> spec_path; DBuf<gaba::Segment> spec_seg; DBuf<unsigned long long> spec_top;      /* chain jobs of the heaviest reads of a launch (K3Args.jobs) */
	DBuf<CigItem> cig_items; DBuf<CigEnt> cig_ent; DBuf<char> cig_text;          /* K4 (mm_cigar.hpp): the CIGAR strings of a batch made on the device: where the string of a segment slot stands, the text (cursors: d_tops[36 ..]) */
	DBuf<uint64_t> tap_words;              /* mm_batch_tap: the minimizer stream words of the batch, parallel to min_pool */
	DBuf<uint8_t> d_text, d_codes; DBuf<TextRead> d_tinfo; DBuf<uint32_t> d_tn;      /* packing on the device: text range of the batch, per-read extents, code bytes of the arena, bases found per read */
	/* shared DP workspaces (streaming engine): owned by the primary context, used by every lane; see K3Args.ring */
	mm_align_s *root = nullptr;            /* the primary context of a lane (NULL on the primary itself) */
	bool shared_slabs = false; DBuf<uint32_t> slab_ring; DBuf<unsigned long long> slab_ring_ctr; uint32_t slab_ring_n = 0;
	/* the classes above the ordinary one (reads of more than 32 k bases), when the input has a long tail of lengths; h_cls[0] describes slabs / slab_ring */
	static const uint32_t MAX_CLS = 8;
	DBuf<uint8_t> xslabs[MAX_CLS]; DBuf<uint32_t> xring[MAX_CLS]; DBuf<unsigned long long> xctr[MAX_CLS]; std::vector<K3Class> h_cls; DBuf<K3Class> d_cls;          /* h_cls: the classes as lane 0 sees them; d_cls: slab_lanes tables of h_cls.size() entries, one per lane (its own rings) */
	DBuf<uint32_t> pring[MAX_CLS]; DBuf<unsigned long long> pctr[MAX_CLS]; uint32_t slab_lanes = 0; int lane_ix = 0;          /* the lanes' own rings of every class (K3Class.pring / pctr: lane-major); lanes the workspaces were made for; this context's place among the lanes of its device */ uint32_t slab_total = 0; uint64_t slab_max = 0;
	DBuf<unsigned long long> d_tops;       /* [0] seed [1] resc [2] root [3] bin [4] aln [5] seg [6] path [8..16) stats [16] counter */
	uint32_t rlen_carry = 0;               /* self->rlen of the reference's thread buffer, carried across reads (and batches) */
	struct PredBoard *pred = nullptr; uint32_t pred_k = 0;          /* the streaming engine's board of predicted carried values and the number of the batch this lane has in hand (stream_map; NULL: none) */
	/* reusable host buffers of the streaming engine (primary context only): pinned result buffers for the D2H of a batch, text pieces with their capacity */
	struct PinSet { void *p[7] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }; size_t cap[7] = { 0, 0, 0, 0, 0, 0, 0 };
		void *get(int i, size_t bytes) { if(bytes > cap[i]) { if(p[i]) (void)hipHostFree(p[i]); p[i] = nullptr; cap[i] = 0; size_t want = bytes + bytes / 4 + (1u << 20); if(hipHostMalloc(&p[i], want, hipHostMallocPortable) != hipSuccess) return nullptr; cap[i] = want; } return p[i]; }          /* (portable: a set is pooled on the first context and handed to lanes of any device) */
		~PinSet() { for(int i = 0; i < 7; i++) if(p[i]) (void)hipHostFree(p[i]); } };
	std::vector<PinSet *> pin_free; std::vector<std::vector<std::string>> piece_free; std::mutex pool_mu;
	/* the head of the last stream mapped through this context (mm_map_*): what decides whether another carried value at its start changes anything */
	struct HeadRec { uint32_t apos0, cond0, used, rid_last; };
	std::vector<HeadRec> head; uint32_t head_carry_in = 0;
	std::vector<uint64_t> head_txt; uint64_t head_txt_end = 0;      /* for a stream over a text (mm_map_text / mm_map_file): where the record of read i starts in that text; the length of the text */
	std::vector<uint64_t> head_off; bool head_off_closed = false;      /* byte offset of the first record of read i in the text of that stream (one more entry = the end, when the stream is shorter than the head) */
	bool streaming = false;                /* stream_map is running on this context (the shared workspaces cannot be re-sized then) */
	struct ChunkPool *chunk_pool = nullptr; /* device buffers of the text reader (primary context) */
	mm_align_s *sib = nullptr;             /* second lane: own streams and pools, shares index / reference / DP constants (see mm_batch_run) */
	bool is_sib = false; int dev = 0; bool own_index = true;
	/* the other devices of the node: one primary context each (own streams, lanes, pools, DP workspaces, a replica of the index), owned by the first context.  The
	 * streaming engine deals its batches over all of them; the per-batch entries stay on the first */
	std::vector<mm_align_s *> peers;
	std::vector<int> span; std::mutex span_mu; bool span_failed = false;          /* the devices this context was made to span (mm_align_init); the peers are made when first needed (ensure_peers) */
	/* what the batches of this device have asked of the seed / rescue / chain-root pools so far, in entries per base of batch (primary context; the lanes note it after
	 * every sketch launch, note_demand): K1 counts a read's hits before it claims room for them (minialign.c:3454-3540 is what a read can emit), so the launch leaves the exact
	 * demand in the pool cursors -- the pools are sized for what the run has seen (+ a quarter), not for the caps a read could reach, and a batch that asks for more
	 * has its pools sized to its demand and its sketch launch run again (run_rounds) */
	std::mutex need_mu; double need_seed = 0.30, need_resc = 0.03, need_root = 0.16; uint64_t batch_bases = 0;
	/* experiment (MM_K3_CONCURRENT=n): at most n extension launches of this device in flight at a time, the lanes queue for their turn */
	std::mutex k3_gate_mu; std::condition_variable k3_gate_cv; int k3_in_flight = 0;
	/* the watchdog of the extension launches (primary context; k3_watchdog): every launch of a device goes through the gate above; a launch that lasts longer than the
	 * deadline gets the census of its waves printed (wd: the lane's window into its launch, K3Args.wd) and is called off with every other launch of the device in flight,
	 * the workspace rings are set up again, and the batches run again in the safe mode (no jobs, no waiting for carried values inside a launch, one extension launch at a
	 * time: nothing left in the kernel that waits for another wave but the ring of a scarce workspace class, whose holders then wait for nobody) until the stream ends */
	std::thread wd_thread; bool wd_running = false, wd_stop = false, wd_recovering = false; std::atomic<bool> safe_mode{false}; double wd_longest_ms = 0;
	uint32_t *wd = nullptr; std::atomic<double> k3_t0{0.0}; std::atomic<bool> k3_called_off{false}; uint32_t k3_wd_waves = 0, k3_wd_work = 0;          /* per lane: the pinned window, when its launch in flight began (0: none), whether it was called off, its shape */
	mm_stats_t st; double t_wall0;
	/* knobs (grown on overflow) */
	uint32_t bin_cap = 192, aln_cap = 96, kh_cap = 1024, next_cap = 256, rs_stride = 512 + 3 * 1024;
	void *pin_stage = nullptr; size_t pin_stage_cap = 0;      /* pinned staging buffer of the lane: the per-read state records and the packed reads cross PCIe through it (one DMA each instead of a train of staged blits) */
	bool tap_stop = false;                 /* mm_batch_tap: stop behind the sort + chain stage of the first round */
	unsigned long long *pin_note = nullptr;    /* 64 bytes of pinned host memory the sketch kernel writes the pool cursors to (K1Args.note) */
	std::vector<uint32_t> np0, wp0;            /* diagnostics (MM_VERBOSE): passing chains and their summed length of every read as the first chaining left them (run_rounds -> batch_run_spec) */
	std::vector<uint32_t> ran_with;            /* the carried reference length every read of the batch at hand was handed before its extension launch (run_rounds) */
	uint32_t k2_leaf_shift = 2;            /* leaf area of the first chaining attempt: (n + 1) >> shift; lowered when more than 2 % of a batch had to be retried */
};

/* every context that belongs to `a`: its own lanes, then the primaries of the other devices and their lanes */
template<typename F> static void each_context(mm_align_t *a, F fn)
{
	for(mm_align_t *q = a; q; q = q->sib) fn(q);
	for(mm_align_t *p : a->peers) for(mm_align_t *q = p; q; q = q->sib) fn(q);
}
namespace {


struct BatchOut {                       /* host copies of what post-map / SAM need */
	std::vector<ReadState> st; std::vector<Root> root; std::vector<uint64_t> bin; std::vector<AlnRec> aln;
	std::vector<gaba::Segment> seg; std::vector<uint32_t> path;
};

/* copies / memsets of the run path go to the lane's own (non-blocking) stream and wait on that stream only: the legacy default
 * stream would serialise every lane against every other */
#define CPY(_a, _dst, _src, _n, _kind) do { CK(hipMemcpyAsync((_dst), (_src), (_n), (_kind), (_a)->stream)); CK(hipStreamSynchronize((_a)->stream)); } while(0)
#define MM_SIDE 2          /* side streams per lane: every stream wants a hardware queue of its own (streams that share one run in order: the extension kernel of one lane
                            * then blocks the sort + chain launches of another), and with GPU_MAX_HW_QUEUES=16 five lanes of 1 + 2 streams still get one each */
#define CK(_e) do { hipError_t _r = (_e); if(_r != hipSuccess) { fprintf(stderr, "[minialign_amd] HIP error %s at %s:%d\n", hipGetErrorString(_r), __FILE__, __LINE__); return false; } } while(0)

inline void *lane_stage(mm_align_t *a, size_t bytes)
{
	if(bytes > a->pin_stage_cap) { if(a->pin_stage) (void)hipHostFree(a->pin_stage); a->pin_stage = nullptr; a->pin_stage_cap = 0; size_t want = bytes + bytes / 4 + (1u << 20); if(hipHostMalloc(&a->pin_stage, want, hipHostMallocPortable) != hipSuccess) return nullptr; a->pin_stage_cap = want; }
	return a->pin_stage;
}
/* host <-> device copies of a lane through its pinned staging buffer (falls back to the plain copy when pinning fails) */
bool lane_d2h(mm_align_t *a, void *dst, const void *src, size_t n)
{
	void *st = lane_stage(a, n);
	if(!st) { CPY(a, dst, src, n, hipMemcpyDeviceToHost); return true; }
	CPY(a, st, src, n, hipMemcpyDeviceToHost); memcpy(dst, st, n); return true;
}
bool lane_h2d(mm_align_t *a, void *dst, const void *src, size_t n)
{
	/* (K4 of the extension launch in front may still be walking the states and pools on its side stream: whatever goes up -- the states of the reads a carried-value
	 * re-run maps again, batch_verify_carry -- goes up behind it) */
	if(a->k4_pending) { CK(hipStreamWaitEvent(a->stream, a->k4e, 0)); }
	void *st = lane_stage(a, n);
	if(!st) { CPY(a, dst, src, n, hipMemcpyHostToDevice); return true; }
	memcpy(st, src, n); CPY(a, dst, st, n, hipMemcpyHostToDevice); return true;
}
void k3_watchdog_start(mm_align_s *GP);
/* the CIGAR strings of this context's output are made on the device (K4): SAM without MD tags; MM_HOST_CIGAR: the host walks the path words as in rounds 1-5 */
static bool device_cigar(const mm_align_s *a) { static const bool host_cigar = getenv("MM_HOST_CIGAR") != NULL; return a->o.format == 0 && !((a->o.ptags() >> 8) & 1) && !host_cigar && !a->k4_off; }
/* run K1..K3 over `work` (indices into the batch) with rlen_in already stored in d_st[].rlen */
bool run_rounds(mm_align_t *a, uint32_t n_reads, const std::vector<uint32_t> &work_in, bool run_k1, std::vector<ReadState> &hst,
	const std::vector<uint32_t> *rlen_fixed, const std::vector<uint32_t> &qlens)
{
	std::vector<uint32_t> work(work_in);
	if(work.empty()) return true;
	unsigned long long *tops = a->d_tops.p;
	float ms;
	mm_align_s *const GPw = a->root ? a->root : a;
	const bool safe = GPw->safe_mode.load();          /* the watchdog has called a launch of this device off (k3_watchdog_main): nothing in the extension launches that waits for another wave */
	a->k3_called_off.store(false);
	if(a->k4_pending) { CK(hipStreamWaitEvent(a->stream, a->k4e, 0)); }          /* (K4 of the launch in front still reads the states and pools this one is about to change) */
	/* a re-run of a few reads with the carried value given (batch_verify_carry): the short way -- one sort + chain launch, no round trip of the states between chaining
	 * and extension (the caller has put the value and the reset fields in place), the work list as it is */
	const bool small_rerun = rlen_fixed != nullptr && run_k1 && work.size() < 256;
	if(run_k1) {
		CK(hipMemcpyAsync(a->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, a->stream));
		CK(hipMemsetAsync(tops + 16, 0, 8, a->stream));
		K1Args k1; k1.idx = a->dix; k1.qar = gaba::SeqArena{ a->q_pk.p, a->q_nm.p }; k1.in = a->d_in.p; k1.st = a->d_st.p; k1.n_reads = (uint32_t)work.size();
		k1.min_pool = a->min_pool.p; k1.seed_pool = a->seed_pool.p; k1.seed_pool_cap = a->seed_pool.n; k1.seed_top = tops + 0;
		k1.resc_pool = a->resc_pool.p; k1.resc_pool_cap = a->resc_pool.n; k1.resc_top = tops + 1;
		k1.root_pool = a->root_pool.p; k1.root_pool_cap = a->root_pool.n; k1.root_top = tops + 2;
		k1.counter = (uint32_t *)(tops + 16); k1.stats = tops + 8; k1.work = a->d_work.p; k1.tap = nullptr; k1.note = a->pin_note;
		/* the overflow region for the minimizer records of low-complexity reads: behind the reads' own shares (ensure_pools), its cursor reset with every sketch launch */
		k1.min_over_top = nullptr; k1.min_over_base = 0; k1.min_over_cap = 0;
		if(a->min_over_n && a->min_over_base + a->min_over_n <= a->min_pool.n) { k1.min_over_top = tops + 32; k1.min_over_base = a->min_over_base; k1.min_over_cap = a->min_over_n; CK(hipMemsetAsync(tops + 32, 0, 8, a->stream)); }
		if(a->pin_note) { a->pin_note[0] = a->pin_note[1] = a->pin_note[2] = ~0ull; }
		if(a->tap_stop) { if(!a->tap_words.ensure(a->min_pool.n)) return false; k1.tap = a->tap_words.p; }
		uint32_t waves = std::min<uint32_t>(a->n_waves, (uint32_t)((work.size() + 3) & ~3ull));
		CK(hipEventRecord(a->ev0, a->stream));
		hipLaunchKernelGGL(mm_sketch_seed_kernel, dim3(waves / 4), dim3(256), 0, a->stream, k1);
		CK(hipGetLastError()); CK(hipEventRecord(a->ev1, a->stream)); CK(hipEventSynchronize(a->ev1));
		CK(hipEventElapsedTime(&ms, a->ev0, a->ev1)); a->st.k1_ms += ms; a->st.k1_launches++;
		if(!rlen_fixed && a->batch_bases) {
			/* the first sketch launch of a batch: the pool cursors hold what its reads asked for, exactly (a read counts its hits before it claims room).  Noted for
			 * the batches to come; and a batch that asked for more than the pools hold gets pools of its size and the launch again -- a few milliseconds, early in a run */
			unsigned long long t3[3] = { a->pin_note[0], a->pin_note[1], a->pin_note[2] };          /* (written by the last wave of the launch, which is over) */
			if(t3[0] == ~0ull) { CPY(a, t3, tops, sizeof(t3), hipMemcpyDeviceToHost); }
			mm_align_s *NP = a->root ? a->root : a; const double bb = (double)a->batch_bases;
			/* (noted from batches of 16 Mb and more only, and clamped: the ratio of one short repeat-family read -- a map_split half, the tail batch of a file, a single read through
			 * mm_align_seq -- would otherwise size every later 300 Mb batch of the context at tens of GB per lane; a small batch that asks for more than it got is regrown below all the same) */
			if(a->batch_bases >= (16ull << 20)) { std::lock_guard<std::mutex> lk(NP->need_mu); NP->need_seed = std::min(4.0, std::max(NP->need_seed, t3[0] / bb)); NP->need_resc = std::min(1.0, std::max(NP->need_resc, t3[1] / bb)); NP->need_root = std::min(2.0, std::max(NP->need_root, t3[2] / bb)); }
			if(t3[0] > a->seed_pool.n || t3[1] > a->resc_pool.n || t3[2] > a->root_pool.n) {
				if(getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd]   the batch asks for %.1f / %.1f / %.1f M seed / rescue / root entries (%.3f / %.3f / %.3f per base), the pools hold %.1f / %.1f / %.1f M: sized again, sketch launch repeated\n",
					t3[0] * 1e-6, t3[1] * 1e-6, t3[2] * 1e-6, t3[0] / bb, t3[1] / bb, t3[2] / bb, a->seed_pool.n * 1e-6, a->resc_pool.n * 1e-6, a->root_pool.n * 1e-6);
				auto grow = [](uint64_t need) -> uint64_t { const uint64_t w = need + need / 8 + (1ull << 20), q = w < (64ull << 20) ? (1ull << 20) : (16ull << 20); return (w + q - 1) / q * q; };
				if((t3[0] > a->seed_pool.n && !a->seed_pool.ensure(grow(t3[0]))) || (t3[1] > a->resc_pool.n && !a->resc_pool.ensure(grow(t3[1]))) || (t3[2] > a->root_pool.n && !a->root_pool.ensure(grow(t3[2])))) return false;
				if(!a->k2w_scratch.ensure(a->seed_pool.n * 8)) return false;
				a->st.pool_regrows++;
				if(!lane_h2d(a, a->d_st.p, hst.data(), (uint64_t)n_reads * sizeof(ReadState))) return false;          /* (the states as they were before the launch) */
				CK(hipMemsetAsync(tops, 0, 3 * 8, a->stream)); CK(hipMemsetAsync(tops + 8, 0, 2 * 8, a->stream)); CK(hipMemsetAsync(tops + 16, 0, 8, a->stream));
				k1.seed_pool = a->seed_pool.p; k1.seed_pool_cap = a->seed_pool.n; k1.resc_pool = a->resc_pool.p; k1.resc_pool_cap = a->resc_pool.n; k1.root_pool = a->root_pool.p; k1.root_pool_cap = a->root_pool.n;
				k1.note = nullptr;
				if(k1.min_over_top) { CK(hipMemsetAsync(tops + 32, 0, 8, a->stream)); }
				CK(hipEventRecord(a->ev0, a->stream));
				hipLaunchKernelGGL(mm_sketch_seed_kernel, dim3(waves / 4), dim3(256), 0, a->stream, k1);
				CK(hipGetLastError()); CK(hipEventRecord(a->ev1, a->stream)); CK(hipEventSynchronize(a->ev1));
				CK(hipEventElapsedTime(&ms, a->ev0, a->ev1)); a->st.k1_ms += ms; a->st.k1_launches++;
			}
		}
	}
	const bool deferred = false;          /* (round 4's experiment -- the later rounds of rescue-heavy reads as launches of their own -- lost to the round jobs of round 5: HISTORY.md) */
	for(uint32_t round = 0; round < a->mi->n_occ && !work.empty(); round++) {
		CK(hipMemcpyAsync(a->d_work.p, work.data(), work.size() * 4, hipMemcpyHostToDevice, a->stream));
		CK(hipEventRecord(a->ev0, a->stream));
		if(round == 0) {
			/* wave-per-read kernel with the seed array in LDS.  Reads are split into size classes by the LDS they need, one launch
			 * per class (largest first), so that small reads run at 5 blocks per CU while the few large ones still stay on chip;
			 * whatever exceeds 160 KB is sorted in place in HBM by the same code. */
			static const uint32_t cls_div[] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12 };     /* blocks per CU of each class; 0 = HBM */
			const int n_cls = (int)(sizeof(cls_div) / sizeof(cls_div[0]));
			K2aArgs ka; ka.st = a->d_st.p; ka.work = a->d_work.p; ka.n_work = (uint32_t)work.size(); ka.seed_pool = a->seed_pool.p; ka.root_pool = a->root_pool.p;
			ka.prof = tops + 24; ka.twlen = a->twlen; ka.mcoef = a->mcoef; ka.min_score = a->o.min_score; ka.seq_len = a->dix.seq_len; ka.seq_circ = a->dix.seq_circ;
			CK(hipMemsetAsync(a->d_k2cnt.p, 0, 32 * 4, a->stream));
			CK(hipEventRecord(a->ev0, a->stream));          /* re-recorded behind the memset: the side streams start from here */
			/* the sort first (mm_sort_kernel: 4 B of LDS per seed, a dozen reads per CU), in size classes by LDS need on the side streams; every stream of the
			 * chain launches below then waits for all of them (a read's sort class is not its chain class) */
			/* the few reads of a re-run (the carried-value check): ONE launch, sort + chain in place in HBM (the form the largest reads take) -- the nine launches of the
			 * size classes each wait 10 - 40 ms for a wave slot beside the extension waves of the other lanes, and every lane behind this one waits for the check */
			const bool presort = !small_rerun;
			if(small_rerun) {
				ka.presorted = 0; ka.leaf_shift = a->k2_leaf_shift ? 1u : 0u; ka.big_only = 0; ka.retry = 0; ka.lds_bytes = 1536 * 4; ka.n_lo = 0; ka.n_hi = 0xffffffffu; ka.counter = a->d_k2cnt.p + 12;
				hipLaunchKernelGGL(mm_sort_chain_lds_kernel, dim3((uint32_t)work.size()), dim3(64), 1536 * 4, a->stream, ka);
				CK(hipGetLastError());
			}
			if(presort) {
				static const uint32_t s_kb[] = { 10, 14, 20, 32, 64, 104 };          /* k2s_bytes(K2S_MAX_N) <= 104 KB */
				const int n_s = (int)(sizeof(s_kb) / sizeof(s_kb[0]));
				K2sArgs ks; ks.st = a->d_st.p; ks.work = a->d_work.p; ks.n_work = (uint32_t)work.size(); ks.seed_pool = a->seed_pool.p; ks.prof = tops + 28;
				{ const size_t ns = a->mi->seq.size(); ks.start_shift = (ns <= 1 ? 24u : (ns <= 256 ? 32u : (ns <= 65536 ? 40u : 56u))); }
				const uint32_t n_cu = a->n_waves / (4 * MM_K3_WAVES_PER_SIMD);
				for(int si = 0; si < n_s; si++) {
					ks.lds_bytes = s_kb[si] * 1024u; ks.n_lo = si ? s_kb[si - 1] * 1024u : 0u; ks.n_hi = ks.lds_bytes; ks.counter = a->d_k2cnt.p + 16 + si;
					const uint32_t per_cu = std::min<uint32_t>(16u, 160u / s_kb[si]);
					const uint32_t grid = std::min<uint32_t>((uint32_t)work.size(), n_cu * per_cu);
					hipStream_t sq = a->k2s[si % MM_SIDE];
					if(si < MM_SIDE) { CK(hipStreamWaitEvent(sq, a->ev0, 0)); }
					hipLaunchKernelGGL(mm_sort_kernel, dim3(grid), dim3(64), ks.lds_bytes, sq, ks);
					CK(hipGetLastError());
				}
				for(int j = 0; j < MM_SIDE; j++) { CK(hipEventRecord(a->k2e[12 + j], a->k2s[j])); }
				for(int j = 0; j < MM_SIDE; j++) { for(int i = 0; i < MM_SIDE; i++) { if(i != j) { CK(hipStreamWaitEvent(a->k2s[j], a->k2e[12 + i], 0)); } } }
			}
			ka.presorted = presort ? 1u : 0u; ka.leaf_shift = a->k2_leaf_shift; ka.big_only = 0;
			if(presort) {
				/* chaining in two launches: the window scans of all seeds at full occupancy (mm_chain_scan_kernel, no LDS), then the sequential sweep on a
				 * compact LDS image of each read (mm_chain_kernel, size classes by LDS need); what the old kernel is left with are reads too large for these */
				const uint32_t n_cu = a->n_waves / (4 * MM_K3_WAVES_PER_SIMD);
				K2pArgs kp; kp.st = a->d_st.p; kp.work = a->d_work.p; kp.n_work = (uint32_t)work.size(); kp.seed_pool = a->seed_pool.p; kp.twlen = a->twlen; kp.counter = a->d_k2cnt.p + 24; kp.prof = tops + 24;
				hipLaunchKernelGGL(mm_chain_scan_kernel, dim3(std::min<uint32_t>((uint32_t)work.size(), n_cu * 32u)), dim3(64), 0, a->k2s[0], kp);
				CK(hipGetLastError());
				CK(hipEventRecord(a->k2e[14], a->k2s[0]));
				for(int j = 1; j < MM_SIDE; j++) { CK(hipStreamWaitEvent(a->k2s[j], a->k2e[14], 0)); }
				{
					/* the sweep with one lane per read in HBM (mm_chain_sweep_kernel): every read of the batch in flight at once, no LDS, a few hundred waves */
					K2wArgs kw; kw.st = a->d_st.p; kw.work = a->d_work.p; kw.n_work = (uint32_t)work.size(); kw.seed_pool = a->seed_pool.p; kw.root_pool = a->root_pool.p; kw.scratch = a->k2w_scratch.p; kw.scratch_top = tops + 30; kw.scratch_bytes = a->k2w_scratch.bytes;
					CK(hipMemsetAsync(tops + 30, 0, 8, a->k2s[0]));
					kw.mcoef = a->mcoef; kw.min_score = a->o.min_score; kw.twlen = a->twlen; kw.seq_len = a->dix.seq_len; kw.seq_circ = a->dix.seq_circ;
					hipLaunchKernelGGL(mm_chain_sweep_kernel, dim3(((uint32_t)work.size() + 63) / 64), dim3(64), 0, a->k2s[0], kw);
					CK(hipGetLastError());
					CK(hipEventRecord(a->k2e[0], a->k2s[0])); CK(hipStreamWaitEvent(a->stream, a->k2e[0], 0));
				}
				/* reads with more than K2S_MAX_N seeds: the old kernel's in-HBM form, behind everything else on the main stream */
				ka.big_only = 2; ka.retry = 0; ka.leaf_shift = a->k2_leaf_shift ? 1u : 0u; ka.lds_bytes = 1536 * 4; ka.n_lo = 0; ka.n_hi = 0xffffffffu; ka.counter = a->d_k2cnt.p + 12;
				hipLaunchKernelGGL(mm_sort_chain_lds_kernel, dim3(std::min<uint32_t>((uint32_t)work.size(), n_cu)), dim3(64), 1536 * 4, a->stream, ka);
				CK(hipGetLastError());
			}
		}

		CK(hipEventRecord(a->ev1, a->stream)); CK(hipEventSynchronize(a->ev1));
		CK(hipEventElapsedTime(&ms, a->ev0, a->ev1)); a->st.k2_ms += ms; a->st.k2_launches++;

		if(round == 0) {
			/* seed the carried reference-length state (see ReadIn.rlen_in): either given exactly (re-runs), or predicted from
			 * the chain lists: read i starts with the length of the last reference read i - 1 loads */
			if(small_rerun) { /* nothing comes back here, nothing goes up: rlen, apos0, rid_last, bin_off and the counters of the reads are as batch_verify_carry put them */ }
			else {
			if(!lane_d2h(a, hst.data(), a->d_st.p, (uint64_t)n_reads * sizeof(ReadState))) return false;
			if(rlen_fixed) { for(size_t i = 0; i < work.size(); i++) { hst[work[i]].rlen = (*rlen_fixed)[i]; hst[work[i]].dep = gaba::NIL; hst[work[i]].flags = 0; } }
			else {
				/* ... except behind a read that has no chain worth a trial at this threshold but rescue minimizers waiting: what it leaves is whatever its later rounds find, which
				 * nobody knows before they have run.  Such a read is a source (RS_CARRY_SRC): the reads whose starting value it decides (the next read, and on through reads that
				 * load nothing at all) take the value from it inside the launch (ReadState.dep, mm_extend_kernel) instead of running with a guess and being run again from the
				 * sketch on when the check finds the guess wrong -- 71 such re-runs per step of the headline set, 154 on a tenth of the hard-repeat set, where the re-runs of a batch
				 * (one read inside a repeat family alone on a launch: a second) were three quarters of the step; MM_NO_CARRY_DEPS: all by prediction as before */
				static const bool no_deps = getenv("MM_NO_CARRY_DEPS") != NULL;
				const bool deps = !no_deps && !safe;
				std::vector<uint8_t> in_work(n_reads, 0); for(uint32_t wi : work) in_work[wi] = 1;
				const double weak_unit = 128 * 2.0 * (double)a->o.min_score / a->mcoef;          /* (the factor measured at 0 / 128 / 256 / 384 / 1 024 in round 5: HISTORY.md) */
				uint32_t cur = a->rlen_carry, src = gaba::NIL;          /* src: the source read that decides the value at hand */
				if(a->pred && a->pred_k > 0) { uint32_t pv; if(a->pred->get(a->pred_k - 1, &pv, 100)) { cur = pv; } }          /* (what the batch in front expects to leave: PredBoard) */
				a->ran_with.resize(n_reads);
				if(getenv("MM_VERBOSE")) { a->np0.resize(n_reads); a->wp0.resize(n_reads); for(uint32_t i = 0; i < n_reads; i++) { a->np0[i] = hst[i].n_pass; a->wp0[i] = hst[i].w_pass; } }
				for(uint32_t i = 0; i < n_reads; i++) {
					hst[i].rlen = cur; a->ran_with[i] = cur;          /* (what the read runs with, kept as it is handed out: see batch_run_spec) */
					hst[i].dep = src; hst[i].flags = 0; hst[i].carry_ready = 0; hst[i].rlen_in = cur;
					if(!in_work[i]) continue;
					if(hst[i].pred_rid != gaba::NIL) {
						cur = a->mi->seq[hst[i].pred_rid].blen(); src = gaba::NIL;
						/* ... and a read whose passing chains are weak -- together less than 128 times the length a chain needs to be tried: 110 of the 14 400 reads of a headline
						 * batch, and 13 of the 16 whose trials all fail and that go on to the next threshold, where the rescued minimizers change what they leave -- is a source too:
						 * the reads behind it wait for what it really leaves (the prediction stays the guess for the reads further on).  MM_CARRY_WEAK=n: another factor, 0: none */
						if(deps && weak_unit > 0.0 && hst[i].n_resc > 0 && !hst[i].err && (double)hst[i].w_pass < weak_unit) { hst[i].flags = RS_CARRY_SRC; src = i; }
					}
					else if(deps && hst[i].n_resc > 0 && !hst[i].err) { hst[i].flags = RS_CARRY_SRC; src = i; }
				}
				if(a->pred) { a->pred->post(a->pred_k, cur); }
			}
			for(uint32_t wi : work) { hst[wi].apos0 = gaba::NIL; hst[wi].cond0 = 0; hst[wi].rid_last = gaba::NIL; hst[wi].bin_off = ~0ull; hst[wi].n_bin = 0; hst[wi].n_aln = 0; hst[wi].n_res = 0; }
			if(!lane_h2d(a, a->d_st.p, hst.data(), (uint64_t)n_reads * sizeof(ReadState))) return false;
			}
		}
		if(a->tap_stop) { return true; }
		uint32_t n_heavy = 0; uint32_t seg_beg[8], seg_len[8];
		std::vector<uint32_t> by_len(work);          /* (lives until the extension launch is over: its upload below is not waited for on its own) */
		{
			/* longest read first: with ~5 reads per wave the tail of the launch is one read long, so the short ones go last */
			/* (ordering by the chain count, the best predictor of a read's DP work, was tried and is worse: the heaviest reads then run
			 * under full contention from the start and become the critical path) */
			if(qlens.size() == n_reads && !small_rerun) { std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t x, uint32_t y) { return qlens[x] > qlens[y]; }); }
			/* ... except the few reads with by far the most chains (repeats: dozens of extension trials, several times a wave's
			 * fair share of the DP work): one of them is the critical path of the launch, so they start first.  Moving *all* reads
			 * into chain-count order is worse (measured): the bulk of moderately heavy reads then crowds the start. */
			if(by_len.size() >= 256) {
				/* which reads are heavy is known once they are chained: the summed length of the chains that pass the length test of mm_search_load_root (ReadState.w_pass) is
				 * what the extension will walk -- a read inside a repeat family has 5 - 8 such chains and costs as many full alignments, 15 DP vectors per base against 2.
				 * (The number of chains, n_root, says nothing: every read of a human-size reference has 50 - 120 of them, nearly all too short to be tried.) */
				std::vector<uint32_t> nr(by_len.size()); for(size_t i = 0; i < by_len.size(); i++) nr[i] = hst[by_len[i]].w_pass;
				std::nth_element(nr.begin(), nr.begin() + (nr.size() - 1 - nr.size() / 32), nr.end());
				const uint32_t thr = std::max<uint32_t>(nr[nr.size() - 1 - nr.size() / 32], 1);          /* top ~3 % */
				auto mid = std::stable_partition(by_len.begin(), by_len.end(), [&](uint32_t x) { return hst[x].w_pass >= thr && hst[x].n_pass >= 2; });
				std::stable_sort(by_len.begin(), mid, [&](uint32_t x, uint32_t y) { return hst[x].w_pass > hst[y].w_pass; });
				n_heavy = (uint32_t)(mid - by_len.begin());
			}
			{
				/* ... and, in front of everything, the reads that found NO chain worth a trial at the first occurrence threshold but have rescue minimizers waiting (the sources
				 * of the carried value, RS_CARRY_SRC above): they go on to the next thresholds inside the launch -- a read inside a repeat family finds its hundreds of chains
				 * there (published as jobs since round 5) -- and the reads behind them in the batch wait for what they leave, so a wave must have taken every one of them before
				 * any wave can be waiting: the most rescue hits first */
				auto resc_hits = [&](uint32_t x) -> uint32_t { const uint32_t half = hst[x].seed_cap / 2, base = hst[x].seed_n0 + 2; return half > base ? half - base : 0u; };
				std::vector<uint32_t> front;
				for(uint32_t x : by_len) { if(hst[x].flags & RS_CARRY_SRC) front.push_back(x); }
				if(!front.empty() && !deferred && round == 0) {
					std::stable_sort(front.begin(), front.end(), [&](uint32_t x, uint32_t y) { return resc_hits(x) > resc_hits(y); });
					std::vector<uint8_t> is_front(n_reads, 0); for(uint32_t x : front) is_front[x] = 1;
					std::vector<uint32_t> rest; rest.reserve(by_len.size());
					for(uint32_t x : by_len) if(!is_front[x]) rest.push_back(x);
					/* behind the heavy reads, which keep the front of the list and with it the top issue priority (the first 64th, mm_extend_kernel): with the sources in front of
					 * them the headline set lost 3 % -- every read in the first few thousand entries is taken by a wave the moment the launch starts, which is all the waiting needs */
					by_len.assign(rest.begin(), rest.begin() + std::min<size_t>(n_heavy, rest.size())); by_len.insert(by_len.end(), front.begin(), front.end()); by_len.insert(by_len.end(), rest.begin() + std::min<size_t>(n_heavy, rest.size()), rest.end());
				}
			}
			{
				/* with several workspace classes: the reads of a class together, the highest class first (within a class the order made above).  A wave takes reads of
				 * the highest class that has a workspace free and falls through to the ordinary class otherwise (mm_extend_kernel), instead of waiting at the front of
				 * one list where all the long reads are: on the ONT-like set the reads of 32 - 128 kb spent four fifths of their wave time waiting for a workspace, and
				 * the waves that waited held the wave slots of four launches */
				const mm_align_s *P = a->root ? a->root : a;
				for(int c = 0; c < 8; c++) { seg_beg[c] = 0; seg_len[c] = 0; }
				seg_len[0] = (uint32_t)by_len.size();
				if(P->shared_slabs && P->h_cls.size() > 1 && qlens.size() == n_reads) {
					const int n_cls = (int)P->h_cls.size();
					auto cls_of = [&](uint32_t r) { int c = 0; while(c + 1 < n_cls && qlens[r] > P->h_cls[c].qmax) { c++; } return c; };
					std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t x, uint32_t y) { return cls_of(x) > cls_of(y); });
					for(int c = 0; c < 8; c++) seg_len[c] = 0;
					for(uint32_t r : by_len) seg_len[cls_of(r)]++;
					uint32_t at = 0; for(int c = n_cls - 1; c >= 0; c--) { seg_beg[c] = at; at += seg_len[c]; }
				}
			}
			if(!small_rerun || round != 0) { CK(hipMemcpyAsync(a->d_work.p, by_len.data(), by_len.size() * 4, hipMemcpyHostToDevice, a->stream)); }          /* (a small re-run: the list K1 and K2 used) */
		}
		CK(hipMemsetAsync(tops + 16, 0, 8, a->stream));
		K3Args k3; k3.idx = a->dix; k3.gc = a->gctx->hc; k3.roots = a->gctx->droots; k3.ar_ref = gaba::SeqArena{ a->ref_ar->pk, a->ref_ar->nm }; k3.ar_q = gaba::SeqArena{ a->q_pk.p, a->q_nm.p };
		k3.in = a->d_in.p; k3.st = a->d_st.p; k3.work = a->d_work.p; k3.n_work = (uint32_t)work.size();
		k3.seed_pool = a->seed_pool.p; k3.root_pool = a->root_pool.p; k3.slabs = a->slabs.p; k3.slab_bytes = a->slab_stride;
		k3.ring = nullptr; k3.ring_ctr = nullptr; k3.ring_n = 0;
		k3.cls = nullptr; k3.n_cls = 0;
		{ const mm_align_s *P = a->root ? a->root : a; if(P->shared_slabs) { k3.slabs = P->slabs.p; k3.slab_bytes = P->slab_stride; k3.ring = P->slab_ring.p; k3.ring_ctr = P->slab_ring_ctr.p; k3.ring_n = P->slab_ring_n;
			k3.n_cls = (uint32_t)P->h_cls.size(); k3.cls = P->d_cls.p + (size_t)std::min<uint32_t>((uint32_t)a->lane_ix, P->slab_lanes - 1) * k3.n_cls; } }
		k3.kh_pool = a->kh_pool.p; k3.kh_cap = a->kh_cap; k3.kh_top = tops + 7; k3.kh_base = (uint64_t)n_reads * a->kh_cap; k3.kh_pool_cap = a->kh_pool.n; k3.round = round; k3.next_pool = a->next_pool.p; k3.next_cap = a->next_cap;
		k3.bin_pool = a->bin_pool.p; k3.bin_pool_cap = a->bin_pool.n; k3.bin_top = tops + 3; k3.bin_cap_per_read = a->bin_cap;
		k3.aln_pool = a->aln_pool.p; k3.aln_pool_cap = a->aln_pool.n; k3.aln_top = tops + 4; k3.aln_cap_per_read = a->aln_cap;
		k3.seg_pool = a->seg_pool.p; k3.seg_pool_cap = a->seg_pool.n; k3.seg_top = tops + 5;
		k3.path_pool = a->path_pool.p; k3.path_pool_cap = a->path_pool.n; k3.path_top = tops + 6;
		k3.tglen = a->tglen; k3.mcoef = a->mcoef; k3.min_ratio = a->o.min_ratio; k3.min_score = a->o.min_score;
		k3.counter = (uint32_t *)(tops + 16); k3.stats = tops + 8;
		for(int c = 0; c < 8; c++) { k3.seg_beg[c] = seg_beg[c]; k3.seg_len[c] = seg_len[c]; }
		k3.seg_cnt = a->d_k2cnt.p + 32; CK(hipMemsetAsync(k3.seg_cnt, 0, 8 * 4, a->stream));
		/* a read without a result goes on to the next occurrence threshold on the wave that holds it (k3_rescue_round) instead of coming back through the host
		 * for another round of launches: 2.77 against 3.20 s per step on the headline workload (the latency-bound rescue launches -- a serial sort + chain and
		 * an extension launch of some eighty waves, twice -- took half of a lane's time per batch).  MM_K3_HOST_ROUNDS: the rounds as separate launches */
		const bool inkernel = true;          /* (the rounds as launches of their own through the host, rounds 1-2: 3.20 against 2.77 s per headline step -- HISTORY.md) */
		k3.inkernel_rounds = inkernel ? 1u : 0u; k3.resc_pool = a->resc_pool.p; k3.twlen = a->twlen;
		k3.rjobs = nullptr; k3.rmemo = nullptr; k3.rstate = nullptr; k3.rq_cap = 0; k3.rq_ctl = nullptr;
		/* retry jobs: the look-ahead of the reads that try seed after seed of a chain, taken by waves that have run out of reads (K3Args.rjobs) */
		/* (1: chain jobs of the workspace class held.  Retry jobs as well (3) cost the ONT-like set 7 % -- the retry trial of a 100 kb read in front of a wave's own next read -- and
		 * give the hard-repeat set nothing: there the long pole of a launch is the SECOND trial of each of a read's 150 - 270 chains, which its owner runs itself, one after the
		 * other, 0.5 - 1 M DP vectors on one wave -- profiles/round5_hard_read_cost.txt) */
		k3.rq_helper_mask = 127u;          /* one wave in 128 is a helper: 4.17 / 4.40 / 4.56 G bases/s with one in 8 / 32 / 128 (4.45 without) when helpers were the waves that had run out of reads -- the launch is 13 % shorter with any of them, but a helper holds a wave slot the other lanes' short kernels wait for; as helpers from the start, one in 8 / 16 / 32 on the ONT-like set: 2.37 / 2.34 / 2.46 against 2.7 - 2.9 */
		/* (any number of workspace classes, any number of workspaces: a helper takes the workspace a job needs before it claims the job and without waiting, K3_TRY_SLAB, so the
		 * wave that waits for a claimed job waits for one that is running; on the ONT-like set the reads that decide the launch are 60 - 160 kb long with 9 - 27 trials for
		 * one alignment, profiles/round3_ont_read_cost.txt) */
		/* (a launch of a few reads -- the re-runs of the carried-value check -- has them too, with a helper in every workgroup: one read inside a repeat family alone on a launch
		 * walked its hundreds of chains on one wave for a second while the lanes behind it waited for their turn at the carried value) */
		/* (... when one of its reads walked many chains the last time: a launch of 256 workgroups ends when the last of them has had its turn at a wave slot, and the lanes
		 * behind a re-run wait for it -- the ordinary re-run of two or three reads stays a launch of one workgroup) */
		const bool small_launch = work.size() < 256, small_heavy = small_launch && (a->rerun_heavy || !rlen_fixed);
		if(small_heavy) { k3.rq_helper_mask = 3u; }
		if(round == 0 && k3.ring && k3.cls && inkernel && !safe && (!small_launch || small_heavy) && !getenv("MM_K3_NO_RETRY_JOBS")) {
			const uint32_t rq_cap = 1u << 17;
			if(a->rq_jobs.ensure(rq_cap) && a->rq_memo.ensure(rq_cap) && a->rq_state.ensure(rq_cap + 16)) {
				CK(hipMemsetAsync(a->rq_state.p, 0, ((size_t)rq_cap + 16) * 4, a->stream));
				k3.rjobs = a->rq_jobs.p; k3.rmemo = a->rq_memo.p; k3.rstate = a->rq_state.p; k3.rq_cap = rq_cap; k3.rq_ctl = (unsigned int *)(a->rq_state.p + rq_cap);
			}
		}
		k3.jobs = nullptr; k3.memo = nullptr; k3.job_top = nullptr; k3.job_cap = 0; k3.spath = nullptr; k3.spath_cap = 0; k3.sseg = nullptr; k3.sseg_cap = 0; k3.stage_top = nullptr; k3.round_jobs = 0;
		k3.dyn0_min = small_heavy ? 2u : 0u;          /* (a small launch has no chain jobs from before the launch: its reads publish their chains themselves) */
		/* the staging area of the traced jobs (path words, segments) and its cursors: for the chain jobs enumerated before the launch and for the chains a read publishes from
		 * inside it (K3Args.rjobs, JOB_FULL) alike */
		const uint64_t stage_job_cap = getenv("MM_K3_JOB_CAP") ? (uint64_t)std::max(1, atoi(getenv("MM_K3_JOB_CAP"))) : (1u << 16), stage_path_cap = 48ull << 20, stage_seg_cap = (stage_job_cap + k3.rq_cap) * 8;          /* (MM_K3_JOB_CAP: test hook, a launch with more chain jobs than slots) */
		const bool staged = k3.ring && k3.cls && !safe && (k3.rjobs || (((round == 0 && inkernel) || deferred) && n_heavy > 0 && !getenv("MM_K3_NO_JOBS")))
			&& a->spec_path.ensure(stage_path_cap) && a->spec_seg.ensure(stage_seg_cap) && a->spec_top.ensure(8);
		if(staged) {
			CK(hipMemsetAsync(a->spec_top.p, 0, 64, a->stream));
			k3.spath = a->spec_path.p; k3.spath_cap = stage_path_cap; k3.sseg = a->spec_seg.p; k3.sseg_cap = stage_seg_cap; k3.job_top = a->spec_top.p;
			k3.stage_top = a->spec_top.p + 2;
			/* (from four chains on: the ordinary rescued read has one to three, and publishing those cost the headline set 3 % for nothing; MM_K3_ROUND_JOBS_MIN: another number;
			 * MM_K3_NO_ROUND_JOBS: the chains of the later rounds walked by the read's own wave, as before round 5) */
			k3.round_jobs = (k3.rjobs && !getenv("MM_K3_NO_ROUND_JOBS")) ? 4u : 0u;
		}
		/* chain jobs: the first trials of the chains of the heaviest reads (the front of the work list), taken by all waves of the launch before the reads (K3Args.jobs) */
		/* (a wave that has claimed a job takes the workspace for it without waiting, K3_TRY_SLAB, and hands the job back undone when none of its class is free: with fewer
		 * workspaces than waves, or on the class ladder of a long-tailed set, the waves that hold the workspaces may be the ones that wait for the job) */
		if(staged && ((round == 0 && inkernel) || deferred) && n_heavy > 0 && !getenv("MM_K3_NO_JOBS")) {
			const uint64_t job_cap = stage_job_cap;
			if(a->spec_jobs.ensure(job_cap) && a->spec_memo.ensure(job_cap)) {
				SpecJobsArgs sj; sj.idx = a->dix; sj.in = a->d_in.p; sj.st = a->d_st.p; sj.work = a->d_work.p; sj.n_heavy = n_heavy; sj.seed_pool = a->seed_pool.p; sj.root_pool = a->root_pool.p;
				sj.mcoef = a->mcoef; sj.min_score = a->o.min_score; sj.min_roots = 2; sj.jobs = a->spec_jobs.p; sj.memo = a->spec_memo.p; sj.job_cap = job_cap; sj.job_top = a->spec_top.p;
				hipLaunchKernelGGL(mm_spec_jobs_kernel, dim3((n_heavy + 63) / 64), dim3(64), 0, a->stream, sj);
				CK(hipGetLastError());
				k3.jobs = a->spec_jobs.p; k3.memo = a->spec_memo.p; k3.job_cap = job_cap;
			}
		}
		uint32_t waves = std::min<uint32_t>(a->k3_waves, (uint32_t)((work.size() + 3) & ~3ull));
		if(small_heavy && k3.rjobs) { waves = std::min<uint32_t>(a->k3_waves, std::max<uint32_t>(waves, 1024u)); }          /* (waves for the jobs of a small launch: 256 workgroups, the first wave of each a helper) */
		/* several batches in flight (lanes): 5 persistent waves per SIMD keep the integer VALU as busy as 8 do (a wave issues at most every 4th cycle, about two
		 * thirds of its instructions are VALU) and leave wave slots for the sketch and sort + chain kernels of the other lanes, which otherwise wait for the
		 * tail of this launch: +4 % on round 1's bench workload with 3 in flight, -10 % for a launch running alone; 2 / 3 / 4 / 6 measure the same on the headline workload of
		 * round 2 */
		if(a->is_sib || a->sib) { waves = std::min<uint32_t>(waves, (a->n_waves / MM_K3_WAVES_PER_SIMD) * 5u); }
		hipStream_t xs = a->stream;
		CK(hipEventRecord(a->ev0, xs));
		/* persistent waves stealing reads from a counter, never more of them than there are workspaces.  MM_K3_ONE_READ_PER_WAVE (with a workspace for every wave
		 * the device can hold): grid = reads / 4, a wave maps one read and ends -- wave slots then come free read by read for the other lanes' launches; measured
		 * no faster on the headline workload (2.86 against 2.77 s per step with the rounds in the kernel, 3.7 against 3.2 without), kept as an experiment */
		if(k3.ring) { waves = std::min<uint32_t>(waves, (k3.ring_n * 8u) & ~3u); }          /* (never more waves than the shared ring and the lane's own hold workspaces of the ordinary class) */
		const int k3_conc = safe ? 1 : 0;          /* (a cap on the extension launches in flight measured flat or worse with 4 / 6 / 8 lanes: profiles/round5_sweep_lanes_batches_registers.txt) */
		mm_align_s *GP = GPw;
		/* the watchdog's window into this launch (k3_watchdog_main): where every wave is, and the word that calls the launch off */
		if(!a->wd) { if(hipHostMalloc((void **)&a->wd, (K3_WD_HEAD + (size_t)GP->n_waves) * 4, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { a->wd = nullptr; } }
		if(a->wd) { memset(a->wd, 0, (K3_WD_HEAD + (size_t)GP->n_waves) * 4); k3_watchdog_start(GP); }
		k3.wd = a->wd; k3.wd_n = GP->n_waves; k3.test_hang = 0;
		if(const char *e = getenv("MM_TEST_K3_HANG")) { static std::atomic<int> once{0}; if(round == 0 && !rlen_fixed && k3.n_work > (uint32_t)atoi(e) && once.fetch_add(1) == 0) { k3.test_hang = (uint32_t)atoi(e) + 1u; } }          /* test hook: one wave of the first launch of the process waits for nothing */
		{
			/* the gate of the device's extension launches: closed while the watchdog recovers; in the safe mode (and with MM_K3_CONCURRENT) one launch at a time */
			std::unique_lock<std::mutex> lk(GP->k3_gate_mu);
			GP->k3_gate_cv.wait(lk, [&]() { return !GP->wd_recovering && (k3_conc == 0 || GP->k3_in_flight < k3_conc); });
			if(GP->safe_mode.load() != safe) { a->k3_called_off.store(true); return false; }          /* (the mode changed while this launch was being set up: the batch again, as if it had been called off) */
			GP->k3_in_flight++; a->k3_wd_waves = waves; a->k3_wd_work = k3.n_work; a->k3_t0.store(now_ms());
			if(k3_conc) { (void)hipEventRecord(a->ev0, xs); }
		}
		hipLaunchKernelGGL(mm_extend_kernel, dim3(waves / 4), dim3(256), inkernel ? K3_LDS_BYTES : 0, xs, k3);
		{ const hipError_t le = hipGetLastError(); hipError_t se = le == hipSuccess ? hipEventRecord(a->ev1, xs) : le;
		  /* K4 behind it, on a side stream that waits for the launch: the CIGAR strings of what this launch records (mm_cigar.hpp) are made while the host checks the carried
		   * value of the batch and waits for its turn -- a lane per read walking its paths is a few milliseconds of dependent steps alone and several times that beside the
		   * extension waves of the other lanes: on the lane's own stream it stood in front of the check (3.7 against 4.2 G bases/s); batch_fetch waits for it (k4e).  A run
		   * that prints the strings from the path words on the host (MD tags, the other formats, MM_HOST_CIGAR) does without */
		  if(se == hipSuccess && device_cigar(a) && a->cig_items.p && a->cig_ent.p && a->cig_text.p) {
			se = hipEventRecord(a->k2e[15], xs); if(se == hipSuccess) se = hipStreamWaitEvent(a->k2s[MM_SIDE - 1], a->k2e[15], 0);
			CigArgs ca; ca.st = a->d_st.p; ca.work = a->d_work.p; ca.n_work = k3.n_work; ca.aln_pool = a->aln_pool.p; ca.seg_pool = a->seg_pool.p; ca.path_pool = a->path_pool.p;
			ca.items = a->cig_items.p; ca.item_cap = a->cig_items.n; ca.ent = a->cig_ent.p; ca.ent_cap = a->cig_ent.n; ca.text = a->cig_text.p; ca.text_cap = std::min<uint64_t>(a->cig_text.n, 0xf0000000ull); ca.ctl = tops + 36;
			if(se == hipSuccess) se = hipMemsetAsync(tops + 39, 0, 8, a->k2s[MM_SIDE - 1]);
			if(se == hipSuccess) { hipLaunchKernelGGL(mm_cigar_list_kernel, dim3((k3.n_work + 255) / 256), dim3(256), 0, a->k2s[MM_SIDE - 1], ca); se = hipGetLastError(); }
			if(se == hipSuccess) { hipLaunchKernelGGL(mm_cigar_kernel, dim3(std::min<uint32_t>(2048u, std::max<uint32_t>(1u, (k3.n_work * 4u + 255u) / 256u))), dim3(256), 0, a->k2s[MM_SIDE - 1], ca); se = hipGetLastError(); }
			if(se == hipSuccess) { se = hipEventRecord(a->k4e, a->k2s[MM_SIDE - 1]); a->k4_pending = true; }
		  }
		  if(se == hipSuccess) se = hipEventSynchronize(a->ev1);
		  { std::lock_guard<std::mutex> lk(GP->k3_gate_mu); GP->k3_in_flight--; if(!a->k3_called_off.load()) { GP->wd_longest_ms = std::max(GP->wd_longest_ms, now_ms() - a->k3_t0.load()); } a->k3_t0.store(0.0); }
		  GP->k3_gate_cv.notify_all();
		  CK(se); }
		if(a->k3_called_off.load()) { return false; }          /* called off by the watchdog: the caller runs the batch again (batch_run_spec) */
		CK(hipEventElapsedTime(&ms, a->ev0, a->ev1)); a->st.k3_ms += ms; a->st.k3_launches++;
		/* next round: reads that still have no result (minialign.c:4444-4448) */
		if(!lane_d2h(a, hst.data(), a->d_st.p, (uint64_t)n_reads * sizeof(ReadState))) return false;
		std::vector<uint32_t> nxt;
		(void)nxt;
		break;          /* every round of every read has run inside that launch */
	}
	return true;
}

#include "host_print.hpp"
/* minimizer records a read gets room for, per base: at most one per position; 2 / (w + 1) per base on average, so 4 / (w + 1) is ample for all but the reads inside low-complexity sequence (w = 10: 0.36; a run of one repeated k-mer emits a minimizer per base); a read whose
 * hashes keep falling emits one per position: after an overflow the batch is redone with room for that (scale > 1) */
inline double min_cap_frac(uint32_t w, uint64_t scale) { return scale > 1 ? 1.0 : std::min(1.0, 4.0 / ((double)w + 1.0)); }
/* DP workspace of a wave for reads up to qlen bases (a DOWN and an UP fill chain coexist; each runs at most about 2 x (qlen + 96) + drift vectors), qlen in steps of 8 k */
uint64_t slab_bytes_for(uint32_t qlen) { qlen = (qlen + 8191u) & ~8191u; const uint64_t blocks = 2 * ((2ull * qlen + 8192) / 32 + 64); return (gaba::SLAB_HEAD + blocks * sizeof(gaba::Blk) + 32 * sizeof(gaba::Tail) + 4095) & ~4095ull; }
bool ensure_pools(mm_align_t *a, uint32_t n_reads, uint64_t bases, uint32_t max_qlen, uint64_t scale)
{
	bool ok = true;
	/* sized with an eighth to spare, in steps of 8 192 reads / 32 Mb: the batches of a run differ by a few per cent, and a pool that grows in the middle of a run
	 * costs a hipFree -- which waits for every stream of the device, the other lanes' extension launches included (seen as 100 - 450 ms "upload" times of single
	 * batches, with the lanes behind them waiting for their turn at the carried value) */
	{ const uint64_t rq = n_reads >= 8192 ? 8192 : 256, bq = bases >= (32ull << 20) ? (32ull << 20) : (1ull << 20);          /* (small steps for small batches: tests, the per-read entries) */
	  n_reads = (uint32_t)std::min<uint64_t>(0xffffe000u, ((uint64_t)n_reads + n_reads / 8 + rq - 1) / rq * rq);
	  bases = (bases + bases / 8 + bq - 1) / bq * bq; }
	ok &= a->d_in.ensure(n_reads); ok &= a->d_st.ensure(n_reads); ok &= a->d_work.ensure(n_reads);
	ok &= a->q_pk.ensure(bases / 16 + 8); ok &= a->q_nm.ensure(bases / 32 + 8);
	const uint64_t min_total = (uint64_t)((double)bases * min_cap_frac(a->mi->w, scale)) + 64ull * n_reads + 1024;        /* as the per-read caps of batch_upload */
	/* ... and behind the reads' shares the overflow region of the sketch kernel (K1Args.min_over_*): a read inside low-complexity sequence emits a minimizer per position;
	 * such reads take room for that there instead of making the whole batch run again with larger pools (the hard-repeat human-size set: 35 - 45 of the 7 500 reads of every
	 * batch, every batch twice and the extension-side caps of the lane multiplied each time -- round 5) */
	const uint64_t min_over = bases / 32 + 2ull * max_qlen + 4096;
	ok &= a->min_pool.ensure(min_total + min_over);
	a->min_over_base = min_total; a->min_over_n = min_over;
	{
		/* room for what the batches of this run have asked for (entries per base, + a quarter; note_demand), in steps of 16 M entries; a batch that asks for more is
		 * caught right behind its sketch launch (run_rounds) */
		mm_align_s *NP = a->root ? a->root : a; double ns, nr, nt;
		{ std::lock_guard<std::mutex> lk(NP->need_mu); ns = NP->need_seed; nr = NP->need_resc; nt = NP->need_root; }
		auto room = [&](double per_base, uint64_t per_read) -> uint64_t { const uint64_t w = (uint64_t)((double)bases * per_base * 1.25) * scale + per_read * n_reads + (1ull << 20), q = w < (64ull << 20) ? (1ull << 20) : (16ull << 20); return (w + q - 1) / q * q; };
		ok &= a->seed_pool.ensure(room(ns, 8)); ok &= a->root_pool.ensure(room(nt, 4)); ok &= a->resc_pool.ensure(room(nr, 2));
	}
	ok &= a->k2w_scratch.ensure(a->seed_pool.n * 8);          /* (16 B per seed found, K2wArgs.scratch; a read claims two pool entries per seed it can find) */
	/* (the result pools carry room beyond the reads' ordinary regions: the few reads with hundreds of chains take larger ones as they go, mm_extend_kernel) */
	const uint64_t heavy = bases / 64 + (1ull << 16);
	ok &= a->kh_pool.ensure((uint64_t)n_reads * a->kh_cap + 8 * heavy);
	ok &= a->next_pool.ensure((uint64_t)std::max(a->n_waves, (a->root ? a->root : a)->slab_total) * MM_NEXT_STRIDE(a->next_cap));
	ok &= a->bin_pool.ensure(((uint64_t)n_reads + 2048) * a->bin_cap + 4 * heavy);
	ok &= a->aln_pool.ensure(((uint64_t)n_reads + 2048) * a->aln_cap + heavy);
	ok &= a->seg_pool.ensure((uint64_t)n_reads * a->aln_cap * 2 + 4096 + 8 * heavy);
	ok &= a->path_pool.ensure((bases / 4 + 1024ull * n_reads) * scale + (1ull << 20));
	/* K4's text: two characters per base of batch (PacBio-CLR-like reads make 0.38 per base and record; a read inside a repeat family has a few records) -- a batch that needs
	 * more has its strings made by the host (batch_fetch) */
	if(device_cigar(a)) { ok &= a->cig_items.ensure(a->seg_pool.n) && a->cig_ent.ensure(a->seg_pool.n) && a->cig_text.ensure(std::min<uint64_t>(0xf0000000ull, (2 * bases + (32ull << 20)) & ~((32ull << 20) - 1))); }
	/* DP workspace: a DOWN and an UP fill chain coexist; each runs at most about 2 x (qlen + 96) + drift vectors */
	/* re-allocating tens of GB costs seconds: size for the longest read of the whole input when the caller knows it (qlen_hint), in steps of 8 k bases */
	max_qlen = (std::max(max_qlen, a->qlen_hint) + 8191u) & ~8191u;
	uint64_t blocks = 2 * ((2ull * max_qlen + 8192) / 32 + 64);
	uint64_t slab = (gaba::SLAB_HEAD + blocks * sizeof(gaba::Blk) + 32 * sizeof(gaba::Tail) + 4095) & ~4095ull;
	/* the workspace is per persistent wave and grows with the longest read of the batch (5 MB for 27 kb): with very long reads fewer waves
	 * are launched rather than more than `budget` of HBM taken per lane (MM_SLAB_GB, default 48) */
	const uint64_t budget = (getenv("MM_SLAB_GB") ? (uint64_t)atoll(getenv("MM_SLAB_GB")) : 48ull) << 30;
	const uint32_t lane_waves = a->is_sib ? (a->n_waves / MM_K3_WAVES_PER_SIMD) * 5u : a->n_waves;      /* lanes other than the first never launch more (run_rounds) */
	uint32_t kw = (uint32_t)std::min<uint64_t>(lane_waves, std::max<uint64_t>(256, budget / slab)) & ~3u;
	mm_align_s *P = a->root ? a->root : a;
	if(P->shared_slabs) {
		/* the engine sized the shared workspaces for the longest read of the input before the lanes started */
		/* outside the streaming engine nothing is in flight on the context (the per-call entries after a stream): the workspaces are simply sized again */
		if(P->slab_max < slab && !P->streaming) { ok &= ensure_shared_slabs(P, max_qlen, P->slab_lanes); }
		if(P->slab_max < slab) { fprintf(stderr, "[minialign_amd] a read longer than announced (%u bases) does not fit the shared DP workspaces\n", max_qlen); ok = false; }
		a->k3_waves = lane_waves & ~3u;
	}
	else if(a->slab_stride >= slab && a->k3_waves >= kw) { /* the current allocation already serves */ }
	else { ok &= a->slabs.ensure(slab * kw); if(ok) { a->slab_stride = a->slabs.n / kw; a->k3_waves = kw; } }
	ok &= a->d_tops.ensure(48); ok &= a->d_k2cnt.ensure(48);
	return ok;
}
/* A ring of free workspace numbers as it stands before anything has been taken (k3_ring_try / k3_ring_give): per XCD x the numbers base + x * per .. in their slots,
 * take tickets at 0, give tickets and numbers on offer at `per` */
static bool ring_fill(uint32_t *ring, unsigned long long *ctr, uint32_t per, uint32_t n_xcd, uint32_t base)
{
	if(per == 0) return true;
	std::vector<uint32_t> r((size_t)per * n_xcd); for(size_t i = 0; i < r.size(); i++) r[i] = base + (uint32_t)i;
	std::vector<unsigned long long> c(4 * n_xcd, 0ull); for(uint32_t x = 0; x < n_xcd; x++) { c[4 * x + 1] = per; c[4 * x + 2] = per; }
	return hipMemcpy(ring, r.data(), r.size() * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(ctr, c.data(), c.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
}
/* every ring of the context as new (after the workspaces were made, and after the watchdog has called launches off: a wave that left a wait may have drawn a ticket) */
static bool rings_reset(mm_align_s *P)
{
	const uint32_t n_xcd = 8; bool ok = true;
	for(size_t c = 0; c < P->h_cls.size(); c++) {
		const K3Class &k = P->h_cls[c];
		ok &= ring_fill(k.ring, k.ctr, k.n, n_xcd, 0);
		for(uint32_t l = 0; l < P->slab_lanes; l++) ok &= ring_fill(k.pring + (size_t)l * n_xcd * k.pn, k.pctr + (size_t)l * n_xcd * 4, k.pn, n_xcd, n_xcd * k.n + l * n_xcd * k.pn);
	}
	return ok;
}
/*
 * The DP workspaces of a context, for `lanes` lanes (the streaming engine calls this before its lane threads start, with the longest read of the input).  Per class of
 * read lengths one allocation and, per XCD, a SHARED ring of free numbers that the launches of all lanes take from plus one PRIVATE ring per lane (K3Class).  The
 * ordinary class (reads up to 32 k bases): 3/4 of a workspace per wave the device can hold in the shared ring and one wave per SIMD's worth in every lane's own -- a
 * launch alone reaches its five waves per SIMD, two launches fill the device, as when all of them were shared (rounds 2-5) --, fewer when MM_SLAB_GB (default 64) says
 * so.  What the private rings are for: a wave may only WAIT for a workspace that waves of its own launch hold.
 */
bool ensure_shared_slabs(mm_align_t *P, uint32_t max_qlen, uint32_t lanes)
{
	auto slab_of = [](uint32_t qlen) -> uint64_t { const uint64_t blocks = 2 * ((2ull * qlen + 8192) / 32 + 64); return (gaba::SLAB_HEAD + blocks * sizeof(gaba::Blk) + 32 * sizeof(gaba::Tail) + 4095) & ~4095ull; };
	max_qlen = (std::max(max_qlen, P->qlen_hint) + 8191u) & ~8191u;
	lanes = std::max<uint32_t>(1, std::max(lanes, P->slab_lanes));
	const uint64_t budget = (getenv("MM_SLAB_GB") ? (uint64_t)atoll(getenv("MM_SLAB_GB")) : 64ull) << 30;
	const uint32_t n_xcd = 8;
	/* reads up to 32 k bases (the PacBio-like sets whole, nine tenths of an ONT-like one) take the ordinary class; a longer maximum adds classes of 64 k, 128 k, ... bases
	 * up to it, each with a share of the budget of its own (below) */
	const uint32_t q_small = 32768;
	std::vector<uint32_t> qmax;
	if(max_qlen > q_small && !getenv("MM_ONE_SLAB_CLASS")) {
		for(uint32_t q = q_small; ; q *= 2) { if(q >= max_qlen || qmax.size() + 1 == mm_align_s::MAX_CLS) { qmax.push_back(max_qlen); break; } qmax.push_back(q); }
	} else { qmax.push_back(max_qlen); }
	std::vector<uint64_t> bytes(qmax.size()); std::vector<uint32_t> per(qmax.size()), pn(qmax.size()), sn(qmax.size());
	for(size_t c = 0; c < qmax.size(); c++) {
		bytes[c] = slab_of(qmax[c]);
		const uint32_t full = P->n_waves / n_xcd;          /* a workspace for every wave an XCD can hold */
		if(c == 0) { per[c] = (uint32_t)std::min<uint64_t>(full + full / 4, std::max<uint64_t>(32, budget / bytes[c] / n_xcd)); }
		else {
			/* a quarter of the budget each for the two classes above the ordinary one, an eighth for every class above those.  The waves a class needs go with the share of the
			 * DP work in its reads -- on the ONT-like set 17 % in 32 - 64 kb, 7 % in 64 - 128 kb, 1 % above -- and more at the start of a launch, where the long reads are; with
			 * 8 GB for the 64 - 128 kb class (320 workspaces for the 2 164 such reads of a step on four lanes) its reads spent four fifths of their wave time waiting for one */
			const uint64_t share = c <= 2 ? budget / 4 : budget / 8;
			per[c] = (uint32_t)std::min<uint64_t>(full, std::max<uint64_t>(4, share / bytes[c] / n_xcd));
		}
		/* half of a class in the lanes' own rings (the ordinary class: at most a wave per SIMD per lane), the rest shared */
		pn[c] = std::max<uint32_t>(1, per[c] / (2 * lanes)); if(c == 0) pn[c] = std::min<uint32_t>(pn[c], std::max<uint32_t>(1, full / 8));
		sn[c] = per[c] > lanes * pn[c] ? per[c] - lanes * pn[c] : 0;
	}
	bool same = P->shared_slabs && P->h_cls.size() == qmax.size() && P->slab_lanes >= lanes;
	for(size_t c = 0; same && c < qmax.size(); c++) same = P->h_cls[c].bytes >= bytes[c] && P->h_cls[c].n >= sn[c] && P->h_cls[c].pn >= pn[c] && (c + 1 == qmax.size() || P->h_cls[c].qmax == qmax[c]);
	if(same) return true;
	/* (re)allocation: only between runs -- nothing is in flight when the engine calls */
	if(hipDeviceSynchronize() != hipSuccess) return false;
	P->h_cls.clear(); P->slab_total = 0; P->slab_max = 0; P->slab_lanes = lanes;
	for(size_t c = 0; c < qmax.size(); c++) {
		DBuf<uint8_t> &sl = c ? P->xslabs[c] : P->slabs; DBuf<uint32_t> &rg = c ? P->xring[c] : P->slab_ring; DBuf<unsigned long long> &ct = c ? P->xctr[c] : P->slab_ring_ctr;
		const uint64_t total = (uint64_t)(sn[c] + lanes * pn[c]) * n_xcd;
		if(!sl.ensure(bytes[c] * total) || !rg.ensure(std::max<uint64_t>(1, (uint64_t)sn[c] * n_xcd)) || !ct.ensure(4 * n_xcd) || !P->pring[c].ensure((uint64_t)lanes * pn[c] * n_xcd) || !P->pctr[c].ensure((uint64_t)lanes * 4 * n_xcd)) return false;
		K3Class k; k.slabs = sl.p; k.bytes = bytes[c]; k.ctr = ct.p; k.ring = rg.p; k.n = sn[c]; k.qmax = qmax[c]; k.pn = pn[c]; k.pad = 0; k.pctr = P->pctr[c].p; k.pring = P->pring[c].p;
		P->h_cls.push_back(k); P->slab_total += (uint32_t)total; P->slab_max = bytes[c];
	}
	for(size_t c = qmax.size(); c < mm_align_s::MAX_CLS; c++) { if(c) { P->xslabs[c].release(); P->xring[c].release(); P->xctr[c].release(); } P->pring[c].release(); P->pctr[c].release(); }
	if(!rings_reset(P)) return false;
	/* the table of classes as each lane's launches see it: the lane's own rings */
	std::vector<K3Class> tab;
	for(uint32_t l = 0; l < lanes; l++) for(size_t c = 0; c < P->h_cls.size(); c++) { K3Class k = P->h_cls[c]; k.pring += (size_t)l * n_xcd * k.pn; k.pctr += (size_t)l * n_xcd * 4; tab.push_back(k); }
	if(!P->d_cls.ensure(tab.size()) || hipMemcpy(P->d_cls.p, tab.data(), tab.size() * sizeof(K3Class), hipMemcpyHostToDevice) != hipSuccess) return false;
	P->slab_stride = bytes[0]; P->slab_ring_n = sn[0] + pn[0]; P->k3_waves = P->n_waves;
	if(getenv("MM_VERBOSE")) { for(auto &k : P->h_cls) fprintf(stderr, "[minialign_amd] workspace class: reads up to %u bases, %u shared + %u x %u of the lanes' own, %.1f MB each\n", k.qmax, k.n * n_xcd, lanes, k.pn * n_xcd, k.bytes / 1048576.0); }
	P->shared_slabs = true;
	return true;
}

/*
 * The watchdog of the extension launches.  mm_extend_kernel is the one kernel whose waves wait for each other (workspace rings, published jobs, carried values, the tables
 * of a workgroup); every such wait ticks into a window of pinned host memory (K3Args.wd) and ends when the word at its head is set.  This thread looks at the launches of
 * its device every 50 ms.  One that has been in flight longer than the deadline (MM_K3_WATCHDOG_MS, default: 20 s or 40 x the longest launch that has ended, whichever is
 * more) is not slow, it is stuck: the thread prints where every wave of every launch in flight is, calls all of them off (the waves that wait leave, the others finish
 * the read they hold and take no more), waits for the launches to end, sets the rings up again (a wave that left a wait may have drawn a ticket it never used), and
 * switches the device to the safe mode for the rest of the stream.  The lanes run their batches again from the upload on: same bytes, later.  A launch that does not end
 * within 60 s of being called off is a wave that spins outside every wait -- nothing the host can recover from: the process says so and ends with status 86 instead of
 * spinning with the device for ever.
 */
static const char *k3_wd_site(uint32_t s)
{
	static const char *nm[16] = { "not started / ended", "looks for a DP workspace (none on offer on its XCD)", "gives a DP workspace back (slot busy)", "takes a DP workspace without waiting (number on its way)", "waits for the tables of its workgroup (rescue round)",
		"waits for the carried value of the read in front", "waits for a chain job of before the launch", "waits for a chain job another wave claimed", "waits for a retry job another wave claimed", "without reads, looking for jobs",
		"TEST HOOK: waits for nothing", "?", "?", "runs a job", "at work again after a wait", "at work on a read" };
	return nm[s & 15];
}
static void k3_wd_census(mm_align_s *GP, double now)
{
	int li = 0;
	for(mm_align_s *q = GP; q; q = q->sib, li++) {
		const double t0 = q->k3_t0.load();
		if(t0 == 0.0 || !q->wd) { fprintf(stderr, "[minialign_amd] watchdog: device %d lane %d: no extension launch in flight\n", GP->dev, li); continue; }
		const uint32_t nw = std::min<uint32_t>(q->k3_wd_waves, GP->n_waves);
		uint32_t cnt[16] = { 0 }; std::vector<uint32_t> ex[16];
		for(uint32_t w = 0; w < nw; w++) { const uint32_t v = q->wd[K3_WD_HEAD + w], sidx = v >> 28; cnt[sidx]++; if(ex[sidx].size() < 12) { ex[sidx].push_back(w); ex[sidx].push_back(v & 0x0fffffffu); } }
		fprintf(stderr, "[minialign_amd] watchdog: device %d lane %d: extension launch in flight for %.1f s, %u waves, %u reads\n", GP->dev, li, (now - t0) * 1e-3, nw, q->k3_wd_work);
		for(int sidx = 0; sidx < 16; sidx++) {
			if(!cnt[sidx]) continue;
			fprintf(stderr, "[minialign_amd] watchdog:   %5u wave(s): %s", cnt[sidx], k3_wd_site((uint32_t)sidx));
			if(sidx != 0) { fprintf(stderr, "; e.g."); for(size_t i = 0; i + 1 < ex[sidx].size(); i += 2) { const uint32_t d = ex[sidx][i + 1]; if(sidx == 1 && (d & 0x1000000u)) fprintf(stderr, " wave %u (holds no read; work list at %u)", ex[sidx][i], d & 0xffffffu); else if(sidx == 1) fprintf(stderr, " wave %u (holds a read of class %u)", ex[sidx][i], d); else fprintf(stderr, " wave %u (%u)", ex[sidx][i], d); } }
			fprintf(stderr, "\n");
		}
	}
}
static void k3_wd_rings_dump_and_reset(mm_align_s *GP)
{
	const uint32_t n_xcd = 8;
	for(size_t c = 0; c < GP->h_cls.size(); c++) {
		const K3Class &k = GP->h_cls[c];
		std::vector<unsigned long long> ctr(4 * n_xcd);
		if(hipMemcpy(ctr.data(), k.ctr, ctr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
			fprintf(stderr, "[minialign_amd] watchdog: workspace class %zu (reads up to %u bases), shared ring of %u per XCD: taken / given back / on offer, per XCD:", c, k.qmax, k.n);
			for(uint32_t x = 0; x < n_xcd; x++) fprintf(stderr, " %llu/%llu/%lld", ctr[4 * x], ctr[4 * x + 1] - k.n, (long long)ctr[4 * x + 2]);
			fprintf(stderr, "\n");
		}
		for(uint32_t l = 0; l < GP->slab_lanes; l++) {
			if(hipMemcpy(ctr.data(), k.pctr + (size_t)l * n_xcd * 4, ctr.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) continue;
			fprintf(stderr, "[minialign_amd] watchdog:   lane %u's own ring of %u per XCD:", l, k.pn);
			for(uint32_t x = 0; x < n_xcd; x++) fprintf(stderr, " %llu/%llu/%lld", ctr[4 * x], ctr[4 * x + 1] - k.pn, (long long)ctr[4 * x + 2]);
			fprintf(stderr, "\n");
		}
	}
	if(!rings_reset(GP)) fprintf(stderr, "[minialign_amd] watchdog: the workspace rings could not be set up again\n");
}
static void k3_watchdog_main(mm_align_s *GP)
{
	(void)hipSetDevice(GP->dev);
	const double fixed_ms = getenv("MM_K3_WATCHDOG_MS") ? atof(getenv("MM_K3_WATCHDOG_MS")) : 0.0;
	std::unique_lock<std::mutex> lk(GP->k3_gate_mu);
	while(!GP->wd_stop) {
		GP->k3_gate_cv.wait_for(lk, std::chrono::milliseconds(50));
		if(GP->wd_stop) break;
		if(GP->k3_in_flight == 0) continue;
		const double now = now_ms(), deadline = fixed_ms > 0.0 ? fixed_ms : std::max(20000.0, 40.0 * GP->wd_longest_ms);
		bool late = false;
		for(mm_align_s *q = GP; q; q = q->sib) { const double t0 = q->k3_t0.load(); if(t0 != 0.0 && now - t0 > deadline) late = true; }
		if(!late) continue;
		GP->wd_recovering = true;          /* (no new launch gets through the gate from here on) */
		fprintf(stderr, "[minialign_amd] watchdog: an extension launch on device %d has not ended within %.1f s: it is called off with every other one in flight, and the batches run again in the safe mode\n", GP->dev, deadline * 1e-3);
		k3_wd_census(GP, now);
		for(mm_align_s *q = GP; q; q = q->sib) { if(q->k3_t0.load() != 0.0 && q->wd) { q->k3_called_off.store(true); __atomic_store_n(&q->wd[0], 1u, __ATOMIC_SEQ_CST); } }
		const bool ended = GP->k3_gate_cv.wait_for(lk, std::chrono::seconds(60), [&]() { return GP->k3_in_flight == 0; });
		if(!ended) {
			k3_wd_census(GP, now_ms());
			fprintf(stderr, "[minialign_amd] watchdog: the launch does not end after it was called off (a wave spins outside every wait): giving up\n");
			fflush(stderr); _exit(86);
		}
		k3_wd_rings_dump_and_reset(GP);
		for(mm_align_s *q = GP; q; q = q->sib) { if(q->wd) { memset(q->wd, 0, (K3_WD_HEAD + (size_t)GP->n_waves) * 4); } }
		GP->safe_mode.store(true); GP->st.k3_aborts++;
		GP->wd_recovering = false;
		GP->k3_gate_cv.notify_all();
	}
}
void k3_watchdog_start(mm_align_s *GP)
{
	std::lock_guard<std::mutex> lk(GP->k3_gate_mu);
	if(GP->wd_running) return;
	GP->wd_running = true; GP->wd_stop = false;
	GP->wd_thread = std::thread(k3_watchdog_main, GP);
}
static void k3_watchdog_stop(mm_align_s *GP)
{
	{ std::lock_guard<std::mutex> lk(GP->k3_gate_mu); if(!GP->wd_running) return; GP->wd_stop = true; }
	GP->k3_gate_cv.notify_all();
	GP->wd_thread.join(); GP->wd_running = false;
}

} /* anonymous */

/* the streams of a context: the lane's own and its side streams (every stream wants a hardware queue of its own: MM_SIDE).  Stream priorities and a CU mask for the
 * extension launches were tried in rounds 2 and 5 and measured nothing (HISTORY.md) */
static bool make_streams(mm_align_s *a)
{
	const int greatest = 0;
	if(hipStreamCreateWithPriority(&a->stream, hipStreamNonBlocking, greatest) != hipSuccess || hipEventCreate(&a->ev0) != hipSuccess || hipEventCreate(&a->ev1) != hipSuccess) return false;
	for(int i = 0; i < 16; i++) { if((i < MM_SIDE && hipStreamCreateWithPriority(&a->k2s[i], hipStreamNonBlocking, greatest) != hipSuccess) || hipEventCreateWithFlags(&a->k2e[i], hipEventDisableTiming) != hipSuccess) return false; }
	a->k2s_ok = true;
	if(hipEventCreateWithFlags(&a->k4e, hipEventDisableTiming) != hipSuccess) return false;
	if(hipHostMalloc((void **)&a->pin_note, 64, hipHostMallocPortable) != hipSuccess) return false;
	return true;
}
/* a primary context on the current device */
static mm_align_t *align_init_here(mm_opt_t const *o, mm_idx_t const *mi, int force_replica = 0)
{
	mm_align_t *a = new mm_align_s();
	a->o = *o; a->mi = mi;
	a->gctx = gaba_init(&o->p);
	if(!a->gctx) { fprintf(stderr, "[minialign_amd] mm_align_init: scoring parameters rejected\n"); delete a; return NULL; }
	a->twlen = (uint32_t)(((int32_t)o->wlen << 1) - (int32_t)o->wlen); a->tglen = (uint32_t)(((int32_t)o->glen << 1) - (int32_t)o->glen);   /* _ud(wlen, wlen), minialign.c:4506 */
	/* mcoef / xcoef: both accumulate score_matrix[0] in the reference (minialign.c:4676-4681); kept */
	double mc = 0, xc = 0; for(int i = 0; i < 16; i++) { if((i & 3) == (i >> 3)) mc += o->p.score_matrix[0]; else xc += o->p.score_matrix[0]; }
	a->mcoef = mc / 4.0; a->xcoef = xc / 12.0;
	if(!make_streams(a)) { delete a; return NULL; }
	/* dynamic LDS limits of the sort / chain kernels: per device, set with every context (lane threads only launch) */
	if(hipFuncSetAttribute((const void *)mm_sort_chain_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess || hipFuncSetAttribute((const void *)mm_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { fprintf(stderr, "[minialign_amd] mm_align_init: kernel attributes rejected\n"); delete a; return NULL; }
	/* reference: one arena, per-sequence offsets */
	std::vector<uint64_t> off; std::vector<uint32_t> len;
	bool ok = true;
	if(mi->on_device) {
		/* an index built on a device: the packed reference, the table and the value array are there already -- on the device that built them; any other gets a copy, once */
		int cur = 0; (void)hipGetDevice(&cur);
		if(!idx_replica(mi, cur, force_replica, &a->d_slot, &a->d_val, &a->ref_ar)) { fprintf(stderr, "[minialign_amd] mm_align_init: the index was built on device %d and could not be copied to device %d\n", mi->dev, cur); a->d_slot = nullptr; a->d_val = nullptr; a->ref_ar = nullptr; a->own_index = false; delete a; return NULL; }
		uint64_t total = 0; for(const HSeq &s : mi->seq) { off.push_back(total); len.push_back(s.blen()); total += ((uint64_t)s.blen() + 63) & ~63ull; }
		a->own_index = false;
	} else {
		a->ref_ar = upload_reference(mi, &off, &len);
		if(!a->ref_ar) { delete a; return NULL; }
		ok &= hipMalloc(&a->d_slot, mi->slot.size() * sizeof(IdxSlot)) == hipSuccess;
		ok &= hipMalloc(&a->d_val, mi->val.size() * 8) == hipSuccess;
		if(ok) { (void)hipMemcpy(a->d_slot, mi->slot.data(), mi->slot.size() * sizeof(IdxSlot), hipMemcpyHostToDevice); (void)hipMemcpy(a->d_val, mi->val.data(), mi->val.size() * 8, hipMemcpyHostToDevice); }
	}
	ok &= hipMalloc(&a->d_seq_len, len.size() * 4) == hipSuccess && hipMalloc(&a->d_seq_off, off.size() * 8) == hipSuccess;
	if(!ok) { fprintf(stderr, "[minialign_amd] mm_align_init: index upload failed\n"); delete a; return NULL; }
	(void)hipMemcpy(a->d_seq_len, len.data(), len.size() * 4, hipMemcpyHostToDevice);
	(void)hipMemcpy(a->d_seq_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
	a->dix.slot = a->d_slot; a->dix.mask = mi->mask; a->dix.val = a->d_val; a->dix.seq_len = a->d_seq_len; a->dix.seq_off = a->d_seq_off;
	a->dix.seq_circ = nullptr;
	{
		std::vector<uint8_t> circ; bool any = false;
		for(const HSeq &s : mi->seq) { circ.push_back(s.circular ? 1 : 0); any |= s.circular; }
		if(any) {          /* circular references (-c): the chain stage links across the origin, the extension continues into the reference itself */
			if(hipMalloc(&a->d_seq_circ, circ.size()) != hipSuccess) { fprintf(stderr, "[minialign_amd] mm_align_init: index upload failed\n"); delete a; return NULL; }
			(void)hipMemcpy(a->d_seq_circ, circ.data(), circ.size(), hipMemcpyHostToDevice);
			a->dix.seq_circ = a->d_seq_circ;
		}
	}
	a->dix.n_seq = (uint32_t)mi->seq.size(); a->dix.k = mi->k; a->dix.w = mi->w; a->dix.n_occ = mi->n_occ;
	for(int i = 0; i < 8; i++) a->dix.occ[i] = i < (int)mi->n_occ ? mi->occ[i] : 0;
	hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&prop, dev); a->dev = dev;
	a->n_waves = (uint32_t)prop.multiProcessorCount * 4 * MM_K3_WAVES_PER_SIMD;       /* persistent waves of the extension kernel */
	memset(&a->st, 0, sizeof(a->st)); a->t_wall0 = now_ms();
	return a;
}
/* The devices a context spans.  The reference scales with `-t N` inside one process (source and drain on the main thread, workers between: minialign.c:1013-1048,
 * 4565-4645, 4729); here the workers are the GPUs of the node: mm_align_init takes every visible device (HIP_VISIBLE_DEVICES picks them; MM_DEVICES=n takes the
 * first n from the current one on), one primary context and one replica of the index per device, and the streaming engine (stream_map) deals batches to
 * device x lane.  MM_DEVICE_CONTEXTS=n (test hook for one-GPU boxes): n contexts dealt round robin over the devices taken, several per device. */
static std::vector<int> context_devices()
{
	int ndev = 0, cur = 0; std::vector<int> out;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return out;
	(void)hipGetDevice(&cur);
	int take = ndev;
	if(const char *e = getenv("MM_DEVICES")) take = std::max(1, std::min(ndev, atoi(e)));
	int n_ctx = take;
	if(const char *e = getenv("MM_DEVICE_CONTEXTS")) n_ctx = std::max(1, std::min(16, atoi(e)));
	for(int i = 0; i < n_ctx; i++) out.push_back((cur + i % take) % ndev);
	return out;
}
extern "C" mm_align_t *mm_align_init(mm_opt_t const *o, mm_idx_t const *mi)
{
	const std::vector<int> devs = context_devices();
	if(devs.empty()) { fprintf(stderr, "[minialign_amd] mm_align_init: no HIP device available (the device stages have no CPU path)\n"); return NULL; }
	mm_align_t *a = align_init_here(o, mi);
	if(a) { a->span = devs; }
	return a;
}
/* The contexts on the other devices of the span are made when a streaming entry first runs (stream_map) or when somebody asks how many devices there are
 * (mm_align_devices) -- not by mm_align_init: the per-batch entries (mm_align_seq, mm_align_batch, mm_batch_*) only ever use the first device and should not pay for
 * streams, DP constants and a 20 GB index replica on seven others.  All of them side by side, one thread per device (idx_replica: what serves which copy).
 * false: one of them could not be made (the context is then left with its first device only, and says so). */
static bool ensure_peers(mm_align_t *a)
{
	std::lock_guard<std::mutex> lk(a->span_mu);
	if(a->span.size() <= 1 || !a->peers.empty() || a->span_failed) return !a->span_failed;
	const std::vector<int> &devs = a->span;
	int cur = 0; (void)hipGetDevice(&cur);
	const bool force = getenv("MM_TEST_REPLICA") != NULL;
	const double t0 = now_ms();
	std::vector<mm_align_t *> pe(devs.size(), nullptr); std::vector<std::thread> th;
	for(size_t i = 1; i < devs.size(); i++) th.emplace_back([&, i]() { if(hipSetDevice(devs[i]) == hipSuccess) pe[i] = align_init_here(&a->o, a->mi, force ? (int)i : 0); });
	for(auto &t : th) t.join();
	bool ok = true; for(size_t i = 1; i < devs.size(); i++) ok &= pe[i] != nullptr;
	if(!ok) {
		fprintf(stderr, "[minialign_amd] a context on one of %d devices could not be made: mapping on the first device only\n", (int)devs.size());
		for(size_t i = 1; i < devs.size(); i++) if(pe[i]) { (void)hipSetDevice(pe[i]->dev); mm_align_destroy(pe[i]); }
		a->span_failed = true; (void)hipSetDevice(cur); return false;
	}
	for(size_t i = 1; i < devs.size(); i++) a->peers.push_back(pe[i]);
	(void)hipSetDevice(cur);
	if(getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd] %d device contexts (index replicas included) in %.1f ms\n", (int)devs.size(), now_ms() - t0);
	return true;
}
extern "C" int mm_align_devices(mm_align_t const *a) { (void)ensure_peers(const_cast<mm_align_t *>(a)); return 1 + (int)a->peers.size(); }
static void free_chunk_pool(struct ChunkPool *p);
extern "C" void mm_align_destroy(mm_align_t *a)
{
	if(!a) return;
	if(!a->peers.empty()) { int cur = 0; (void)hipGetDevice(&cur); for(mm_align_s *p : a->peers) { (void)hipSetDevice(p->dev); mm_align_destroy(p); } a->peers.clear(); (void)hipSetDevice(cur); }
	if(a->sib) { mm_align_destroy(a->sib); a->sib = nullptr; }
	for(auto *ps : a->pin_free) delete ps;
	a->pin_free.clear();
	if(!a->is_sib) {
		if(a->own_index) { (void)hipFree(a->d_slot); (void)hipFree(a->d_val); gaba_arena_free(a->ref_ar); }          /* (a device-built index keeps its own, mm_idx_destroy) */
		(void)hipFree(a->d_seq_len); (void)hipFree(a->d_seq_off); if(a->d_seq_circ) (void)hipFree(a->d_seq_circ);
		gaba_clean(a->gctx);
	}
	a->q_pk.release(); a->q_nm.release(); a->d_in.release(); a->d_st.release(); a->d_work.release(); a->min_pool.release(); a->seed_pool.release();
	a->d_text.release(); a->d_codes.release(); a->d_tinfo.release(); a->d_tn.release(); for(uint32_t c = 0; c < mm_align_s::MAX_CLS; c++) { a->xslabs[c].release(); a->xring[c].release(); a->xctr[c].release(); a->pring[c].release(); a->pctr[c].release(); } a->d_cls.release(); a->slab_ring.release(); a->slab_ring_ctr.release(); a->resc_pool.release(); a->root_pool.release(); a->rs_scratch.release(); a->slabs.release(); a->kh_pool.release(); a->next_pool.release();
	a->bin_pool.release(); a->aln_pool.release(); a->seg_pool.release(); a->path_pool.release(); a->d_tops.release(); a->d_k2cnt.release(); a->tap_words.release(); a->k2w_scratch.release(); a->rq_jobs.release(); a->rq_memo.release(); a->rq_state.release(); a->spec_jobs.release(); a->spec_memo.release(); a->spec_path.release(); a->spec_seg.release(); a->spec_top.release(); a->cig_items.release(); a->cig_ent.release(); a->cig_text.release();
	if(a->pin_stage) (void)hipHostFree(a->pin_stage);
	if(a->pin_note) (void)hipHostFree(a->pin_note);
	if(!a->is_sib) { k3_watchdog_stop(a); }
	if(a->wd) { (void)hipHostFree(a->wd); a->wd = nullptr; }
	free_chunk_pool(a->chunk_pool); a->chunk_pool = nullptr;
	(void)hipEventDestroy(a->ev0); (void)hipEventDestroy(a->ev1); (void)hipStreamDestroy(a->stream);
	if(a->k4e) (void)hipEventDestroy(a->k4e);
	if(a->k2s_ok) { for(int i = 0; i < 16; i++) { if(i < MM_SIDE) { (void)hipStreamDestroy(a->k2s[i]); } (void)hipEventDestroy(a->k2e[i]); } }
	delete a;
}
extern "C" void mm_print_sam_header(mm_align_t const *a, FILE *out, char const *arg_line)
{
	if(a->o.format != 0) return;          /* only SAM has a header (minialign.c:5666-5671) */
	fputs("@HD\tVN:1.0\tSO:unsorted\n", out);
	for(const HSeq &s : a->mi->seq) fprintf(out, "@SQ\tSN:%s\tLN:%u\n", s.name.c_str(), s.blen());
	if((a->o.ptags() & 1) && !a->o.rg_line.empty()) fprintf(out, "%s\n", a->o.rg_line.c_str());       /* minialign.c:5111 */
	fprintf(out, "@PG\tID:minialign\tPN:minialign\tVN:%s\tCL:%s\n", "0.6.0-devel", arg_line ? arg_line : "");
}
extern "C" void mm_stats(mm_align_t *a, mm_stats_t *out, int reset)
{
	a->st.wall_ms = now_ms() - a->t_wall0;
	if(out) {
		*out = a->st;
		each_context(a, [&](mm_align_t *ln) {            /* all lanes of all devices: counters add up; kernel times add up too (the lanes overlap in wall time) */
			if(ln == a) return;
			const mm_stats_t &q = ln->st;
			out->k1_ms += q.k1_ms; out->k2_ms += q.k2_ms; out->k3_ms += q.k3_ms; out->k1_launches += q.k1_launches; out->k2_launches += q.k2_launches; out->k3_launches += q.k3_launches;
			out->reads += q.reads; out->bases += q.bases; out->minimizers += q.minimizers; out->seeds += q.seeds; out->fills += q.fills; out->vectors += q.vectors; out->blocks += q.blocks;
			out->traces += q.traces; out->trace_steps += q.trace_steps; out->reruns += q.reruns; out->host_post_ms += q.host_post_ms; out->host_sam_ms += q.host_sam_ms;
			out->k3_cycles_fill += q.k3_cycles_fill; out->k3_cycles_leaf += q.k3_cycles_leaf; out->k3_cycles_trace += q.k3_cycles_trace; out->k3_cycles_total += q.k3_cycles_total;
			out->k3_cycles_next += q.k3_cycles_next; out->k3_cycles_max += q.k3_cycles_max;             /* summed over launches; k3_waves stays the per-launch count */
			out->k2_cycles_sort += q.k2_cycles_sort; out->k2_cycles_chain += q.k2_cycles_chain; out->k2_cycles_total += q.k2_cycles_total; out->k2_reads_hbm += q.k2_reads_hbm;
			out->pool_grows += q.pool_grows; out->pool_regrows += q.pool_regrows; out->batch_splits += q.batch_splits; out->text_bytes += q.text_bytes; out->reader_ms += q.reader_ms; out->k3_aborts += q.k3_aborts; out->d2h_bytes += q.d2h_bytes; out->cigar_bytes_device += q.cigar_bytes_device;
		});
	}
	if(reset) { a->t_wall0 = now_ms(); each_context(a, [](mm_align_t *ln) { memset(&ln->st, 0, sizeof(ln->st)); }); }
}

/* ---------------------------------------------------------------------------------------------
 * a batch in three phases: upload (H2D of packed reads), run (the hot path: K1..K3 in rounds plus the re-runs the
 * carried reference-length state asks for; results stay in HBM), finish (D2H, post-map, SAM text)
 * --------------------------------------------------------------------------------------------- */
struct mm_reads_s { std::vector<HSeq> r; uint64_t bases = 0; std::vector<std::shared_ptr<std::vector<char>>> text; };          /* text: the files' text when kept (reads then carry where their bases stand in it: the device packs from there) */
struct Batch {
	uint32_t n = 0; uint64_t total = 0; uint32_t max_qlen = 0;
	/* reads that stand in a text the device reader scanned: no parsed record, no base codes on the host -- names, bases and qualities are read off the text when a
	 * record is printed (materialize).  dch: the stretches in HBM that hold the reads (first read, count), until the batch has been packed */
	std::shared_ptr<TextSrc> tsrc; std::vector<RRec> trec;
	struct Piece { std::shared_ptr<DevChunk> ch; uint32_t first, n; };
	std::vector<Piece> dch;
	std::vector<uint32_t> lens; std::vector<uint64_t> qoff; std::vector<const uint8_t *> seq; std::vector<std::string> names;
	std::vector<const HSeq *> rec;         /* the parsed records (qualities, comments) when the batch comes from a file; empty for in-memory batches */
	mm_reg_t **regs = nullptr;             /* when set: one mm_reg_t per read (NULL = unmapped) instead of text (mm_align_batch_regs) */
	std::vector<uint32_t> pk, nm; std::vector<ReadIn> in; std::vector<ReadState> hst; std::vector<uint32_t> work;
	uint64_t scale = 1; bool packed = false, uploaded = false, ran = false;
	std::vector<uint32_t> used;            /* the carried reference length each read actually ran with */
	std::shared_ptr<std::vector<char>> text;   /* when every read of the batch stands in this one text: 2-bit packing on the device from it (batch_upload) */
	const std::vector<std::shared_ptr<std::vector<char>>> *text_src = nullptr;      /* the texts of the read set the batch was cut from */
};
namespace {
bool batch_upload(mm_align_t *a, Batch &b)
{
	const bool verbose = getenv("MM_VERBOSE") != NULL; double tv = now_ms();
	if(a->k4_pending) { CK(hipStreamSynchronize(a->k2s[MM_SIDE - 1])); a->k4_pending = false; }          /* (a batch that goes up again -- pools grown, a launch called off -- while K4 still walks what the last attempt recorded: the pools may move) */
	if(!ensure_pools(a, b.n, b.total + 64, b.max_qlen, b.scale)) return false;
	if(verbose) { fprintf(stderr, "[minialign_amd]   pools %.1f ms\n", now_ms() - tv); tv = now_ms(); }
	b.hst.assign(b.n, ReadState()); b.work.clear();
	uint64_t moff = 0; const double mcf = min_cap_frac(a->mi->w, b.scale);
	a->batch_bases = b.total;
	for(uint32_t i = 0; i < b.n; i++) {
		memset(&b.hst[i], 0, sizeof(ReadState));
		/* minimizers of a read: at most one per position; 2 / (w + 1) per base on average, so half the length is ample from w = 4 up */
		/* (a read whose hashes keep falling emits one per position: after a pool overflow the batch is redone with room for that) */
		b.hst[i].min_off = moff; b.hst[i].min_cap = (uint32_t)((double)b.lens[i] * mcf) + 64; moff += b.hst[i].min_cap;
		b.hst[i].bin_off = ~0ull; b.hst[i].apos0 = gaba::NIL; b.hst[i].rid_last = gaba::NIL; b.hst[i].pred_rid = gaba::NIL; b.hst[i].dep = gaba::NIL;
		/* unmappable reads are skipped outright (minialign.c:4434) */
		if(!(b.lens[i] < a->mi->k || b.lens[i] * a->mcoef < (double)a->o.min_score)) b.work.push_back(i);
	}
	if(b.tsrc) {
		/* K0 over reads the device reader found: their text is in HBM already (the stretches the reader scanned), or -- a batch that is uploaded again after its
		 * pools overflowed -- goes up once more from the host's copy */
		auto spare = [](uint64_t x) -> uint64_t { return (x + x / 8 + (1ull << 20)) & ~((1ull << 20) - 1); };
		const uint64_t arena = (b.total + 64 + 63) & ~63ull;
		if(!a->d_tinfo.ensure(spare(b.n)) || !a->d_codes.ensure(spare(arena + 64)) || !a->d_tn.ensure(spare(b.n))) return false;
		std::vector<TextRead> tr(b.n);
		CK(hipMemsetAsync(a->d_codes.p, 0, arena + 64, a->stream));
		if(!b.dch.empty()) {
			/* The whole upload of a batch behind ONE wait.  Every wait of a lane for its stream is a wait for a wave slot beside the persistent extension waves of the other
			 * lanes (10 - 40 ms each under load), and this function had five: the extents, the packed-base count, the read table, the states, the cursors.  Everything the
			 * device needs goes through one pinned staging buffer in three copies, the kernels and the memsets follow on the stream, the base counts come back into the same
			 * buffer, and the host waits once: +3 % in three pairs of headline runs of round 5 (4.22 / 4.42 / 4.29 against 4.06 / 4.24 / 4.21 G bases/s), whole suite green.  It
			 * was taken back then because the hard-repeat record did not finish with it -- which was the oversubscribed runlist of DESIGN.md 4b, not this */
			for(const Batch::Piece &pc : b.dch) for(uint32_t i = pc.first; i < pc.first + pc.n; i++) tr[i] = TextRead{ b.trec[i].t_off - pc.ch->off, b.trec[i].t_len, 0, b.qoff[i] };
			const size_t s_tr = ((size_t)b.n * sizeof(TextRead) + 255) & ~(size_t)255, s_in = ((size_t)b.n * sizeof(ReadIn) + 255) & ~(size_t)255, s_st = ((size_t)b.n * sizeof(ReadState) + 255) & ~(size_t)255, s_tn = ((size_t)b.n * 4 + 255) & ~(size_t)255;
			uint8_t *stg = (uint8_t *)lane_stage(a, s_tr + s_in + s_st + s_tn);
			if(stg) {
				memcpy(stg, tr.data(), (size_t)b.n * sizeof(TextRead)); memcpy(stg + s_tr, b.in.data(), (size_t)b.n * sizeof(ReadIn)); memcpy(stg + s_tr + s_in, b.hst.data(), (size_t)b.n * sizeof(ReadState));
				CK(hipMemcpyAsync(a->d_tinfo.p, stg, (size_t)b.n * sizeof(TextRead), hipMemcpyHostToDevice, a->stream));
				CK(hipMemcpyAsync(a->d_in.p, stg + s_tr, (size_t)b.n * sizeof(ReadIn), hipMemcpyHostToDevice, a->stream));
				CK(hipMemcpyAsync(a->d_st.p, stg + s_tr + s_in, (size_t)b.n * sizeof(ReadState), hipMemcpyHostToDevice, a->stream));
				for(const Batch::Piece &pc : b.dch) {
					if(pc.n == 0) continue;
					hipLaunchKernelGGL(mm_text_codes_kernel, dim3((pc.n + 3) / 4), dim3(256), 0, a->stream, pc.ch->d, a->d_tinfo.p + pc.first, pc.n, a->d_codes.p, a->d_tn.p + pc.first);
					CK(hipGetLastError());
				}
				const uint64_t nw1 = (b.total + 64 + 31) / 32;
				hipLaunchKernelGGL(mm_codes_pack_kernel, dim3((uint32_t)((nw1 + 255) / 256)), dim3(256), 0, a->stream, a->d_codes.p, nw1, a->q_pk.p, a->q_nm.p);
				CK(hipGetLastError());
				CK(hipMemsetAsync(a->d_tops.p, 0, 40 * 8, a->stream));
				CK(hipMemcpyAsync(stg + s_tr + s_in + s_st, a->d_tn.p, (size_t)b.n * 4, hipMemcpyDeviceToHost, a->stream));
				CK(hipStreamSynchronize(a->stream));
				const uint32_t *tn1 = (const uint32_t *)(stg + s_tr + s_in + s_st);
				for(uint32_t i = 0; i < b.n; i++) if(tn1[i] != b.lens[i]) { fprintf(stderr, "[minialign_amd] read %u of a batch: %u bases in its text when packed, %u when scanned\n", i, tn1[i], b.lens[i]); return false; }
				b.dch.clear();
				if(verbose) { fprintf(stderr, "[minialign_amd]   host state + H2D %.1f ms (one wait)\n", now_ms() - tv); }
				b.uploaded = true; b.ran = false;
				return true;
			}
		}
		if(!b.dch.empty()) {
			for(const Batch::Piece &pc : b.dch) for(uint32_t i = pc.first; i < pc.first + pc.n; i++) tr[i] = TextRead{ b.trec[i].t_off - pc.ch->off, b.trec[i].t_len, 0, b.qoff[i] };
			if(!lane_h2d(a, a->d_tinfo.p, tr.data(), b.n * sizeof(TextRead))) return false;
			for(const Batch::Piece &pc : b.dch) {
				if(pc.n == 0) continue;
				hipLaunchKernelGGL(mm_text_codes_kernel, dim3((pc.n + 3) / 4), dim3(256), 0, a->stream, pc.ch->d, a->d_tinfo.p + pc.first, pc.n, a->d_codes.p, a->d_tn.p + pc.first);
				CK(hipGetLastError());
			}
		} else {
			uint64_t lo = ~0ull, hi = 0; for(uint32_t i = 0; i < b.n; i++) { lo = std::min<uint64_t>(lo, b.trec[i].t_off); hi = std::max<uint64_t>(hi, b.trec[i].t_off + b.trec[i].t_len); }
			if(lo > hi) { lo = hi = 0; }
			for(uint32_t i = 0; i < b.n; i++) tr[i] = TextRead{ b.trec[i].t_off - lo, b.trec[i].t_len, 0, b.qoff[i] };
			if(!a->d_text.ensure(spare(hi - lo + 64))) return false;
			if(hi > lo && !lane_h2d(a, a->d_text.p, b.tsrc->p + lo, hi - lo)) return false;
			if(!lane_h2d(a, a->d_tinfo.p, tr.data(), b.n * sizeof(TextRead))) return false;
			hipLaunchKernelGGL(mm_text_codes_kernel, dim3((b.n + 3) / 4), dim3(256), 0, a->stream, a->d_text.p, a->d_tinfo.p, b.n, a->d_codes.p, a->d_tn.p);
			CK(hipGetLastError());
		}
		const uint64_t nw = (b.total + 64 + 31) / 32;
		hipLaunchKernelGGL(mm_codes_pack_kernel, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, a->stream, a->d_codes.p, nw, a->q_pk.p, a->q_nm.p);
		CK(hipGetLastError());
		/* the scan and the packing must agree on how many bases a read has */
		std::vector<uint32_t> tn(b.n); CPY(a, tn.data(), a->d_tn.p, b.n * 4, hipMemcpyDeviceToHost);
		for(uint32_t i = 0; i < b.n; i++) if(tn[i] != b.lens[i]) { fprintf(stderr, "[minialign_amd] read %u of a batch: %u bases in its text when packed, %u when scanned\n", i, tn[i], b.lens[i]); return false; }
		b.dch.clear();          /* the stretches go back to the reader's pool */
	}
	else if(b.text) {
		/* K0: the stretch of the file's text that holds the batch goes up as it is; newlines are squeezed out, bases coded and packed on the device */
		uint64_t lo = ~0ull, hi = 0; for(uint32_t i = 0; i < b.n; i++) { lo = std::min<uint64_t>(lo, b.rec[i]->t_off); hi = std::max<uint64_t>(hi, b.rec[i]->t_off + b.rec[i]->t_len); }
		if(lo > hi) { lo = hi = 0; }
		std::vector<TextRead> tr(b.n); for(uint32_t i = 0; i < b.n; i++) tr[i] = TextRead{ b.rec[i]->t_off - lo, b.rec[i]->t_len, 0, b.qoff[i] };
		const uint64_t arena = (b.total + 64 + 63) & ~63ull;
		/* (with an eighth to spare, like the pools: a buffer that grows in mid-run costs a hipFree, which waits for every stream of the device) */
		auto spare = [](uint64_t x) -> uint64_t { return (x + x / 8 + (1ull << 20)) & ~((1ull << 20) - 1); };
		if(!a->d_text.ensure(spare(hi - lo + 64)) || !a->d_tinfo.ensure(spare(b.n)) || !a->d_codes.ensure(spare(arena + 64)) || !a->d_tn.ensure(spare(b.n))) return false;
		if(hi > lo && !lane_h2d(a, a->d_text.p, b.text->data() + lo, hi - lo)) return false;
		if(!lane_h2d(a, a->d_tinfo.p, tr.data(), b.n * sizeof(TextRead))) return false;
		CK(hipMemsetAsync(a->d_codes.p, 0, arena + 64, a->stream));
		hipLaunchKernelGGL(mm_text_codes_kernel, dim3((b.n + 3) / 4), dim3(256), 0, a->stream, a->d_text.p, a->d_tinfo.p, b.n, a->d_codes.p, a->d_tn.p);
		CK(hipGetLastError());
		const uint64_t nw = (b.total + 64 + 31) / 32;
		hipLaunchKernelGGL(mm_codes_pack_kernel, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, a->stream, a->d_codes.p, nw, a->q_pk.p, a->q_nm.p);
		CK(hipGetLastError());
		/* the parser and the device must agree on how many bases a read has */
		std::vector<uint32_t> tn(b.n); CPY(a, tn.data(), a->d_tn.p, b.n * 4, hipMemcpyDeviceToHost);
		for(uint32_t i = 0; i < b.n; i++) if(tn[i] != b.lens[i]) { fprintf(stderr, "[minialign_amd] read `%s': %u bases in its text on the device, %u from the parser\n", b.names[i].c_str(), tn[i], b.lens[i]); return false; }
	}
	else if(!lane_h2d(a, a->q_pk.p, b.pk.data(), b.pk.size() * 4) || !lane_h2d(a, a->q_nm.p, b.nm.data(), b.nm.size() * 4)) return false;
	if(!lane_h2d(a, a->d_in.p, b.in.data(), b.n * sizeof(ReadIn))) return false;
	if(!lane_h2d(a, a->d_st.p, b.hst.data(), b.n * sizeof(ReadState))) return false;
	CK(hipMemsetAsync(a->d_tops.p, 0, 40 * 8, a->stream)); CK(hipStreamSynchronize(a->stream));
	if(verbose) { fprintf(stderr, "[minialign_amd]   host state + H2D %.1f ms\n", now_ms() - tv); }
	b.uploaded = true; b.ran = false;
	return true;
}
/* host side of a batch: arena offsets and the 2-bit / N-mask words of its reads (no device work) */
void batch_pack(Batch &b, bool on_host = true)
{
	b.n = (uint32_t)b.lens.size(); b.total = 0; b.max_qlen = 0; b.qoff.resize(b.n); b.in.resize(b.n);
	for(uint32_t i = 0; i < b.n; i++) { b.qoff[i] = b.total; b.total += ((uint64_t)b.lens[i] + 63) & ~63ull; b.max_qlen = std::max(b.max_qlen, b.lens[i]); b.in[i] = ReadIn{ b.qoff[i], b.lens[i], 0 }; }
	/* the reads of a batch that come from one file whose text was kept are packed on the device, from that text (batch_upload) */
	b.text.reset();
	if(b.tsrc) { b.scale = 1; b.packed = true; return; }          /* reads in a scanned text: packed on the device (batch_upload) */
	if(!on_host && !b.rec.empty() && b.rec.size() == b.n && b.text_src) {
		bool all = true; for(uint32_t i = 0; i < b.n && all; i++) all = b.rec[i]->t_id == b.rec[0]->t_id && b.rec[i]->t_id >= 0 && (size_t)b.rec[i]->t_id < b.text_src->size();
		if(all && b.n) b.text = (*b.text_src)[b.rec[0]->t_id];
	}
	if(!b.text) {
	b.pk.assign((b.total + 64) / 16 + 8, 0); b.nm.assign((b.total + 64) / 32 + 8, 0);
	host_parallel(b.n / 64 + 1, [&](uint32_t t, uint32_t nth) {          /* reads occupy disjoint, word-aligned stretches of the arena */
		for(uint32_t i = (uint32_t)((uint64_t)b.n * t / nth); i < (uint32_t)((uint64_t)b.n * (t + 1) / nth); i++) { pack_bases(b.seq[i], b.lens[i], b.pk, b.nm, b.qoff[i]); }
	});
	}
	b.scale = 1; b.packed = true;
	if(getenv("MM_VERBOSE")) { fprintf(stderr, "[minialign_amd]   pack done\n"); }
}
bool batch_prepare(mm_align_t *a, Batch &b)
{
	if(!b.packed) batch_pack(b);
	return batch_upload(a, b);
}
/* the hot path over the uploaded batch; returns 0 ok, 1 device pools overflowed (caller grows and retries), 2 an extension launch was called off by the watchdog (caller
 * uploads the batch again and runs it again: batch_again), -1 error */
/* the carried reference length: check what each read ran with (b.used) against the chain of values the reads actually
 * produce, starting from a->rlen_carry, and re-run the reads it changes until nothing moves.  0 ok, 1 overflow, -1 error */
int batch_verify_carry(mm_align_t *a, Batch &b)
{
	const uint32_t n_reads = b.n;
	std::vector<ReadState> &hst = b.hst; std::vector<uint32_t> &used = b.used;
	bool overflow = false;
	for(int iter = 0; iter < 64; iter++) {
		std::vector<uint32_t> redo, redo_rlen;
		uint32_t cur = a->rlen_carry;
		for(uint32_t i = 0; i < n_reads; i++) {
			if(hst[i].err) { if(!overflow && getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd]   read %u (%u bases) reports err 0x%x: seed_n %u n_seed %u seed_cap %u n_root %u\n", i, b.lens[i], hst[i].err, hst[i].seed_n, hst[i].n_seed, hst[i].seed_cap, hst[i].n_root); overflow = true; }
			uint32_t truth = cur;
			if(truth != used[i] && hst[i].apos0 != gaba::NIL && !hst[i].cond0 && ((hst[i].apos0 >= used[i]) != (hst[i].apos0 >= truth))) { redo.push_back(i); redo_rlen.push_back(truth); }
			cur = hst[i].rid_last != gaba::NIL ? a->mi->seq[hst[i].rid_last].blen() : truth;
		}
		if(redo.empty() || overflow) break;
		a->st.reruns += redo.size();
		a->rerun_heavy = false;
		for(size_t j = 0; j < redo.size(); j++) { if(hst[redo[j]].k3_chains >= 8u || hst[redo[j]].presc != 0u) a->rerun_heavy = true; }          /* (a read that walked many chains, or went on to the later thresholds: its re-run gets helpers, run_rounds) */
		for(size_t j = 0; j < redo.size(); j++) {
			uint32_t i = redo[j]; uint64_t mo = hst[i].min_off; uint32_t mc = hst[i].min_cap;
			memset(&hst[i], 0, sizeof(ReadState)); hst[i].min_off = mo; hst[i].min_cap = mc;
			hst[i].bin_off = ~0ull; hst[i].apos0 = gaba::NIL; hst[i].rid_last = gaba::NIL; hst[i].pred_rid = gaba::NIL; hst[i].dep = gaba::NIL;
			hst[i].rlen = redo_rlen[j];          /* (run_rounds' short way for a few reads takes it from here) */
			used[i] = redo_rlen[j];
		}
		if(!lane_h2d(a, a->d_st.p, hst.data(), n_reads * sizeof(ReadState))) return -1;
		if(!run_rounds(a, n_reads, redo, true, hst, &redo_rlen, b.lens)) return a->k3_called_off.load() ? 2 : -1;
	}
	return overflow ? 1 : 0;
}
/* K1..K3 over the batch with the carried reference length predicted from a->rlen_carry on (b.used = what each read ran with); the device pools are
 * checked right away.  0 ok, 1 a pool or a per-read cap overflowed, -1 error */
int batch_run_spec(mm_align_t *a, Batch &b)
{
	const uint32_t n_reads = b.n;
	std::vector<ReadState> &hst = b.hst;
	a->ran_with.clear();
	if(!run_rounds(a, n_reads, b.work, true, hst, nullptr, b.lens)) return a->k3_called_off.load() ? 2 : -1;
	/* what every read ran with: the values handed out BEFORE the extension launch (run_rounds).  They must not be derived again from the reads' states afterwards: a read
	 * that goes on to the next occurrence threshold is chained again inside the launch (k3_rescue_round) and then carries the prediction of THAT chaining in pred_rid,
	 * while the reads behind it ran with the prediction of its first one -- derived afterwards, `used` then said what the reads should have run with, the check against
	 * the true chain of values found nothing to re-run, and a read whose `apos >= rlen` decision the difference flips kept the records of the wrong decision (one read
	 * of the 266 589 of the ONT-like hg38-size set, found by the whole-set comparison of round 4; rounds 1-3 compared the first 20 000) */
	if(getenv("MM_VERBOSE") && a->np0.size() == n_reads) {
		/* which reads with a chain worth a trial at the first threshold ended that round without a result (and went on to the next: what they leave is then not what was predicted):
		 * by the summed length of their passing chains in units of the length a chain needs to be tried (2 x min_score / mcoef) */
		const double unit = 2.0 * a->o.min_score / a->mcoef; uint32_t h_all[8] = { 0 }, h_fail[8] = { 0 }, n_all = 0, n_fail = 0;
		for(uint32_t i = 0; i < n_reads; i++) { if(a->np0[i] == 0) continue; const int q = (int)std::min(7.0, a->wp0[i] / unit / 1.0 - 1.0 < 0 ? 0.0 : std::floor(std::log2(std::max(1.0, a->wp0[i] / unit)) )); const bool fail = hst[i].presc != 0 || (hst[i].n_res == 0 && hst[i].n_resc == 0); h_all[q]++; n_all++; if(hst[i].presc != 0) { h_fail[q]++; n_fail++; } (void)fail; }
		fprintf(stderr, "[minialign_amd]   reads with a passing chain: %u, of which on to the later thresholds: %u; by log2 of (summed passing length / trial length): all", n_all, n_fail);
		for(int q = 0; q < 8; q++) fprintf(stderr, " %u", h_all[q]); fprintf(stderr, "; on to the later thresholds"); for(int q = 0; q < 8; q++) fprintf(stderr, " %u", h_fail[q]); fprintf(stderr, "\n");
	}
	if(a->ran_with.size() == n_reads) { b.used = a->ran_with; for(uint32_t i = 0; i < n_reads; i++) { if(hst[i].dep != gaba::NIL) b.used[i] = hst[i].rlen_in; } }          /* (a read behind a source ran with what the source left: the kernel says what that was) */
	else { b.used.assign(n_reads, 0); uint32_t cur = a->rlen_carry; for(uint32_t i = 0; i < n_reads; i++) { b.used[i] = cur; if(hst[i].pred_rid != gaba::NIL) cur = a->mi->seq[hst[i].pred_rid].blen(); } }
	{
		/* which caps gave way, and on how many reads: said once per attempt (the batch is then run again with larger pools) */
		uint32_t bits = 0, n_bad = 0, first = 0; for(uint32_t i = 0; i < n_reads; i++) if(hst[i].err) { if(!n_bad) first = i; bits |= hst[i].err; n_bad++; }
		if(n_bad) {
			static const char *nm[9] = { "seed / minimizer room", "DP workspace", "path pool", "segment pool", "position hash", "result bins", "alignments per read", "next-seed list", "sort stack" };
			std::string w; for(int k = 0; k < 9; k++) if(bits & (1u << k)) { if(!w.empty()) w += ", "; w += nm[k]; }
			fprintf(stderr, "[minialign_amd] %u of %u reads of a batch ran out of room (%s); the first: read %u, %u bases, %u seeds, %u chains of which %u pass the length test, err 0x%x; it holds %u of %u bin words, %u of %u alignments, a position hash of %u of %u slots; pools: bins %.1f M words, alignments %.2f M\n", n_bad, n_reads, w.c_str(), first, b.lens[first], hst[first].seed_n0, hst[first].n_root, hst[first].n_pass, hst[first].err,
				hst[first].n_bin, hst[first].bin_cap, hst[first].n_aln, hst[first].aln_cap, hst[first].kh_cnt, hst[first].kh_cap, a->bin_pool.n * 1e-6, a->aln_pool.n * 1e-6);
			return 1;
		}
	}
	return 0;
}
int batch_run_once(mm_align_t *a, Batch &b)
{
	int rc = batch_run_spec(a, b);
	if(rc < 0) return rc;
	rc = batch_verify_carry(a, b);          /* (also reports which read overflowed, with MM_VERBOSE) */
	if(rc != 0) return rc;
	b.ran = true;
	return 0;
}
/* after an overflow: larger pools and per-read caps for the next attempt; false when there is no point in growing further */
bool batch_grow(mm_align_t *a, Batch &b)
{
	if(b.scale >= 256) { fprintf(stderr, "[minialign_amd] batch does not fit the device pools\n"); return false; }
	a->st.pool_grows++;
	b.scale *= 4; a->bin_cap *= 2; a->aln_cap *= 2; a->kh_cap *= 4; a->next_cap *= 2; a->rs_stride = 512 + (a->rs_stride - 512) * 4;
	fprintf(stderr, "[minialign_amd] device pools overflowed, retrying the batch with scale %lu\n", (unsigned long)b.scale);
	return true;
}
uint32_t batch_carry_out(const mm_align_t *a, const Batch &b, uint32_t cur)
{
	for(uint32_t i = 0; i < b.n; i++) { if(b.hst[i].rid_last != gaba::NIL) cur = a->mi->seq[b.hst[i].rid_last].blen(); }
	return cur;
}
bool batch_run(mm_align_t *a, Batch &b)
{
	while(true) {
		int rc = batch_run_once(a, b);
		if(rc < 0) return false;
		if(rc == 0) return true;
		if(rc == 2) { if(!batch_upload(a, b)) return false; continue; }          /* called off by the watchdog: again from the upload on */
		/* a pool or a per-read cap overflowed: grow and redo the batch */
		if(!batch_grow(a, b) || !batch_upload(a, b)) return false;
	}
}
/* mm_pack_reg (minialign.c:4364-4398) into one malloc block per read: the mm_reg_t header with its pointer array, then for every alignment an mm_aln_t
 * { aid, mapq } directly followed by its gaba_alignment_t (header, path words with the two header words in plen / padding, segments) */
mm_reg_t *build_reg(const OutReg &reg, const AlnRec *alns, const gaba::Segment *segs, const uint32_t *paths)
{
	if(!reg.mapped || reg.n_all == 0) return nullptr;
	auto aln_bytes = [&](const AlnRec &al) { return (size_t)((sizeof(mm_aln_t) + sizeof(gaba_alignment_t) + (((uint64_t)al.plen + 31) / 32 + 2 + 7) / 8 * 8 * 4 + al.slen * sizeof(gaba_path_section_t) + 15) & ~15ull); };
	size_t head = (sizeof(mm_reg_t) + reg.n_all * sizeof(mm_aln_t *) + 15) & ~15ull, total = head;
	for(uint32_t i = 0; i < reg.n_all; i++) total += aln_bytes(alns[reg.aln[i].aln]);
	uint8_t *blk = (uint8_t *)calloc(1, total);
	if(!blk) return nullptr;
	mm_reg_t *r = (mm_reg_t *)blk; r->n_all = reg.n_all; r->n_uniq = reg.n_uniq;
	mm_aln_t const **tab = (mm_aln_t const **)(blk + sizeof(mm_reg_t));
	uint8_t *p = blk + head;
	for(uint32_t i = 0; i < reg.n_all; i++) {
		const AlnRec &al = alns[reg.aln[i].aln];
		mm_aln_t *m = (mm_aln_t *)p; m->aid = i; m->mapq = reg.aln[i].mapq;
		gaba_alignment_t *g = (gaba_alignment_t *)(p + sizeof(mm_aln_t));
		g->score = al.score; g->identity = al.identity; g->agcnt = al.agcnt; g->bgcnt = al.bgcnt; g->dcnt = al.dcnt; g->slen = al.slen; g->plen = al.plen; g->padding = 0x40000000u;
		const uint64_t pw = (((uint64_t)al.plen + 31) / 32 + 2 + 7) / 8 * 8;
		memcpy(g->path, paths + al.path_off, (((uint64_t)al.plen + 31) / 32) * 4);
		gaba_path_section_t *sg = (gaba_path_section_t *)((uint8_t *)g->path + pw * 4);
		memcpy(sg, segs + al.seg_off, al.slen * sizeof(gaba_path_section_t)); g->seg = sg;
		tab[i] = m; p += aln_bytes(al);
	}
	return r;
}
/* finish, first half (needs the lane's device pools): counters, then the result pools of the batch copied to the host */
struct Fetched {
	/* the lane's 32 counters (d_tops), who writes which: [0..7] pool cursors (seed, rescue, root, bin, alignment, segment, path, position-hash regions), [8..15] the sketch and extension
	 * kernels' statistics, [16] their work counters, [17..23] the extension kernel's profile ([17] longest wave, [19] next, [20] fill, [21] leaf, [22] trace, [23] total), [24..27] + [29] the LDS sort + chain kernel and the chain / scan kernels through
	 * `prof` = tops + 24 (their prof[0] sort, [1] chain, [2] total, [3] reads beyond the LDS, [5] = tops[29] reads that did not fit: index 4 is used by none of them), [28] the
	 * HBM sort kernel alone (`prof` = tops + 28, its prof[0] only), [30] the chain sweep's scratch cursor.  The ranges are disjoint as long as nobody starts to use prof[4] */
	unsigned long long tops[40];          /* ([36 .. 38]: K4's cursors -- segments done, text bytes, overflow) */
	Root *root = nullptr; uint64_t *bin = nullptr; AlnRec *aln = nullptr; gaba::Segment *seg = nullptr; uint32_t *path = nullptr;
	const CigEnt *cig_ent = nullptr; const char *cig_text = nullptr;          /* the CIGAR strings of the batch as the device made them (K4), indexed by segment slot; NULL: the printers walk the path words (which are then what was fetched) */
	uint64_t d2h_bytes = 0;
	std::unique_ptr<uint8_t[]> own[7];          /* plain host memory when no pinned set is given */
	mm_align_s::PinSet *pin = nullptr;          /* pinned set the pointers live in (returned to the pool by the caller) */
};
bool batch_fetch(mm_align_t *a, Batch &b, Fetched &f)
{
	const uint32_t n_reads = b.n; std::vector<ReadState> &hst = b.hst;
	a->rlen_carry = batch_carry_out(a, b, a->rlen_carry);
	unsigned long long *tops = f.tops; CPY(a, tops, a->d_tops.p, sizeof(f.tops), hipMemcpyDeviceToHost);
	a->st.minimizers += tops[8]; a->st.seeds += tops[9]; a->st.fills += tops[10]; a->st.vectors += tops[11]; a->st.blocks += tops[12]; a->st.traces += tops[13]; a->st.trace_steps += tops[14];
	a->st.k3_cycles_fill += tops[20]; a->st.k3_cycles_leaf += tops[21]; a->st.k3_cycles_trace += tops[22]; a->st.k3_cycles_total += tops[23]; a->st.k3_cycles_max += tops[17]; a->st.k3_waves = a->k3_waves; a->st.k3_cycles_next += tops[19];
	if(a->k2_leaf_shift > 0 && tops[29] * 50 > (unsigned long long)n_reads) { for(mm_align_t *q = a; q; q = q->sib) { if(q->k2_leaf_shift > 0) q->k2_leaf_shift--; } }
	a->st.k2_cycles_sort += tops[24] + tops[28]; a->st.k2_cycles_chain += tops[25]; a->st.k2_cycles_total += tops[26] + tops[28]; a->st.k2_reads_hbm += tops[27];
	a->st.reads += n_reads; for(uint32_t i = 0; i < n_reads; i++) a->st.bases += b.lens[i];
	double t0 = now_ms();
	/* Host copies of the result pools (uninitialised storage: the copies fill them); pinned when the caller lends a set.  K4 made the CIGAR strings of every alignment
	 * behind the launch that recorded it (run_rounds, on a side stream): the text and an (offset, length) pair per segment come back instead of the path words.  The pools
	 * start on their way first; K4's counters, the pairs and the text follow when K4 is through (k4e).  A batch whose strings did not fit the text buffer, a run with MD
	 * tags, the other formats and the mm_reg_t entries take the path words, and the host walks them as before */
	const uint64_t n_root = std::min<uint64_t>(tops[2], a->root_pool.n), n_bin = std::min<uint64_t>(tops[3], a->bin_pool.n), n_aln = std::min<uint64_t>(tops[4], a->aln_pool.n);
	const uint64_t n_seg = std::min<uint64_t>(tops[5], a->seg_pool.n), n_path = std::min<uint64_t>(tops[6], a->path_pool.n);
	bool dev_cigar = device_cigar(a) && !b.regs && a->k4_pending && a->cig_ent.p && a->cig_text.p && n_seg > 0 && n_seg <= a->cig_ent.n;
	auto pinned = [&](int i, size_t need) -> void * { void *q = f.pin ? f.pin->get(i, need) : nullptr; if(!q) { f.own[i].reset(new uint8_t[need + 16]); q = f.own[i].get(); } return q; };
	f.root = (Root *)pinned(0, (size_t)std::max<uint64_t>(n_root, 1) * sizeof(Root)); f.bin = (uint64_t *)pinned(1, (size_t)std::max<uint64_t>(n_bin, 1) * 8);
	f.aln = (AlnRec *)pinned(2, (size_t)std::max<uint64_t>(n_aln, 1) * sizeof(AlnRec)); f.seg = (gaba::Segment *)pinned(3, (size_t)std::max<uint64_t>(n_seg, 1) * sizeof(gaba::Segment));
	f.path = nullptr; f.cig_ent = nullptr; f.cig_text = nullptr;
	f.d2h_bytes = n_root * sizeof(Root) + n_bin * 8 + n_aln * sizeof(AlnRec) + n_seg * sizeof(gaba::Segment);
	CK(hipMemcpyAsync(f.root, a->root_pool.p, n_root * sizeof(Root), hipMemcpyDeviceToHost, a->stream));
	CK(hipMemcpyAsync(f.bin, a->bin_pool.p, n_bin * 8, hipMemcpyDeviceToHost, a->stream));
	CK(hipMemcpyAsync(f.aln, a->aln_pool.p, n_aln * sizeof(AlnRec), hipMemcpyDeviceToHost, a->stream));
	CK(hipMemcpyAsync(f.seg, a->seg_pool.p, n_seg * sizeof(gaba::Segment), hipMemcpyDeviceToHost, a->stream));
	unsigned long long cig_ctl[4] = { 0, 0, 0, 0 };
	if(a->k4_pending) {
		CK(hipStreamWaitEvent(a->stream, a->k4e, 0)); a->k4_pending = false;
		if(dev_cigar) { CK(hipMemcpyAsync(cig_ctl, a->d_tops.p + 36, sizeof(cig_ctl), hipMemcpyDeviceToHost, a->stream)); CK(hipStreamSynchronize(a->stream)); }
	}
	if(dev_cigar && cig_ctl[2] != 0) { dev_cigar = false; if(getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd]   CIGAR strings of a batch beyond %.1f MB of text: made by the host\n", a->cig_text.n / 1e6); }
	if(dev_cigar) {
		f.cig_ent = (const CigEnt *)pinned(5, (size_t)n_seg * sizeof(CigEnt)); f.cig_text = (const char *)pinned(6, (size_t)cig_ctl[1] + 16);
		f.path = (uint32_t *)pinned(4, 64);
		CK(hipMemcpyAsync((void *)f.cig_ent, a->cig_ent.p, n_seg * sizeof(CigEnt), hipMemcpyDeviceToHost, a->stream));
		if(cig_ctl[1]) { CK(hipMemcpyAsync((void *)f.cig_text, a->cig_text.p, cig_ctl[1], hipMemcpyDeviceToHost, a->stream)); }
		f.d2h_bytes += n_seg * sizeof(CigEnt) + cig_ctl[1];
	} else {
		f.path = (uint32_t *)pinned(4, (size_t)(std::max<uint64_t>(n_path, 2) + 8) * 4);
		CK(hipMemcpyAsync(f.path, a->path_pool.p, n_path * 4, hipMemcpyDeviceToHost, a->stream));
		f.d2h_bytes += n_path * 4;
	}
	CK(hipStreamSynchronize(a->stream));
	a->st.host_post_ms += now_ms() - t0; a->st.d2h_bytes += f.d2h_bytes; if(dev_cigar) { a->st.cigar_bytes_device += cig_ctl[1]; }
	if(getenv("MM_VERBOSE") && dev_cigar) fprintf(stderr, "[minialign_amd]   CIGAR text made on the device: %llu segments, %.1f MB (the path words it stands for: %.1f MB); %.1f MB back in all\n", cig_ctl[0], cig_ctl[1] / 1e6, n_path * 4 / 1e6, f.d2h_bytes / 1e6);
	return true;
}
/* a read of a scanned text as the printers want it: name, comment, base codes and qualities read off the text, as bseq_read_fasta leaves them (minialign.c:1996-2090) */
void materialize(const Batch &b, uint32_t i, HSeq &r, bool keep_qual, bool keep_comment)
{
	static const uint8_t enc[16] = { 0, 0, 0, 1, 3, 3, 0, 2, 0, 0, 0, 0, 0, 0, 4, 0 };          /* low nibble: A 1, C 3, T 4, U 5, G 7, N 14 (minialign.c:223-229) */
	const RRec &q = b.trec[i]; const char *t = b.tsrc->p, *end = t + b.tsrc->n;
	r.name.clear(); r.comment.clear(); r.has_comment = false; r.qual.clear();
	parse_header(t + q.start + 1, end, r, keep_comment);
	r.seq.resize(q.n_bases);
	{ uint8_t *d = r.seq.data(); const char *p = t + q.t_off, *e = p + q.t_len; uint32_t k = 0; for(; p < e && k < q.n_bases; p++) { if(*p != '\n') d[k++] = enc[*p & 15]; } }
	if(keep_qual && q.q_len) {
		/* quality lines: a CR at the end of a line is dropped, lines are joined (minialign.c:2060-2068) */
		const char *p = t + q.q_off, *e = p + q.q_len;
		while(p < e) { const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p)); const char *le = nl ? nl : e; size_t kl = (size_t)(le - p); if(kl > 0 && p[kl - 1] == '\r') kl--; r.qual.append(p, kl); p = nl ? nl + 1 : e; }
	}
}
/* finish, second half (host only): post-map and the output text of every read.  Reads are independent: host threads take contiguous spans, the pieces are
 * joined in input order (mm_align_drain keeps the same order with its heap, minialign.c:4633-4645).  max_threads = 0: -t, or up to 32. */
void batch_format(const mm_align_t *a, Batch &b, const Fetched &f, std::vector<std::string> &piece_out, uint32_t max_threads, std::vector<std::vector<uint32_t>> *read_off = nullptr)
{
	const uint32_t n_reads = b.n; const std::vector<ReadState> &hst = b.hst;
	const uint32_t want = max_threads ? max_threads : (a->o.nth > 1 ? a->o.nth : std::min<uint32_t>(std::max<uint32_t>(1, std::thread::hardware_concurrency()), 32));
	const uint32_t nth = std::max<uint32_t>(1, std::min<uint32_t>(want, std::max<uint32_t>(1, n_reads / 64)));
	std::vector<std::string> piece(std::move(piece_out)); piece_out.clear();          /* strings handed in keep their capacity */
	piece.resize(nth); for(auto &x : piece) x.clear();
	if(read_off) { read_off->assign(nth, std::vector<uint32_t>()); }          /* where the records of each read begin in its piece (the head of a stream: mm_head_offset) */
	Root *root = f.root; uint64_t *bin = f.bin; const AlnRec *aln = f.aln; const gaba::Segment *seg = f.seg; const uint32_t *path = f.path;
	/* spans of equal bases (not equal read counts): the text of a read grows with its length */
	std::vector<uint32_t> cut(nth + 1, n_reads);
	{ uint64_t tot = 0; for(uint32_t i = 0; i < n_reads; i++) tot += b.lens[i] + 256; uint64_t acc = 0; uint32_t t = 0; cut[0] = 0; for(uint32_t i = 0; i < n_reads && t + 1 < nth; i++) { acc += b.lens[i] + 256; if(acc * nth >= tot * (t + 1)) { cut[++t] = i + 1; } } }
	auto span = [&](uint32_t t) {
		const uint32_t lo = cut[t], hi = cut[t + 1];
		std::string &out = piece[t];
		uint64_t est = 0; for(uint32_t i = lo; i < hi; i++) est += b.lens[i];
		out.reserve(est + est / 2 + 4096);
		HSeq tmp;
		for(uint32_t i = lo; i < hi; i++) {
			if(read_off) { (*read_off)[t].push_back((uint32_t)out.size()); }
			OutReg reg; const ReadState &rs = hst[i];
			const AlnRec *alns = rs.bin_off != ~0ull ? &aln[rs.aln_off] : aln;
			if(rs.n_res > 0) post_map(a, rs, &root[rs.root_off], &bin[rs.bin_off], alns, reg);
			if(b.regs) { b.regs[i] = build_reg(reg, alns, seg, path); continue; }
			const char *qname; const uint8_t *qseq; const HSeq *qrec;
			if(b.tsrc) { materialize(b, i, tmp, a->o.keep_qual, (a->o.ptags() >> 1) & 1); qname = tmp.name.c_str(); qseq = tmp.seq.data(); qrec = &tmp; }
			else { qname = b.names[i].c_str(); qseq = b.seq[i]; qrec = i < b.rec.size() ? b.rec[i] : nullptr; }
			if(a->o.format == 0) sam_record(a, out, qname, qseq, b.lens[i], reg, alns, seg, path, qrec, f.cig_ent, f.cig_text);
			else alt_record(a, out, qname, qseq, b.lens[i], reg, alns, seg, path);
		}
	};
	std::vector<std::thread> th;
	for(uint32_t t = 1; t < nth; t++) th.emplace_back(span, t);
	span(0);
	for(auto &x : th) x.join();
	for(auto &x : piece) piece_out.emplace_back(std::move(x));
}
bool batch_finish_pieces(mm_align_t *a, Batch &b, std::vector<std::string> &piece_out)
{
	Fetched f;
	if(!batch_fetch(a, b, f)) return false;
	double t1 = now_ms();
	batch_format(a, b, f, piece_out, 0);
	a->st.host_sam_ms += now_ms() - t1;
	return true;
}
bool batch_finish(mm_align_t *a, Batch &b, std::string &sam)
{
	std::vector<std::string> piece;
	if(!batch_finish_pieces(a, b, piece)) return false;
	size_t tot = sam.size(); for(auto &x : piece) tot += x.size();
	sam.reserve(tot);
	for(auto &x : piece) sam += x;
	return true;
}
} /* anonymous */


extern "C" int mm_align_batch(mm_align_t *a, uint8_t const *bases, uint32_t const *lens, char const *const *names, uint32_t n_reads, char **sam, uint64_t *sam_len)
{
	Batch b; uint64_t off = 0; char nbuf[32];
	for(uint32_t i = 0; i < n_reads; i++) {
		b.lens.push_back(lens[i]); b.seq.push_back(bases + off); off += lens[i];
		if(names) b.names.emplace_back(names[i]); else { snprintf(nbuf, sizeof(nbuf), "r%u", i); b.names.emplace_back(nbuf); }
	}
	std::string s;
	if(n_reads && !(batch_prepare(a, b) && batch_run(a, b) && batch_finish(a, b, s))) return -1;
	uint64_t old = *sam ? *sam_len : 0;
	*sam = (char *)realloc(*sam, old + s.size() + 1);
	memcpy(*sam + old, s.data(), s.size()); (*sam)[old + s.size()] = 0; *sam_len = old + s.size();
	return 0;
}

/* the same batch with structured results, what mm_align_seq returns per read (minialign.c:4427, mm_reg_t :3264): regs[i] = NULL for an unmapped read */
extern "C" int mm_align_batch_regs(mm_align_t *a, uint8_t const *bases, uint32_t const *lens, uint32_t n_reads, mm_reg_t **regs)
{
	Batch b; uint64_t off = 0;
	for(uint32_t i = 0; i < n_reads; i++) { b.lens.push_back(lens[i]); b.seq.push_back(bases + off); off += lens[i]; b.names.emplace_back(); regs[i] = nullptr; }
	b.regs = regs;
	std::string s;
	if(n_reads && !(batch_prepare(a, b) && batch_run(a, b) && batch_finish(a, b, s))) { for(uint32_t i = 0; i < n_reads; i++) { free(regs[i]); regs[i] = nullptr; } return -1; }
	return 0;
}
extern "C" void mm_reg_free(mm_reg_t *r) { free(r); }
/* mm_align_seq (minialign.c:4427): one read; a one-read batch on the device.  qid and lmm are accepted for the reference's argument list (it pins qid to 0,
 * minialign.c:3768; the result is one malloc block, released with mm_reg_free).  NULL = unmapped (or error, reported on stderr). */
extern "C" mm_reg_t const *mm_align_seq(mm_align_t *a, uint32_t l_seq, uint8_t const *seq, uint32_t qid, void *lmm)
{
	(void)qid; (void)lmm;
	mm_reg_t *r = nullptr;
	if(!a || !seq || l_seq == 0 || mm_align_batch_regs(a, seq, &l_seq, 1, &r)) return nullptr;
	return r;
}
/* the one value reads share (minialign.c:3864): the length of the reference sequence the previous read loaded last.  A caller that splits one
 * read set over several contexts (processes, devices) hands it from the end of one part to the start of the next. */
extern "C" uint32_t mm_align_get_carry(mm_align_t const *a) { return a->rlen_carry; }
extern "C" void mm_align_set_carry(mm_align_t *a, uint32_t rlen) { each_context(a, [rlen](mm_align_t *q) { q->rlen_carry = rlen; }); }


/* phase-split entry points over a parsed read set (bench.py times mm_batch_run alone: inputs resident in HBM) */
static mm_reads_t *reads_load(char const *fn, uint32_t min_len, bool keep_qual = false, bool keep_comment = false, bool keep_text = false, uint32_t part = 0, uint32_t n_parts = 1);
extern "C" mm_reads_t *mm_reads_load(char const *fn) { return reads_load(fn, 1); }
extern "C" mm_reads_t *mm_reads_load_text(char const *fn) { return reads_load(fn, 1, false, false, true); }          /* keeps the text of the file: batches cut from it are packed on the device */
/* part `part` of `n_parts` of a read file (one rank's shard, minialign_amd/multi.py): a plain FASTA file is cut by bytes at record starts and only that stretch is read;
 * anything else is read whole and the part keeps its share of the records.  The parts in order are the file. */
extern "C" mm_reads_t *mm_reads_load_part(char const *fn, uint32_t part, uint32_t n_parts) { return (n_parts == 0 || part >= n_parts) ? NULL : reads_load(fn, 1, false, false, false, part, n_parts); }
/* the same with the reader's options as the command line has them (-L: shortest read kept, -Q: qualities, -T CO: comments): what the text path (mm_map_text) applies from the
 * context's options, so that one command line gives one output whatever the format of the read file */
extern "C" mm_reads_t *mm_reads_load_part_opt(mm_opt_t const *o, char const *fn, uint32_t part, uint32_t n_parts) { return (n_parts == 0 || part >= n_parts) ? NULL : reads_load(fn, o->min_len, o->keep_qual, (o->ptags() >> 1) & 1, false, part, n_parts); }
static mm_reads_t *reads_load(char const *fn, uint32_t min_len, bool keep_qual, bool keep_comment, bool keep_text, uint32_t part, uint32_t n_parts)
{
	mm_reads_t *r = new mm_reads_s();
	std::shared_ptr<std::vector<char>> tx = keep_text ? std::make_shared<std::vector<char>>() : nullptr;
	if(!read_seq_file(fn, r->r, min_len, keep_qual, keep_comment, tx.get(), 0, part, n_parts)) { delete r; return NULL; }
	if(tx) r->text.push_back(tx);
	for(const HSeq &s : r->r) r->bases += s.seq.size();
	return r;
}
extern "C" void mm_reads_free(mm_reads_t *r) { delete r; }
/* another file behind the reads already loaded (bench.py: the parts of one read set) */
extern "C" int mm_reads_append(mm_reads_t *r, char const *fn)
{
	std::vector<HSeq> more;
	if(!read_seq_file(fn, more, 1, false, false)) return -1;
	for(HSeq &q : more) { r->bases += q.seq.size(); r->r.emplace_back(std::move(q)); }
	return 0;
}
extern "C" char const *mm_reads_name(mm_reads_t const *r, uint32_t i) { return i < r->r.size() ? r->r[i].name.c_str() : NULL; }
extern "C" uint32_t mm_reads_count(mm_reads_t const *r) { return (uint32_t)r->r.size(); }
extern "C" uint64_t mm_reads_bases(mm_reads_t const *r, uint32_t first, uint32_t n) { uint64_t b = 0; for(uint32_t i = first; i < first + n && i < r->r.size(); i++) b += r->r[i].seq.size(); return b; }
/* Lanes: a batch is bound to one lane of the device context (own streams, pools and -- when run asynchronously -- host thread; index,
 * reference and DP constants shared).  Batches on different lanes overlap on the device: the launch tail and the latency-bound
 * stages of one are filled by the other.  The one piece of state reads share, the carried reference length, is
 * handed from batch to batch by whoever sequences them (align_reads below). */
struct mm_batch_s { Batch b; mm_align_t *ctx = nullptr; std::thread th; int rc = 0; bool running = false; uint32_t k = 0; };          /* k: the batch's number in the order of its stream */
static mm_align_t *align_lane(mm_align_t *a)
{
	if(a->sib) return a->sib;
	while(a->is_sib && false) {}
	mm_align_t *q = new mm_align_s();
	q->o = a->o; q->mi = a->mi; q->gctx = a->gctx; q->dix = a->dix; q->d_slot = a->d_slot; q->d_val = a->d_val; q->d_seq_len = a->d_seq_len; q->d_seq_off = a->d_seq_off; q->d_seq_circ = a->d_seq_circ;
	q->root = a->root ? a->root : a; q->ref_ar = a->ref_ar; q->twlen = a->twlen; q->tglen = a->tglen; q->mcoef = a->mcoef; q->xcoef = a->xcoef; q->n_waves = a->n_waves; q->is_sib = true; q->dev = a->dev;
	q->qlen_hint = a->qlen_hint; q->k2_leaf_shift = a->k2_leaf_shift; q->bin_cap = a->bin_cap; q->aln_cap = a->aln_cap; q->kh_cap = a->kh_cap; q->next_cap = a->next_cap; q->rs_stride = a->rs_stride;
	if(!make_streams(q)) { delete q; return NULL; }
	memset(&q->st, 0, sizeof(q->st)); q->t_wall0 = now_ms();
	q->lane_ix = a->lane_ix + 1;
	a->sib = q;
	return q;
}
extern "C" mm_batch_t *mm_batch_upload(mm_align_t *a, mm_reads_t const *r, uint32_t first, uint32_t n)
{
	mm_batch_t *h = new mm_batch_s();
	const uint32_t last = (uint32_t)std::min<uint64_t>((uint64_t)first + n, r->r.size());
	for(uint32_t i = first; i < last; i++) { h->b.lens.push_back((uint32_t)r->r[i].seq.size()); h->b.seq.push_back(r->r[i].seq.data()); h->b.names.push_back(r->r[i].name); h->b.rec.push_back(&r->r[i]); }
	h->ctx = a;
	if(!batch_prepare(a, h->b)) { delete h; return NULL; }
	return h;
}
/* the same on the second lane of the context (own streams and pools): lets a caller keep two batches in flight, see
 * mm_batch_run_async.  Batches on different lanes do not see each other's carried reference length. */
extern "C" mm_batch_t *mm_batch_upload_lane(mm_align_t *a, mm_reads_t const *r, uint32_t first, uint32_t n, int lane)
{
	if(lane == 0) return mm_batch_upload(a, r, first, n);
	if(lane < 0 || lane > 7) return NULL;
	mm_align_t *q = a;
	for(int i = 0; i < lane && q; i++) { q = align_lane(q); }          /* lanes form a chain off the primary context */
	if(!q) return NULL;
	mm_batch_t *h = new mm_batch_s();
	const uint32_t last = (uint32_t)std::min<uint64_t>((uint64_t)first + n, r->r.size());
	for(uint32_t i = first; i < last; i++) { h->b.lens.push_back((uint32_t)r->r[i].seq.size()); h->b.seq.push_back(r->r[i].seq.data()); h->b.names.push_back(r->r[i].name); h->b.rec.push_back(&r->r[i]); }
	h->ctx = q;
	if(!batch_prepare(q, h->b)) { delete h; return NULL; }
	return h;
}
extern "C" int mm_batch_run(mm_align_t *a, mm_batch_t *h)
{
	mm_align_t *c = h->ctx ? h->ctx : a;
	if(h->b.ran) { if(!batch_upload(c, h->b)) return -1; }        /* a second pass over the same batch starts from clean device state */
	return batch_run(c, h->b) ? 0 : -1;
}
extern "C" int mm_batch_finish(mm_align_t *a, mm_batch_t *h, char **sam, uint64_t *sam_len)
{
	std::string s;
	if(!batch_finish(h->ctx ? h->ctx : a, h->b, s)) return -1;
	if(sam) { uint64_t old = *sam ? *sam_len : 0; *sam = (char *)realloc(*sam, old + s.size() + 1); memcpy(*sam + old, s.data(), s.size()); (*sam)[old + s.size()] = 0; *sam_len = old + s.size(); }
	return 0;
}
/* start the hot path of a batch on a host thread of its own and return; mm_batch_wait joins it (0 ok).  Two batches uploaded to
 * different lanes can be in flight together: the launch tail and the latency-bound stages of one are filled by the other. */
extern "C" int mm_batch_run_async(mm_align_t *a, mm_batch_t *h)
{
	if(h->running) return -1;
	h->running = true; h->rc = -1;
	h->th = std::thread([a, h]() { if(hipSetDevice(a->dev) != hipSuccess) { h->rc = -1; return; } h->rc = mm_batch_run(a, h); });
	return 0;
}
extern "C" int mm_batch_wait(mm_align_t *a, mm_batch_t *h)
{
	(void)a;
	if(!h->running) return 0;
	h->th.join(); h->running = false;
	return h->rc;
}
extern "C" void mm_batch_free(mm_batch_t *h) { if(h && h->running) { h->th.join(); } delete h; }
/* stage taps (tests): sketch + lookup + expansion (K1) and sort + chain (K2s, K2p, K2c) of the first round over the batch, nothing behind them; then for one
 * read the number of minimizers K1 emitted, its seed array as mm_seed leaves it (sorted, sentinel last, lid = INT32_MAX; four words per seed: upos, rid,
 * vpos, lid -- mm_seed_t, minialign.c:3187) and its chain roots as mm_chain leaves them (plen | lid << 32, longest first; mm_root_t :3202).  0 on success. */
extern "C" int mm_batch_tap(mm_align_t *a, mm_batch_t *h, uint32_t read, uint32_t *n_min, uint32_t *seeds, uint32_t seeds_cap, uint32_t *n_seeds, uint64_t *roots, uint32_t roots_cap, uint32_t *n_roots)
{
	mm_align_t *c = h->ctx ? h->ctx : a; Batch &b = h->b;
	if(read >= b.n) return -1;
	if(!b.packed) batch_pack(b);
	if(!batch_upload(c, b)) return -1;
	c->tap_stop = true;
	const bool ok = run_rounds(c, b.n, b.work, true, b.hst, nullptr, b.lens);
	c->tap_stop = false;
	if(!ok) return -1;
	const ReadState &rs = b.hst[read];
	if(rs.err) return -2;
	if(n_min) *n_min = rs.n_min;
	const uint32_t ns = rs.n_seed ? rs.n_seed + 1 : 0;
	if(n_seeds) *n_seeds = ns;
	if(seeds && ns) {
		std::vector<Seed> tmp(ns);
		if(hipMemcpy(tmp.data(), c->seed_pool.p + rs.seed_off, (size_t)ns * sizeof(Seed), hipMemcpyDeviceToHost) != hipSuccess) return -1;
		for(uint32_t i = 0; i < ns && i < seeds_cap; i++) { seeds[4 * i] = tmp[i].upos; seeds[4 * i + 1] = tmp[i].rid; seeds[4 * i + 2] = tmp[i].vpos; seeds[4 * i + 3] = 0x7fffffffu; }
	}
	if(n_roots) *n_roots = rs.n_root;
	if(roots && rs.n_root) {
		std::vector<Root> tmp(rs.n_root);
		if(hipMemcpy(tmp.data(), c->root_pool.p + rs.root_off, (size_t)rs.n_root * sizeof(Root), hipMemcpyDeviceToHost) != hipSuccess) return -1;
		for(uint32_t i = 0; i < rs.n_root && i < roots_cap; i++) roots[i] = (uint64_t)tmp[i].plen | ((uint64_t)tmp[i].lid << 32);
	}
	return 0;
}
/* ... and the minimizer stream of that read as mm_sketch leaves it (minialign.c:2402: hash << 8 | strand << 7 | index inside the block of w), straight out of K1; call
 * after mm_batch_tap on the same batch.  Returns the count (at most cap words are written), -1 on error. */
extern "C" int64_t mm_batch_tap_sketch(mm_align_t *a, mm_batch_t *h, uint32_t read, uint64_t *words, uint32_t cap)
{
	mm_align_t *c = h->ctx ? h->ctx : a; Batch &b = h->b;
	if(read >= b.n || b.hst.size() != b.n || !c->tap_words.p) return -1;
	const ReadState &rs = b.hst[read];
	if(rs.min_off + rs.n_min > c->tap_words.n) return -1;
	const uint32_t n = std::min(rs.n_min, cap);
	if(n && hipMemcpy(words, c->tap_words.p + rs.min_off, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return rs.n_min;
}
/* test entry: the run lengths of path bits [ppos, ppos + len) of a path whose first word is pool[path_word] (two header words in front of it), by the very code the
 * device runs (mm_cigar.hpp: cig_write), on the host; returns the characters written (out may be NULL: the count) */
extern "C" uint64_t mm_cigar_walk(uint32_t const *pool, uint64_t path_word, uint64_t ppos, uint64_t len, char *out) { return cig_write(pool, path_word * 32ull + ppos - 64ull, len, out); }
extern "C" int mm_set_device(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : -1; }

static int align_reads(mm_align_t *a, mm_reads_t *reads, FILE *out, bool keep = false);
static int align_text(mm_align_t *a, const std::shared_ptr<TextSrc> &src, const std::function<bool(uint32_t, std::vector<std::string> &)> &sink, int lanes, int pos_fd = -1, uint64_t *pos_at = nullptr);
static std::shared_ptr<TextSrc> open_text_once(const char *fn);
extern "C" int mm_align_file(mm_align_t *a, char const *reads_fn, FILE *out)
{
	const bool verbose = getenv("MM_VERBOSE") != NULL; double tv = now_ms();
	{
		/* the text of the file goes to the device as it is; records are found there (K0r), bases packed there (K0) */
		std::shared_ptr<TextSrc> src = open_text_once(reads_fn);
		if(!src) { fprintf(stderr, "[minialign_amd] cannot read `%s'\n", reads_fn); return 1; }
		/* where the output is a regular file (`minialign ... > out.sam`): the batches' text goes to its place in the file from several threads (stream_map, drain workers);
		 * a pipe, a terminal or a file opened for appending is written in order by the one writer */
		int pos_fd = -1; uint64_t pos_at = 0;
		if(fflush(out) == 0) {
			const int fd = fileno(out); struct stat sb;
			if(fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) { const int fl = fcntl(fd, F_GETFL); const off_t at = lseek(fd, 0, SEEK_CUR); if(fl >= 0 && !(fl & O_APPEND) && at >= 0) { pos_fd = fd; pos_at = (uint64_t)at; } }
		}
		const int rc = align_text(a, src, [&](uint32_t, std::vector<std::string> &piece) { for(auto &x : piece) { if(fwrite(x.data(), 1, x.size(), out) != x.size()) return false; } return true; }, 0, pos_fd, &pos_at);
		if(pos_fd >= 0 && lseek(pos_fd, (off_t)pos_at, SEEK_SET) < 0) return 1;          /* (the stream goes on behind what was written) */
		(void)verbose; (void)tv;
		return rc;
	}
}
/*
 * The streaming engine.  Batches 0 .. n - 1 of a read set go through `lanes` lanes of the device context (each lane: own streams, pools and host thread; index,
 * reference and DP constants shared), several at a time, so that the launch tails and the latency-bound rescue rounds of one batch are filled by the kernels of
 * the others, and the host halves (pack before, post-map + text behind) overlap the device work:
 *
 *   lane thread (one per lane), batch k:   make(k) [pack]  ->  H2D  ->  K1..K3 with the carried reference length *predicted*  ->  wait until batch k - 1 is
 *                                           verified  ->  verify / re-run against the true value (minialign.c:3864)  ->  D2H of the result pools
 *   finisher threads:                       post-map + SAM text of a fetched batch on their share of the host threads
 *   writer thread:                          hands the pieces to the sink strictly in batch order (mm_align_drain, minialign.c:4633-4645)
 *
 * The only value that couples batches, the carried reference length, is final for batch k as soon as batch k - 1 has been verified; a batch that ran ahead
 * with a guess re-runs the reads whose `apos >= rlen` decision the true value changes (batch_verify_carry), which is what a single stream would have computed.
 */
/* a batch that does not fit the device pools however they are grown (settings under which every minimizer has thousands of hits): its reads in halves,
 * one after the other on the same lane, down to single reads -- slow, but the run goes on where the reference's would (it just grinds) */
static bool map_split(mm_align_t *c, const Batch &b, uint32_t lo, uint32_t hi, std::vector<std::string> &pieces)
{
	Batch sub; std::vector<HSeq> held;          /* (reads of a scanned text are read off the text for this rare path) */
	if(b.tsrc) { held.resize(hi - lo); for(uint32_t i = lo; i < hi; i++) materialize(b, i, held[i - lo], c->o.keep_qual, (c->o.ptags() >> 1) & 1); }
	for(uint32_t i = lo; i < hi; i++) {
		if(b.tsrc) { const HSeq &q = held[i - lo]; sub.lens.push_back((uint32_t)q.seq.size()); sub.seq.push_back(q.seq.data()); sub.names.push_back(q.name); sub.rec.push_back(&q); }
		else { sub.lens.push_back(b.lens[i]); sub.seq.push_back(b.seq[i]); sub.names.push_back(b.names[i]); if(i < b.rec.size()) sub.rec.push_back(b.rec[i]); }
	}
	const uint32_t carry_in = c->rlen_carry;
	const bool pretend = getenv("MM_TEST_SPLIT") && hi - lo >= 8;          /* test hook: as if nothing of 8 reads or more fitted */
	if(!pretend && batch_prepare(c, sub) && batch_run(c, sub)) {
		std::vector<std::string> part;          /* (batch_format takes over whatever strings it is handed, for their capacity) */
		if(!batch_finish_pieces(c, sub, part)) return false;
		for(auto &x : part) pieces.emplace_back(std::move(x));
		return true;
	}
	if(hi - lo < 2) { fprintf(stderr, "[minialign_amd] read `%s' does not fit the device pools\n", sub.names.empty() ? "?" : sub.names[0].c_str()); return false; }
	c->rlen_carry = carry_in;
	const uint32_t mid = lo + (hi - lo) / 2;
	return map_split(c, b, lo, mid, pieces) && map_split(c, b, mid, hi, pieces);
}

#include "host_reader.hpp"
/* the reference through the device reader: its text in stretches of 1 GB to HBM, records found there, bases converted and packed there into ONE arena (every sequence
 * on a multiple of 64 bases, N in between); names from the header lines; nothing of a sequence's bases is touched by the host (ref_codes makes codes for the few host
 * consumers on demand) */
static bool ref_to_device(const mm_opt_s *o, mm_idx_s *mi, const char *fn, std::vector<uint64_t> &off, std::vector<uint32_t> &len)
{
	std::shared_ptr<TextSrc> src = open_text(fn);
	if(!src) return false;
	ChunkPool pool; ReaderDev rd; rd.pool = &pool; rd.fastq = src->delim == '@'; rd.keep_qual = false;
	rd.chunk_bytes = 1ull << 30;
	if(!rd.init(std::max<uint32_t>(2, std::min<uint32_t>(12, std::thread::hardware_concurrency() / 4)))) return false;
	DBuf<uint8_t> d_codes; DBuf<TextRead> d_ti; DBuf<uint32_t> d_tn, d_tb;
	const uint64_t codes_cap = src->n + (256ull << 20);          /* bases + what the alignment of the sequences to 64 adds (room for four million sequences) */
	if(!d_codes.ensure(codes_cap) || hipMemsetAsync(d_codes.p, 4, codes_cap, rd.st) != hipSuccess) return false;
	uint64_t at = src->first, want = rd.chunk_bytes, total = 0; bool ok = true;
	while(ok && at < src->n) {
		const uint64_t ln = std::min<uint64_t>(want, src->n - at); const bool last = at + ln == src->n;
		if(ln > 0x7fff0000ull) { fprintf(stderr, "[minialign_amd] reader: a reference sequence of more than 2 GB of text\n"); ok = false; break; }
		DevChunk *c = pool.get(((ln + 63) & ~63ull) + 64);
		if(!c) { ok = false; break; }
		c->off = at; c->n = (uint32_t)ln;
		std::vector<RRec> recs; uint64_t consumed = 0; bool grow = false;
		ok = rd.upload(c->d, src->p + at, ln, rd.st) && hipStreamSynchronize(rd.st) == hipSuccess && rd.scan(src->p + at, c->d, 0, at, ln, last, recs, consumed, grow);
		if(ok && grow) { pool.put(c); want = std::min<uint64_t>(want * 2, 0x7fff0000ull); if(ln == 0x7fff0000ull) ok = false; continue; }
		if(ok) {
			std::vector<TextRead> tr;
			for(const RRec &r : recs) {
				if(r.n_bases < o->min_len) continue;          /* -L applies to the reference side as well (minialign.c:2077) */
				mi->seq.emplace_back(); HSeq &q = mi->seq.back();
				parse_header(src->p + r.start + 1, src->p + src->n, q, false); q.len = r.n_bases;
				mi->rrec.push_back(r); off.push_back(total); len.push_back(r.n_bases);
				tr.push_back(TextRead{ r.t_off - at, r.t_len, 0, total }); total += ((uint64_t)r.n_bases + 63) & ~63ull;
			}
			if(total + 64 > codes_cap) { fprintf(stderr, "[minialign_amd] reader: more reference sequences than the arena was laid out for\n"); ok = false; }
			if(ok && !tr.empty()) {
				ok = d_ti.ensure(tr.size()) && d_tn.ensure(tr.size()) && hipMemcpyAsync(d_ti.p, tr.data(), tr.size() * sizeof(TextRead), hipMemcpyHostToDevice, rd.st) == hipSuccess;
				if(ok && src->delim == '>') {
					/* a wave per 16 KB tile of a sequence's text (a chromosome is 250 MB: one wave per record would walk it for seconds) */
					const uint32_t tile = 16384; std::vector<uint32_t> tb(tr.size() + 1, 0);
					for(size_t i = 0; i < tr.size(); i++) tb[i + 1] = tb[i] + (tr[i].t_len + tile - 1) / tile;
					ok = d_tb.ensure(tb.size()) && hipMemcpyAsync(d_tb.p, tb.data(), tb.size() * 4, hipMemcpyHostToDevice, rd.st) == hipSuccess;
					if(ok && tb.back()) { hipLaunchKernelGGL(mm_text_codes_tiled_kernel, dim3((tb.back() + 3) / 4), dim3(256), 0, rd.st, c->d, rd.d_ma.p, rd.d_cum.p, c->n, d_ti.p, d_tb.p, (uint32_t)tr.size(), tile, d_codes.p); ok = hipGetLastError() == hipSuccess; }
					ok = ok && hipStreamSynchronize(rd.st) == hipSuccess;
				}
				else if(ok) { hipLaunchKernelGGL(mm_text_codes_kernel, dim3((uint32_t)((tr.size() + 3) / 4)), dim3(256), 0, rd.st, c->d, d_ti.p, (uint32_t)tr.size(), d_codes.p, d_tn.p); ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(rd.st) == hipSuccess; }
			}
		}
		pool.put(c);
		if(ok && consumed == 0) ok = false;
		at += consumed; want = rd.chunk_bytes;
	}
	if(!ok) return false;
	const uint64_t n = total + 64, nw = (n + 15) / 16 + 4, nn = (n + 31) / 32 + 4;
	gaba_arena_t *ar = (gaba_arena_t *)calloc(1, sizeof(gaba_arena_t));
	if(!ar || hipMalloc(&ar->pk, nw * 4) != hipSuccess || hipMalloc(&ar->nm, nn * 4) != hipSuccess) { if(ar) { if(ar->pk) (void)hipFree(ar->pk); free(ar); } return false; }
	ar->n = n; ar->host = NULL;
	(void)hipMemsetAsync(ar->pk, 0, nw * 4, rd.st); (void)hipMemsetAsync(ar->nm, 0, nn * 4, rd.st);
	const uint64_t n32 = (n + 31) / 32;
	hipLaunchKernelGGL(mm_codes_pack_kernel, dim3((uint32_t)((n32 + 255) / 256)), dim3(256), 0, rd.st, d_codes.p, n32, ar->pk, ar->nm);
	ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(rd.st) == hipSuccess;
	d_codes.release(); d_ti.release(); d_tn.release(); d_tb.release();
	if(!ok) { gaba_arena_free(ar); return false; }
	mi->ref_ar = ar; mi->rtext = src;
	if(getenv("MM_VERBOSE")) fprintf(stderr, "[minialign_amd] reference reader: %lu sequences, text to HBM + marks %.1f ms, record tables %.1f ms\n", (unsigned long)mi->seq.size(), rd.t_io, rd.t_scan);
	return true;
}
typedef std::function<bool(uint32_t, std::vector<std::string> &)> PieceSink;          /* (batch, pieces) in batch order; false = stop */
/* where the lanes get their batches: take(device slot) hands out the next batch for a lane of that device with its number in the order of the stream (h->k; the numbers
 * of a stream are 0, 1, 2 ... without gaps, the lanes of one device get theirs in rising order), NULL when there is nothing more for that device.  A source over host
 * data (packed or parsed reads) gives any batch to any device; the text reader gives a device the batches whose text is in its HBM. */
typedef std::function<mm_batch_t *(int)> BatchSource;
static BatchSource counted_source(uint32_t n_batches, const std::function<mm_batch_t *(uint32_t)> &make)
{
	auto st = std::make_shared<std::pair<std::mutex, uint32_t>>(); st->second = 0;
	return [st, n_batches, make](int) -> mm_batch_t * {
		std::lock_guard<std::mutex> lk(st->first);
		if(st->second >= n_batches) return nullptr;
		const uint32_t k = st->second++; mm_batch_t *h = make(k); if(h) h->k = k;
		return h;
	};
}
/* the primary contexts of the devices `a` spans, in slot order */
static std::vector<mm_align_t *> device_slots(mm_align_t *a) { (void)ensure_peers(a); std::vector<mm_align_t *> v; v.push_back(a); for(mm_align_t *p : a->peers) v.push_back(p); return v; }
/* n_batches: how many the source holds when that is known (the lanes made are no more than that), 0xffffffff otherwise */
#define MM_OPEN_ENDED 0xffffffffu
static int stream_map(mm_align_t *a, uint32_t n_batches, const BatchSource &source, const std::function<void(mm_batch_t *)> &release,
	const PieceSink &sink, int lanes_want, int pos_fd = -1, uint64_t *pos_at = nullptr)
{
	if(n_batches == 0) { a->head.clear(); a->head_off.assign(1, 0); a->head_off_closed = false; a->head_carry_in = a->rlen_carry; return 0; }
	const bool verbose = getenv("MM_VERBOSE") != NULL;
	/* device slots: every device the context spans (fewer when the source holds fewer batches than that), `lanes` lanes each */
	std::vector<mm_align_t *> prim = device_slots(a);
	if(n_batches < prim.size()) prim.resize(n_batches);
	const int n_dev = (int)prim.size();
	const int lanes = (int)std::max<uint32_t>(1, std::min<uint32_t>({ (uint32_t)lanes_want, (n_batches + (uint32_t)n_dev - 1) / (uint32_t)n_dev, 8u }));
	struct DevSt { mm_align_t *P = nullptr; std::vector<mm_align_t *> ctx; std::mutex claim_mu; uint32_t active = 0; };      /* active: lanes of the device between taking a batch and the end of its D2H */
	std::vector<std::unique_ptr<DevSt>> dv;
	int cur_dev = 0; (void)hipGetDevice(&cur_dev);
	for(int d = 0; d < n_dev; d++) {
		dv.emplace_back(new DevSt()); DevSt &D = *dv.back(); D.P = prim[d];
		if(hipSetDevice(D.P->dev) != hipSuccess) { (void)hipSetDevice(cur_dev); return 1; }
		mm_align_t *q = D.P; for(int i = 0; i < lanes && q; i++) { D.ctx.push_back(q); if(i + 1 < lanes) q = align_lane(q); }
		if((int)D.ctx.size() < lanes || !D.ctx.back()) { (void)hipSetDevice(cur_dev); return 1; }
	}
	{
		/* the shared DP workspaces of every device, side by side (tens of GB each: seconds on memory nobody has touched yet) */
		std::vector<int> okv(n_dev, 1); std::vector<std::thread> st;
		for(int d = 0; d < n_dev; d++) st.emplace_back([&, d]() { okv[d] = hipSetDevice(dv[d]->P->dev) == hipSuccess && ensure_shared_slabs(dv[d]->P, dv[d]->P->qlen_hint, (uint32_t)lanes); });
		for(auto &t : st) t.join();
		for(int d = 0; d < n_dev; d++) if(!okv[d]) { fprintf(stderr, "[minialign_amd] shared DP workspaces: allocation failed\n"); (void)hipSetDevice(cur_dev); return 1; }
	}
	(void)hipSetDevice(cur_dev);
	/* host threads for post-map + text: -t when given, MM_HOST_THREADS, else up to 96 per device; two finishers per device share them */
	const uint32_t hw = std::max<uint32_t>(1, std::thread::hardware_concurrency());
	const uint32_t fmt_total = getenv("MM_HOST_THREADS") ? (uint32_t)std::max(1, atoi(getenv("MM_HOST_THREADS"))) : (a->o.nth > 1 ? a->o.nth : std::min<uint32_t>(hw, 96u * (uint32_t)n_dev));
	const int n_fin = fmt_total >= 8 ? std::min<int>(2 * n_dev, (int)(fmt_total / 4)) : 1;
	struct Item { mm_batch_t *h = nullptr; Fetched f; std::vector<std::string> piece; std::vector<std::vector<uint32_t>> roff; bool split = false; uint32_t k = 0; };
	std::mutex mu; std::condition_variable cv;
	uint32_t verified = 0, next_write = 0, pending = 0; uint32_t carry = a->rlen_carry; int rc = 0;
	PredBoard board;          /* what each batch expects to leave as the carried value, for the first read of the batch behind it */
	std::vector<Item *> fetched;                       /* waiting for a finisher */
	std::map<uint32_t, Item *> formatted;              /* waiting for the writer */
	uint32_t lanes_done = 0, fin_done = 0; bool head_open = true;
	std::deque<std::pair<Item *, uint64_t>> drain_q; bool writer_done = false;          /* a drain with positions: (batch, where its text goes) waiting for a drain worker */
	const uint32_t lanes_total = (uint32_t)(lanes * n_dev);
	a->head.clear(); a->head_txt.clear(); a->head_txt_end = 0; a->head_off.clear(); a->head_off_closed = false; a->head_carry_in = a->rlen_carry;
	for(auto &D : dv) D->P->streaming = true;
	uint64_t written = 0;                              /* bytes handed to the sink so far (writer thread only) */
	const uint32_t max_pending = lanes_total + 2;
	/* an item leaves: its pinned set and its text pieces (emptied, capacity kept) go back to the pools of the context */
	auto drop_item = [&](Item *it) {
		{ std::lock_guard<std::mutex> lk(a->pool_mu); if(it->f.pin) { a->pin_free.push_back(it->f.pin); } if(!it->piece.empty() && a->piece_free.size() < 16u * (uint32_t)n_dev) { a->piece_free.emplace_back(std::move(it->piece)); } }
		delete it;
	};

	const double t_engine0 = now_ms();
	auto lane_main = [&](int di, int li) {
		DevSt &D = *dv[di]; mm_align_t *P = D.P; mm_align_t *c = D.ctx[li];
		if(hipSetDevice(P->dev) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); rc = 1; lanes_done++; cv.notify_all(); return; }
		while(true) {
			uint32_t k = 0, guess;
			double tv = now_ms();
			mm_batch_t *h = nullptr;
			{
				/* the batches of a device are taken one lane at a time, in order: a batch with a longer read than the shared DP workspaces of the device were sized for has
				 * them sized again before any later batch starts there -- once the batches in front of it have left the device */
				std::lock_guard<std::mutex> cl(D.claim_mu);
				{ std::lock_guard<std::mutex> lk(mu); if(rc) break; }
				tv = now_ms();
				h = source(di);
				if(!h) break;
				k = h->k;
				if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): taken after %.1f ms (at %.1f)\n", k, di, li, now_ms() - tv, now_ms() - t_engine0); }
				if(P->shared_slabs && slab_bytes_for(std::max(h->b.max_qlen, P->qlen_hint)) > P->slab_max) {
					std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return D.active == 0 || rc != 0; });
					const uint32_t want = h->b.max_qlen;
					if(rc == 0 && !ensure_shared_slabs(P, want, (uint32_t)lanes)) { fprintf(stderr, "[minialign_amd] shared DP workspaces: allocation failed\n"); rc = 1; }
					for(mm_align_t *ln = P; ln; ln = ln->sib) ln->qlen_hint = std::max(ln->qlen_hint, want);
					if(verbose) fprintf(stderr, "[minialign_amd] batch %u: a read of %u bases, DP workspaces of device %d sized again\n", k, h->b.max_qlen, di);
				}
				std::lock_guard<std::mutex> lk(mu); D.active++; guess = carry;
			}
			struct Leave { std::mutex &m; std::condition_variable &c; uint32_t &n; bool armed = true; void now() { if(armed) { { std::lock_guard<std::mutex> lk(m); n--; } c.notify_all(); armed = false; } } ~Leave() { now(); } } leave{ mu, cv, D.active };
			bool ok = true;
			{
				h->ctx = c; Batch &b = h->b;
				if(!b.packed) batch_pack(b, false);
				c->rlen_carry = guess; c->pred = &board; c->pred_k = k;
				ok = batch_upload(c, b);
				if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): pack + upload %.1f ms (at %.1f)\n", k, di, li, now_ms() - tv, now_ms() - t_engine0); tv = now_ms(); }
				/* run ahead with the predicted carry; pools that overflow are grown here, before anybody waits for this batch */
				bool split = false; const double k1_0 = c->st.k1_ms, k2_0 = c->st.k2_ms, k3_0 = c->st.k3_ms;
				if(getenv("MM_TEST_SPLIT") && b.n >= 8) { split = true; }          /* test hook: take the path of a batch the pools cannot hold */
				/* a batch whose pools the device cannot hold (at its size, or after they were grown): not the end of the stream -- its reads go in halves (map_split) */
				if(!ok && b.n >= 2) { fprintf(stderr, "[minialign_amd] batch %u: its pools do not fit the device, mapped in halves\n", k); ok = true; split = true; }
				while(ok && !split) { int r = batch_run_spec(c, b); if(r == 0) break; if(r < 0) ok = false; else if(r == 2) { if(!batch_upload(c, b)) ok = false; } else if(!batch_grow(c, b)) split = true; else if(!batch_upload(c, b)) { if(b.n >= 2) { split = true; } else { ok = false; } } }
				if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): run %.1f ms (at %.1f): sketch %.1f, sort + chain %.1f, extension %.1f ms on the device, the rest the host's turns in between\n", k, di, li, now_ms() - tv, now_ms() - t_engine0, c->st.k1_ms - k1_0, c->st.k2_ms - k2_0, c->st.k3_ms - k3_0); tv = now_ms(); }
				c->pred = nullptr; board.leave(k);          /* (from here on the lane runs with the true value; a batch that has posted nothing by now -- one that goes in halves -- will not) */
				uint32_t truth = 0;
				{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return verified == k || rc != 0; }); if(rc) ok = false; truth = carry; }
				while(ok && !split) {
					c->rlen_carry = truth;
					int r = batch_verify_carry(c, b);
					if(r == 0) break;
					/* an overflow among the re-runs: the whole batch again with larger pools, now with the true value from the start */
					if(r < 0) { ok = false; break; }
					if(r != 2 && !batch_grow(c, b)) { split = true; break; }          /* (2: a re-run was called off by the watchdog -- the whole batch again as it is) */
					if(!batch_upload(c, b)) { ok = false; break; }
					r = batch_run_spec(c, b); if(r < 0) { ok = false; }
					while(ok && r == 2) { if(!batch_upload(c, b)) { ok = false; break; } r = batch_run_spec(c, b); if(r < 0) { ok = false; } }
				}
				if(ok && split) {
					/* the pools cannot hold this batch: its reads in halves on this lane, in order, with the true carried value; the text goes straight to the writer */
					Item *it = new Item(); it->h = h; it->k = k; it->split = true;
					c->rlen_carry = truth; b.scale = 1; c->st.batch_splits++;
					ok = map_split(c, b, 0, b.n, it->piece);
					if(ok) {
						{ std::lock_guard<std::mutex> lk(mu); carry = c->rlen_carry; verified = k + 1; pending++; formatted[k] = it; if(k == 0) { a->head.clear(); a->head_carry_in = truth; } head_open = false; }          /* (the reads of a split batch are not recorded: the head ends in front of them) */
						cv.notify_all();
					} else { delete it; }
				}
				else if(ok) {
					b.ran = true;
					if(const char *cfn = getenv("MM_DUMP_CARRY")) {          /* diagnostics: what every read ran with and left behind (the carried reference length), in batch order */
						static std::mutex dmu; std::lock_guard<std::mutex> dl(dmu);
						if(FILE *fp = fopen(cfn, k == 0 ? "w" : "a")) {
							for(uint32_t i = 0; i < b.n; i++) {
								std::string nm; if(b.tsrc) { const char *q = b.tsrc->p + b.trec[i].start + 1, *e = b.tsrc->p + b.tsrc->n; while(q < e && *q != ' ' && *q != '\t' && *q != '\n') nm.push_back(*q++); } else if(i < b.names.size()) nm = b.names[i];
								fprintf(fp, "%s\t%u\t%u\t%u\t%d\t%u\n", nm.c_str(), b.used[i], b.hst[i].apos0, b.hst[i].cond0, b.hst[i].rid_last == gaba::NIL ? -1 : (int)b.hst[i].rid_last, b.hst[i].n_res);
							}
							fclose(fp);
						}
					}
					const uint32_t out = batch_carry_out(c, b, truth);
					{
						std::lock_guard<std::mutex> lk(mu); carry = out; verified = k + 1;
						if(k == 0) { a->head.clear(); a->head_txt.clear(); a->head_carry_in = truth; }
						if(b.tsrc) { a->head_txt_end = b.tsrc->n; }
						for(uint32_t i = 0; i < b.n && a->head.size() < 4096 && head_open; i++) { a->head.push_back(mm_align_s::HeadRec{ b.hst[i].apos0, b.hst[i].cond0, b.used[i], b.hst[i].rid_last }); a->head_txt.push_back(b.tsrc ? (uint64_t)(b.trec[i].start) : ~0ull); }
						if(a->head.size() >= 4096) head_open = false;
					}
					cv.notify_all();
					if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): carry wait + verify %.1f ms (at %.1f)\n", k, di, li, now_ms() - tv, now_ms() - t_engine0); tv = now_ms(); }
					/* the batch that is written next never waits here: everything queued in front of the writer is behind it */
					{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return pending < max_pending || k == next_write || rc != 0; }); if(rc) ok = false; else pending++; }
					if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): wait for the writer %.1f ms (at %.1f)\n", k, di, li, now_ms() - tv, now_ms() - t_engine0); tv = now_ms(); }
				}
				if(ok && !split) {
					Item *it = new Item(); it->h = h; it->k = k;
					{ std::lock_guard<std::mutex> lk(a->pool_mu); if(!a->pin_free.empty()) { it->f.pin = a->pin_free.back(); a->pin_free.pop_back(); } if(!a->piece_free.empty()) { it->piece = std::move(a->piece_free.back()); a->piece_free.pop_back(); } }
					if(!it->f.pin) it->f.pin = new mm_align_s::PinSet();
					ok = batch_fetch(c, b, it->f);
					leave.now();
					if(verbose) { fprintf(stderr, "[minialign_amd] batch %u (device %d lane %d): D2H %.1f ms (at %.1f)\n", k, di, li, now_ms() - tv, now_ms() - t_engine0); }
					if(ok) { std::lock_guard<std::mutex> lk(mu); fetched.push_back(it); } else { drop_item(it); std::lock_guard<std::mutex> lk(mu); pending--; }
					cv.notify_all();
				}
			}
			if(!ok) { std::lock_guard<std::mutex> lk(mu); rc = 1; release(h); cv.notify_all(); break; }
		}
		{ std::lock_guard<std::mutex> lk(mu); lanes_done++; }
		cv.notify_all();
	};
	auto fin_main = [&]() {
		while(true) {
			Item *it = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return !fetched.empty() || lanes_done == lanes_total; });
				if(fetched.empty()) break;
				/* the oldest batch first */
				size_t best = 0; for(size_t i = 1; i < fetched.size(); i++) if(fetched[i]->k < fetched[best]->k) best = i;
				it = fetched[best]; fetched.erase(fetched.begin() + best);
			}
			double tv = now_ms();
			batch_format(it->h->ctx, it->h->b, it->f, it->piece, std::max<uint32_t>(1, fmt_total / n_fin), &it->roff);
			{ std::lock_guard<std::mutex> lk(a->pool_mu); if(it->f.pin) { a->pin_free.push_back(it->f.pin); it->f.pin = nullptr; } }
			{ std::lock_guard<std::mutex> lk(mu); it->h->ctx->st.host_sam_ms += now_ms() - tv; formatted[it->k] = it; }
			if(verbose) { fprintf(stderr, "[minialign_amd] batch %u: post-map + text %.1f ms\n", it->k, now_ms() - tv); }
			cv.notify_all();
		}
		{ std::lock_guard<std::mutex> lk(mu); fin_done++; }
		cv.notify_all();
	};
	auto writer_main = [&]() {
		while(true) {
			Item *it = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return formatted.count(next_write) || fin_done == (uint32_t)n_fin; });
				auto f = formatted.find(next_write);
				if(f == formatted.end()) break;
				it = f->second; formatted.erase(f);
			}
			double tv = now_ms(); size_t nb = 0; for(auto &x : it->piece) nb += x.size();
			/* where the records of the first reads of the stream begin (mm_head_offset): as long as every batch so far came with its per-read offsets */
			if(!a->head_off_closed) {
				if(it->split || it->roff.size() != it->piece.size()) { a->head_off_closed = true; }
				else { uint64_t at = written; for(size_t t = 0; t < it->piece.size() && a->head_off.size() <= 4096; t++) { for(uint32_t o : it->roff[t]) { if(a->head_off.size() > 4096) break; a->head_off.push_back(at + o); } at += it->piece[t].size(); } }
			}
			const uint64_t at0 = written;
			written += nb;
			bool failed; { std::lock_guard<std::mutex> lk(mu); failed = rc != 0; }
			if(pos_fd >= 0 && !failed) {
				/* a drain with positions (a regular file): this thread only says where the batch goes -- the sizes of everything in front of it are known -- and the bytes are
				 * written by the drain workers side by side, so no one thread touches every byte of the output (the reference's drain is one thread: minialign.c:4633-4645; at eight
				 * devices that thread would have 50 GB/s of text to move) */
				{ std::lock_guard<std::mutex> lk(mu); next_write++; drain_q.emplace_back(it, *pos_at + at0); }
				cv.notify_all();
				continue;
			}
			const bool ok = !failed ? sink(it->k, it->piece) : true;
			if(verbose) { fprintf(stderr, "[minialign_amd] batch %u: written %.1f MB in %.1f ms\n", it->k, nb * 1e-6, now_ms() - tv); }
			release(it->h); drop_item(it);
			{ std::lock_guard<std::mutex> lk(mu); next_write++; pending--; if(!ok) rc = 1; }
			cv.notify_all();
		}
		{ std::lock_guard<std::mutex> lk(mu); writer_done = true; }
		cv.notify_all();
		/* whatever is left after an error */
		std::lock_guard<std::mutex> lk(mu);
		for(auto &kv : formatted) { release(kv.second->h); drop_item(kv.second); } formatted.clear();
	};
	auto drain_main = [&]() {
		while(true) {
			Item *it = nullptr; uint64_t at = 0;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return !drain_q.empty() || writer_done; });
				if(drain_q.empty()) break;
				it = drain_q.front().first; at = drain_q.front().second; drain_q.pop_front();
			}
			double tv = now_ms(); size_t nb = 0; bool ok = true;
			for(auto &x : it->piece) {
				const char *q = x.data(); size_t left = x.size();
				while(left && ok) { const ssize_t w = pwrite(pos_fd, q, left, (off_t)at); if(w <= 0) { if(w < 0 && errno == EINTR) continue; ok = false; break; } q += w; left -= (size_t)w; at += (uint64_t)w; nb += (size_t)w; }
			}
			if(verbose) { fprintf(stderr, "[minialign_amd] batch %u: written %.1f MB in %.1f ms (at its place in the file)\n", it->k, nb * 1e-6, now_ms() - tv); }
			release(it->h); drop_item(it);
			{ std::lock_guard<std::mutex> lk(mu); pending--; if(!ok) rc = 1; }
			cv.notify_all();
		}
	};
	std::vector<std::thread> th;
	for(int d = 0; d < n_dev; d++) for(int i = 0; i < lanes; i++) th.emplace_back(lane_main, d, i);
	for(int i = 0; i < n_fin; i++) th.emplace_back(fin_main);
	th.emplace_back(writer_main);
	if(pos_fd >= 0) { for(int i = 0; i < std::min(8, 2 * n_dev); i++) th.emplace_back(drain_main); }
	for(auto &t : th) t.join();
	if(pos_fd >= 0 && pos_at) { *pos_at += written; }
	for(Item *it : fetched) { release(it->h); drop_item(it); }
	if(rc == 0) { each_context(a, [carry](mm_align_t *q) { q->rlen_carry = carry; }); }
	if(!a->head_off_closed && a->head_off.size() <= 4096) { a->head_off.push_back(written); }          /* a stream shorter than the head: its end */
	for(auto &D : dv) { D->P->streaming = false; D->P->safe_mode.store(false); }          /* (the safe mode the watchdog switched on lasts to the end of the stream) */
	if(getenv("MM_VERBOSE")) { for(auto &D : dv) { (void)hipSetDevice(D->P->dev); size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot); fprintf(stderr, "[minialign_amd] device %d memory at the end of the stream: %.1f of %.1f GB free; %.1f GB held in recycled buffers\n", D->P->dev, fr / 1073741824.0, tot / 1073741824.0, dev_cache().held_on(D->P->dev) / 1073741824.0); } }
	(void)hipSetDevice(cur_dev);
	return rc;
}
static void batch_fill(mm_batch_t *h, mm_reads_t const *r, uint32_t first, uint32_t last)
{
	h->b.text_src = r->text.empty() ? nullptr : &r->text;
	for(uint32_t i = first; i < last; i++) { h->b.lens.push_back((uint32_t)r->r[i].seq.size()); h->b.seq.push_back(r->r[i].seq.data()); h->b.names.push_back(r->r[i].name); h->b.rec.push_back(&r->r[i]); }
}
static int default_lanes() { return getenv("MM_LANES") ? std::max(1, atoi(getenv("MM_LANES"))) : 4; }
/* batch boundaries of a read set: bounded by bases (so that the device pools stay modest) and by reads */
static std::vector<std::pair<uint32_t, uint32_t>> batch_spans(mm_reads_t const *reads, uint32_t first, uint32_t n)
{
	const uint32_t max_reads = 1u << 17;
	const uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)first + n, reads->r.size());
	/* 300 Mb per batch (up to 500 Mb for sets of more than five batches per lane): on the whole hg38-size x3 set anything between 250 and 512 Mb measured the same until round 6, on an eighth or a quarter of it (one rank's shard of
	 * the multi-GPU job) the smaller batches give every lane one and are 8 - 13 % faster; a set smaller than lanes x 300 Mb is cut into one batch per lane, not
	 * below 64 Mb.  A set with very long reads wants larger batches: an extension launch lasts at least as long as its longest read (one wave, ~0.8 us per base), and
	 * every batch pays that tail again -- 1 700 bases of batch per base of the longest read (ONT-like set, longest read 385 kb: round 2 measured 2.40 s per step at 300 Mb,
	 * 1.58 at 512 Mb, 1.46 at 800 Mb, profiles/round2_batch_size.txt; with the retry jobs of round 3, 1.46 s at 480 Mb, 1.14 at 640 Mb, 1.26 at 960 Mb -- and the lanes' pools
	 * take 46 bytes of HBM per base of batch, so 960 Mb batches on 4 lanes left 2 GB of the 288 free).  MM_BATCH_BASES: test hook (many small batches) */
	uint64_t max_bases = 300000000ull;
	if(getenv("MM_BATCH_BASES")) { max_bases = (uint64_t)atoll(getenv("MM_BATCH_BASES")); }
	else {
		uint64_t total = 0, longest = 0; for(uint32_t i = first; i < end; i++) { total += reads->r[i].seq.size(); longest = std::max<uint64_t>(longest, reads->r[i].seq.size()); }
		max_bases = std::max<uint64_t>(max_bases, std::min<uint64_t>(700000000ull, longest * MM_BATCH_PER_LONGEST));
		const uint64_t lanes = (uint64_t)default_lanes();
		if(total < lanes * max_bases) max_bases = std::max<uint64_t>(64ull << 20, total / (lanes + lanes / 2) + (1ull << 20));
		else max_bases = std::max<uint64_t>(max_bases, std::min<uint64_t>(500000000ull, total / (lanes * 5)));          /* (more than five batches per lane: larger ones, as the text reader cuts them -- TextReader::start) */
	}
	std::vector<std::pair<uint32_t, uint32_t>> sp;
	for(uint32_t i = first; i < end;) {
		uint32_t j = i; uint64_t nb = 0;
		while(j < end && j - i < max_reads && (nb == 0 || nb + reads->r[j].seq.size() <= max_bases)) { nb += reads->r[j].seq.size(); j++; }
		sp.emplace_back(i, j); i = j;
	}
	return sp;
}
/* maps a parsed read set (consumed unless keep) and writes its records */
static int align_reads(mm_align_t *a, mm_reads_t *reads, FILE *out, bool keep)
{
	const uint32_t n = mm_reads_count(reads);
	{ uint32_t mx = 0; for(const HSeq &q : reads->r) mx = std::max<uint32_t>(mx, (uint32_t)q.seq.size()); each_context(a, [mx](mm_align_t *ln) { ln->qlen_hint = std::max(ln->qlen_hint, mx); }); }
	const auto sp = batch_spans(reads, 0, n);
	const int rc = stream_map(a, (uint32_t)sp.size(),
		counted_source((uint32_t)sp.size(), [&](uint32_t k) { mm_batch_t *h = new mm_batch_s(); batch_fill(h, reads, sp[k].first, sp[k].second); return h; }),
		[](mm_batch_t *h) { mm_batch_free(h); },
		[&](uint32_t, std::vector<std::string> &piece) { for(auto &x : piece) { if(fwrite(x.data(), 1, x.size(), out) != x.size()) return false; } return true; },
		default_lanes());
	if(!keep) mm_reads_free(reads);
	return rc;
}
/* the same engine behind the C-ABI: packed batches prepared ahead of time (bench.py: the timed region then starts from packed reads in host memory) or packed on
 * the fly, text handed to a callback in input order */
static int default_lanes();
/* the largest batch the device memory allows on `lanes` lanes: the pools of a lane take about 52 bytes per base of its batches (seeds and the sweep's scratch are most of it,
 * ensure_pools); what they may take together is what was free when the first stream of the context started, less the DP workspaces at their largest (1.75 x the budget:
 * the ordinary class and a ladder of four above it) and 10 GB for the runtime (kernel scratch) and the text */
static uint64_t batch_cap_bases(mm_align_t *a, int lanes)  /* (bytes per base: 52 with the pools sized by caps, about 24 since they follow the demand) */
{
	mm_align_s *P = a->root ? a->root : a;
	if(!P->mem_for_batches) {
		size_t fr = 0, tot = 0; if(hipMemGetInfo(&fr, &tot) != hipSuccess) return 1000000000ull;
		const uint64_t avail = fr + dev_cache().held_on(P->dev), slab_budget = (getenv("MM_SLAB_GB") ? (uint64_t)atoll(getenv("MM_SLAB_GB")) : 64ull) << 30;
		const uint64_t taken = slab_budget + slab_budget * 3 / 4 + (10ull << 30) - std::min<uint64_t>(P->shared_slabs ? P->slabs.bytes : 0, slab_budget);          /* (workspaces already allocated are no longer in `avail`) */
		P->mem_for_batches = avail > taken + (8ull << 30) ? avail - taken : (8ull << 30);
	}
	return std::max<uint64_t>(128ull << 20, P->mem_for_batches / (uint64_t)std::max(1, lanes) / 28);
}
static int align_text(mm_align_t *a, const std::shared_ptr<TextSrc> &src, const PieceSink &sink, int lanes, int pos_fd, uint64_t *pos_at)
{
	if(lanes <= 0) lanes = default_lanes();
	if(!a->chunk_pool) a->chunk_pool = new ChunkPool();
	TextReader rd; rd.dctx = device_slots(a); rd.src = src; rd.min_len = a->o.min_len; rd.keep_qual = a->o.keep_qual; rd.lanes = lanes;
	rd.cap_bases = std::min<uint64_t>(1000000000ull, batch_cap_bases(a, lanes)); rd.repeat_rich = a->mi->n_occ > 0 && a->mi->occ[a->mi->n_occ - 1] >= 64;
	if(!rd.start()) { fprintf(stderr, "[minialign_amd] reader: no stream / staging memory\n"); return 1; }
	bool err = false;
	{
		/* K4 (the CIGAR strings on the device) for streams of two batches per lane and more: E.coli-size x100 (0.46 Gb, a batch and a half per lane) maps at 2.25 G bases/s with
		 * the strings made by the host and at 1.93 with K4, whose walk ends every batch; from the dm6-size x20 set on (2.9 Gb) it is hidden behind the other lanes' work */
		const uint64_t bb = getenv("MM_BATCH_BASES") ? (uint64_t)atoll(getenv("MM_BATCH_BASES")) : 300000000ull;
		const bool small = src->n < 2ull * (uint64_t)lanes * rd.dctx.size() * bb;
		each_context(a, [small](mm_align_t *q) { q->k4_off = small; });
	}
	int rc = stream_map(a, MM_OPEN_ENDED, [&](int di) { return rd.take(di, &err); }, [](mm_batch_t *h) { mm_batch_free(h); }, sink, lanes, pos_fd, pos_at);
	each_context(a, [](mm_align_t *q) { q->k4_off = false; });
	{ std::lock_guard<std::mutex> lk(rd.mu); if(err || rd.failed) rc = 1; }
	for(ReaderDev *R : rd.rdev) { a->st.text_bytes += R->bytes_up; a->st.reader_ms += R->t_io; }
	return rc;
}
/* standard input can be read once: its text is kept for the case that it is mapped onto several indices (-X, index files with several blocks) */
static std::shared_ptr<TextSrc> open_text_once(const char *fn)
{
	static std::shared_ptr<TextSrc> from_stdin;
	if(strcmp(fn, "-") != 0) return open_text(fn);
	if(!from_stdin) from_stdin = open_text(fn);
	return from_stdin;
}
/* the streaming engine over a text in host memory / over a file (mapped): records found and packed on the device, text to the sink in input order.  0 on success */
extern "C" int mm_map_text(mm_align_t *a, char const *text, uint64_t len, int lanes, mm_sam_sink_t sink, void *opaque)
{
	auto src = std::make_shared<TextSrc>(); src->p = text; src->n = len;
	for(int i = 0; i < 4 && src->first < src->n; i++) { if(text[src->first] == '>' || text[src->first] == '@') { src->delim = text[src->first]; break; } src->first++; }
	if(!src->delim) { if(len == 0) return 0; fprintf(stderr, "[minialign_amd] mm_map_text: neither FASTA nor FASTQ\n"); return 1; }
	return align_text(a, src, [&](uint32_t k, std::vector<std::string> &piece) { for(auto &x : piece) { if(sink && sink(opaque, k, x.data(), x.size())) return false; } return true; }, lanes);
}
extern "C" int mm_map_file(mm_align_t *a, char const *fn, int lanes, mm_sam_sink_t sink, void *opaque)
{
	std::shared_ptr<TextSrc> src = open_text_once(fn);
	if(!src) { fprintf(stderr, "[minialign_amd] cannot read `%s'\n", fn); return 1; }
	return align_text(a, src, [&](uint32_t k, std::vector<std::string> &piece) { for(auto &x : piece) { if(sink && sink(opaque, k, x.data(), x.size())) return false; } return true; }, lanes);
}
/* test entry for the device reader: the records of a file as K0r finds them and K0 packs them, nothing mapped -- names / comments / qualities read off the text at the
 * offsets the scan gave, base codes brought back from the packed arena in HBM.  *host_scanned = records that went through the host's sequential FASTQ reader.  NULL when
 * the file cannot be read or is rejected (a FASTQ record that does not start with '@': the reference gives up on the run). */
extern "C" mm_reads_t *mm_reads_scan(mm_align_t *a, char const *fn, int keep_qual, int keep_comment, uint64_t *host_scanned)
{
	std::shared_ptr<TextSrc> src = open_text(fn);
	if(!src) return NULL;
	TextReader rd; rd.dctx.push_back(a); rd.src = src; rd.min_len = 1; rd.keep_qual = keep_qual != 0; rd.lanes = 1;
	if(!rd.start()) return NULL;
	mm_reads_t *out = new mm_reads_s(); bool err = false;
	for(uint32_t k = 0; ; k++) {
		mm_batch_t *h = rd.take(0, &err);
		if(!h) break;
		Batch &b = h->b;
		if(!batch_upload(a, b)) { err = true; delete h; break; }
		std::vector<uint32_t> pk((b.total + 64) / 16 + 8), nm((b.total + 64) / 32 + 8);
		if(hipMemcpy(pk.data(), a->q_pk.p, ((b.total + 64) / 16) * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(nm.data(), a->q_nm.p, ((b.total + 64) / 32) * 4, hipMemcpyDeviceToHost) != hipSuccess) { err = true; delete h; break; }
		for(uint32_t i = 0; i < b.n; i++) {
			out->r.emplace_back(); HSeq &q = out->r.back();
			materialize(b, i, q, keep_qual != 0, keep_comment != 0);
			for(uint32_t j = 0; j < b.lens[i]; j++) { const uint64_t p = b.qoff[i] + j; q.seq[j] = ((nm[p >> 5] >> (p & 31)) & 1) ? 4 : (uint8_t)((pk[p >> 4] >> (2 * (p & 15))) & 3); }          /* the device's codes, not the host's */
			out->bases += q.seq.size();
		}
		delete h;
	}
	{ std::lock_guard<std::mutex> lk(rd.mu); if(rd.failed) err = true; }
	if(host_scanned) *host_scanned = rd.n_host_scanned;
	if(err) { delete out; return NULL; }
	return out;
}
extern "C" uint32_t mm_reads_codes(mm_reads_t const *r, uint32_t i, uint8_t *out, uint32_t cap)
{
	if(i >= r->r.size()) return 0;
	const uint32_t n = (uint32_t)r->r[i].seq.size(); if(out) memcpy(out, r->r[i].seq.data(), std::min(n, cap));
	return n;
}
extern "C" char const *mm_reads_qual(mm_reads_t const *r, uint32_t i) { return i < r->r.size() ? r->r[i].qual.c_str() : NULL; }
extern "C" char const *mm_reads_comment(mm_reads_t const *r, uint32_t i) { return (i < r->r.size() && r->r[i].has_comment) ? r->r[i].comment.c_str() : NULL; }
extern "C" mm_batch_t *mm_batch_pack(mm_reads_t const *r, uint32_t first, uint32_t n)
{
	mm_batch_t *h = new mm_batch_s();
	batch_fill(h, r, first, (uint32_t)std::min<uint64_t>((uint64_t)first + n, r->r.size()));
	batch_pack(h->b);
	return h;
}
extern "C" uint32_t mm_batch_reads(mm_batch_t const *h) { return h->b.n; }
/* test entry for the packing on the device (K0): every read of a set loaded with mm_reads_load_text as one batch, packed once on the host (pack_bases over the
 * parser's base codes) and once on the device from the text of the file; returns how many words of the two arenas (2-bit words and N-mask words) differ, -1 when
 * the set has no text or something failed */
extern "C" int64_t mm_pack_check(mm_align_t *a, mm_reads_t const *r)
{
	if(r->text.empty() || r->r.empty()) return -1;
	mm_batch_t *h = new mm_batch_s(); batch_fill(h, r, 0, (uint32_t)r->r.size());
	Batch &b = h->b;
	batch_pack(b, true);
	std::vector<uint32_t> pk(b.pk), nm(b.nm);
	b.packed = false; batch_pack(b, false);
	int64_t rc = -1;
	if(b.text && batch_upload(a, b)) {
		std::vector<uint32_t> dpk(pk.size(), 0), dnm(nm.size(), 0);
		const size_t npk = std::min<size_t>(pk.size(), (b.total + 64) / 16), nnm = std::min<size_t>(nm.size(), (b.total + 64) / 32);
		if(hipMemcpy(dpk.data(), a->q_pk.p, npk * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(dnm.data(), a->q_nm.p, nnm * 4, hipMemcpyDeviceToHost) == hipSuccess) {
			rc = 0; for(size_t i = 0; i < npk; i++) rc += dpk[i] != pk[i]; for(size_t i = 0; i < nnm; i++) rc += dnm[i] != nm[i];
		}
	}
	delete h;
	return rc;
}
/* test entry: the reads of a set loaded with mm_reads_load_text packed on the device from the text (K0) as one batch, then the arena brought back and unpacked: codes gets
 * one byte per base (0..3, 4 = N) read after read, lens the number of bases the device found per read.  Returns the number of reads, -1 on failure (no text, codes too small).
 * Nothing of the host parser's base codes enters: a checker compares with its own reader's (tests/test_pack_gpu.py: the oracle's). */
extern "C" int64_t mm_pack_fetch(mm_align_t *a, mm_reads_t const *r, uint8_t *codes, uint64_t cap, uint32_t *lens, uint32_t max_reads)
{
	if(r->text.empty() || r->r.empty() || r->r.size() > max_reads) return -1;
	mm_batch_t *h = new mm_batch_s(); batch_fill(h, r, 0, (uint32_t)r->r.size());
	Batch &b = h->b;
	batch_pack(b, false);
	int64_t rc = -1;
	if(b.text && batch_upload(a, b)) {
		std::vector<uint32_t> pk((b.total + 64) / 16 + 8), nm((b.total + 64) / 32 + 8), tn(b.n);
		if(hipMemcpy(pk.data(), a->q_pk.p, ((b.total + 64) / 16) * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(nm.data(), a->q_nm.p, ((b.total + 64) / 32) * 4, hipMemcpyDeviceToHost) == hipSuccess
			&& hipMemcpy(tn.data(), a->d_tn.p, (size_t)b.n * 4, hipMemcpyDeviceToHost) == hipSuccess) {
			uint64_t at = 0; rc = b.n;
			for(uint32_t i = 0; i < b.n && rc >= 0; i++) {
				lens[i] = tn[i];
				if(at + tn[i] > cap || tn[i] > b.lens[i]) { rc = -1; break; }          /* (the arena holds what the parser announced: batch_upload has compared the counts) */
				for(uint32_t j = 0; j < tn[i]; j++) { const uint64_t p = b.qoff[i] + j; codes[at + j] = ((nm[p >> 5] >> (p & 31)) & 1) ? 4 : (uint8_t)((pk[p >> 4] >> (2 * (p & 15))) & 3); }
				at += tn[i];
			}
		}
	}
	delete h;
	return rc;
}
/* every batch of reads [first, first + n) packed ahead of time, with the boundaries the streaming entries use; returns the count (at most max are made) */
extern "C" uint32_t mm_batch_pack_all(mm_reads_t const *r, uint32_t first, uint32_t n, mm_batch_t **out, uint32_t max)
{
	const auto sp = batch_spans(r, first, n);
	for(size_t k = 0; k < sp.size() && k < max; k++) { out[k] = mm_batch_pack(r, sp[k].first, sp[k].second - sp[k].first); }
	return (uint32_t)sp.size();
}
/*
 * A read set split over several contexts (one process per GPU, minialign_amd/multi.py): each part runs ahead with a guess for the carried reference length
 * (minialign.c:3864) at its start; once the part in front has finished, mm_carry_check tells what the true value changes, from the head of the stream the
 * context mapped last: 0 = nothing (every read made the same `apos >= rlen` decision and the chain of values has met the old one again), 1 = read
 * *first_affected decides differently (re-map from there), 2 = the recorded head (4 096 reads) ran out before the chains met (re-map the part).
 * mm_carry_after(i) is the value the chain had behind read i of that stream (what a re-mapped window must reproduce at its end to be spliced in).
 */
extern "C" int mm_carry_check(mm_align_t const *a, uint32_t truth, uint32_t *first_affected)
{
	uint32_t cur = truth, old = a->head_carry_in;
	if(first_affected) *first_affected = 0;
	if(cur == old) return 0;
	for(size_t i = 0; i < a->head.size(); i++) {
		const mm_align_s::HeadRec &h = a->head[i];
		/* the read ran with h.used (the old chain value, predicted and verified); with `cur` in its place the one test that reads it is apos0 >= rlen */
		if(h.apos0 != gaba::NIL && !h.cond0 && ((h.apos0 >= h.used) != (h.apos0 >= cur))) { if(first_affected) *first_affected = (uint32_t)i; return 1; }
		if(h.rid_last != gaba::NIL) return 0;          /* both chains continue from the length of this reference */
	}
	return 2;
}
extern "C" uint32_t mm_carry_after(mm_align_t const *a, uint32_t i)
{
	if(i >= a->head.size()) return 0xffffffffu;          /* beyond the recorded head: unknown */
	uint32_t cur = a->head_carry_in;
	for(size_t j = 0; j <= i && j < a->head.size(); j++) { if(a->head[j].rid_last != gaba::NIL) cur = a->mi->seq[a->head[j].rid_last].blen(); }
	return cur;
}
/* byte offset, in the text the last stream handed to its sink, of the first record of read i (i = the number of reads of a stream shorter than the recorded
 * head: the end of the text); UINT64_MAX beyond what was recorded (the first 4 096 reads, fewer when a batch had to be split) */
/* ... and, for a stream over a text, where the record of read i starts in that text (i = the number of reads of a stream shorter than the head: the length of the text) */
extern "C" uint64_t mm_head_text_offset(mm_align_t const *a, uint32_t i) { return i < a->head_txt.size() ? a->head_txt[i] : (i == a->head_txt.size() && i == a->head.size() && i < 4096 && !a->head_off_closed ? a->head_txt_end : ~0ull); }
extern "C" uint32_t mm_head_count(mm_align_t const *a) { return (uint32_t)a->head.size(); }          /* reads of the last stream that were recorded (at most 4 096) */
extern "C" uint64_t mm_head_offset(mm_align_t const *a, uint32_t i) { return i < a->head_off.size() ? a->head_off[i] : ~0ull; }
extern "C" int mm_map_packed(mm_align_t *a, mm_batch_t *const *batches, uint32_t n_batches, int lanes, mm_sam_sink_t sink, void *opaque)
{
	uint32_t mx = 0; for(uint32_t k = 0; k < n_batches; k++) mx = std::max(mx, batches[k]->b.max_qlen);
	each_context(a, [mx](mm_align_t *ln) { ln->qlen_hint = std::max(ln->qlen_hint, mx); });
	return stream_map(a, n_batches, counted_source(n_batches, [&](uint32_t k) { return batches[k]; }), [](mm_batch_t *) {},
		[&](uint32_t k, std::vector<std::string> &piece) { for(auto &x : piece) { if(sink && sink(opaque, k, x.data(), x.size())) return false; } return true; }, lanes > 0 ? lanes : default_lanes());
}
extern "C" int mm_map_reads(mm_align_t *a, mm_reads_t const *reads, uint32_t first, uint32_t n, int lanes, mm_sam_sink_t sink, void *opaque)
{
	const auto sp = batch_spans(reads, first, n);
	uint32_t mx = 0; for(auto &q : sp) for(uint32_t i = q.first; i < q.second; i++) mx = std::max<uint32_t>(mx, (uint32_t)reads->r[i].seq.size());
	each_context(a, [mx](mm_align_t *ln) { ln->qlen_hint = std::max(ln->qlen_hint, mx); });
	return stream_map(a, (uint32_t)sp.size(),
		counted_source((uint32_t)sp.size(), [&](uint32_t k) { mm_batch_t *h = new mm_batch_s(); batch_fill(h, reads, sp[k].first, sp[k].second); return h; }),
		[](mm_batch_t *h) { mm_batch_free(h); },
		[&](uint32_t k, std::vector<std::string> &piece) { for(auto &x : piece) { if(sink && sink(opaque, k, x.data(), x.size())) return false; } return true; }, lanes > 0 ? lanes : default_lanes());
}

namespace {
bool ends_with(const char *s, const char *suf) { size_t n = strlen(s), m = strlen(suf); return n >= m && strcmp(s + n - m, suf) == 0; }
/* minialign -d idx.mai ref.fa [ref2.fa ...]: one index block per reference file, no mapping (main_index, minialign.c:6293-6345) */
int main_index(mm_opt_t *o, const char *const *files, int nf, double t0)
{
	FILE *fp = fopen(o->fnw.c_str(), "wb");
	if(!fp) { fprintf(stderr, "[E::main_index] failed to open index file `%s' in write mode. Please check file path and its permission.\n", o->fnw.c_str()); return 1; }
	int rc = 0;
	for(int i = 0; i < nf && rc == 0; i++) {
		mm_idx_t *mi = mm_idx_gen(o, files[i]);
		if(!mi) { fprintf(stderr, "[E::main_index] failed to build index for `%s'. Please check file path and format.\n", files[i]); rc = 1; break; }
		rc = mm_idx_dump(mi, fp);
		if(rc) fprintf(stderr, "[E::main_index] failed to write the index to `%s'.\n", o->fnw.c_str());
		else fprintf(stderr, "[M::main_index::%.3f] built and dumped index for %u target sequence(s).\n", (now_ms() - t0) * 1e-3, mm_idx_n_seq(mi));
		mm_idx_destroy(mi);
	}
	fclose(fp);
	return rc;
}
}
extern "C" int mm_main(int argc, char **argv)
{
	setenv("GPU_MAX_HW_QUEUES", "16", 0);          /* lanes and side streams should not share hardware queues (read when the HIP runtime starts) */
	mm_opt_t *o = mm_opt_init();
	const char *files[64]; int nf = 0;
	const int prc = mm_opt_parse(o, argc, (char const *const *)argv, files, 63, &nf);
	if(prc || nf < 1 || o->help) {
		/* -h: the text goes to stdout (minialign.c:6466-6470); the exit status is 1 either way -- the reference sets 0 for -h and then overwrites it with the
		 * return value of mm_print_help, which is 1 once the text is printed (minialign.c:6469, 6301) */
		FILE *hf = (!prc && o->help) ? stdout : stderr;
		fprintf(hf, "usage: minialign [-x preset] [options] [-d idx.mai] ref.{fa,fa.gz,mai} reads.{fa,fq}[.gz] ... > out.sam\n"
			"  presets    -x pacbio[.clr|.ccs] | ont[.r7|.r9[.4|.5[.1]]][.1d|.1dsq|.2d] | ava\n"
			"  index      -k INT  -w INT  -B INT  -f FLOAT,...  -c [NAME,...]  -L INT  -d FILE\n"
			"  scores     -a INT  -b INT  -e XYn,...  -p INT  -q INT  -r INT[,INT]  -Y INT\n"
			"  mapping    -s INT  -m FLOAT  -W INT  -G INT  -X  -A  -C [INT,INT]\n"
			"  output     -O sam|maf|blast6|paf  -T TAG,...  -R '@RG\\tID:...'  -Q  -P\n"
			"  accepted   -t INT  -v [INT]  -1 INT  -2 INT  -h\n");
		const int ret = 1;
		mm_opt_destroy(o); return ret;
	}
	double t0 = now_ms();
	if(!o->fnw.empty()) { int rc = main_index(o, files, nf, t0); mm_opt_destroy(o); return rc; }
	/* a prebuilt index (file name ending in .mai) may hold several blocks: every query file is mapped onto each in turn, with a header per block;
	 * -X without a prebuilt index maps every file onto every file, one index per file (minialign.c:6373-6379, 6413-6436) */
	const bool prebuilt = ends_with(files[0], ".mai");
	const bool ava = o->ava && !prebuilt;           /* the mapper's own flag word (a.flag); -R sets the same bit in the printer's only */
	const int n_ref = ava ? nf : 1, qh = ava ? 0 : 1;
	if(qh == nf) { fprintf(stderr, "[M::main_align] query-side input redirected to stdin.\n"); files[nf++] = "-"; }     /* minialign.c:6380-6384 */
	/* the first query file is parsed on a thread of its own while the index is built or loaded */
	mm_reads_t *first_reads = NULL;
	std::thread rt([]() {});
	std::thread hw([]() { int n = 0; if(hipGetDeviceCount(&n) == hipSuccess && n > 0) { (void)hipFree(0); } });      /* bring the HIP runtime up meanwhile */
	const bool keep_first = prebuilt || n_ref > 1;      /* the parsed first query file serves every index */
	FILE *pg = prebuilt ? fopen(files[0], "rb") : NULL;
	int rc = 0, at_eof = 0; uint32_t micnt = 0;
	bool joined = false;
	auto join = [&]() { if(!joined) { hw.join(); rt.join(); joined = true; } };
	if(prebuilt && !pg) { fprintf(stderr, "[E::main_align] failed to build index for `%s'. Please check file path and format.\n", files[0]); rc = 1; }
	mm_stats_t tot; memset(&tot, 0, sizeof(tot));
	while(rc == 0) {
		mm_idx_t *mi = prebuilt ? mm_idx_load(pg, &at_eof) : ((int)micnt < n_ref ? mm_idx_gen(o, files[micnt]) : NULL);
		if(!mi) {
			if(prebuilt && (micnt == 0 || !at_eof)) { fprintf(stderr, "[E::main_align] failed to load index block from `%s'. Please check file path and version, or rebuild the index.\n", files[0]); rc = 1; }
			else if(!prebuilt && (int)micnt < n_ref) { fprintf(stderr, "[E::main_align] failed to build index for `%s'. Please check file path and format.\n", files[micnt]); rc = 1; }
			break;
		}
		const bool vb = getenv("MM_VERBOSE") != NULL; double tq = now_ms();
		auto lapm = [&](const char *w) { if(vb) { double t = now_ms(); fprintf(stderr, "[minialign_amd] main: %s %.1f ms (at %.3f s)\n", w, t - tq, (t - t0) * 1e-3); tq = t; } };
		lapm("index");
		if(!joined) hw.join();
		lapm("wait for the HIP runtime");
		mm_align_t *a = mm_align_init(o, mi);
		lapm("device context");
		if(!joined) { rt.join(); joined = true; }
		lapm("wait for the read parser");
		if(!a) { fprintf(stderr, "[E::main_align] failed to instanciate alignment context.\n"); mm_idx_destroy(mi); rc = 1; break; }
		fprintf(stderr, "[M::main_align::%.3f] loaded/built index for %u target sequence(s).\n", (now_ms() - t0) * 1e-3, mm_idx_n_seq(mi));
		if(o->format == 0) mm_print_sam_header(a, stdout, o->arg_line.c_str());        /* only SAM has a header (minialign.c:5666-5671) */
		for(int i = qh; i < nf && rc == 0; i++) {
			if(i == qh && first_reads) { rc = align_reads(a, first_reads, stdout, keep_first); if(!keep_first) first_reads = NULL; }
			else { rc = mm_align_file(a, files[i], stdout); }
			if(rc) fprintf(stderr, "[E::main_align] failed to map sequence file `%s'. Please check file path and format.\n", files[i]);
			else fprintf(stderr, "[M::main_align::%.3f] finished mapping `%s' onto `%s'.\n", (now_ms() - t0) * 1e-3, files[i], files[prebuilt ? 0 : micnt]);
		}
		mm_stats_t st; mm_stats(a, &st, 0);
		tot.reads += st.reads; tot.bases += st.bases; tot.k1_ms += st.k1_ms; tot.k2_ms += st.k2_ms; tot.k3_ms += st.k3_ms; tot.host_post_ms += st.host_post_ms; tot.host_sam_ms += st.host_sam_ms; tot.reruns += st.reruns;
		mm_align_destroy(a); mm_idx_destroy(mi); micnt++;
	}
	join();
	if(first_reads) mm_reads_free(first_reads);
	if(pg) fclose(pg);
	if(rc == 0) fprintf(stderr, "[M::main] %lu reads, %lu bases; kernels: sketch %.1f ms, sort+chain %.1f ms, extend %.1f ms; host post-map %.1f ms, SAM %.1f ms; %lu re-run(s)\n",
		(unsigned long)tot.reads, (unsigned long)tot.bases, tot.k1_ms, tot.k2_ms, tot.k3_ms, tot.host_post_ms, tot.host_sam_ms, (unsigned long)tot.reruns);
	mm_opt_destroy(o);
	return rc;
}
