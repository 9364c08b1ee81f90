/* post-map (prune, supplementary, MAPQ: minialign.c:4185-4398), the reverse CIGAR walk (gaba_parse.h:168-221) and the printers: SAM with its tags, MAF, BLAST6, PAF (minialign.c:5096-5625) -- part of mm_host.hip (included from there at the place it stood; split out in round 6 so that it can be read on its own) */
/* ---- post-map on the host (minialign.c:4185-4398) ---- */
struct OutAln { uint32_t aln; uint32_t mapq; };
struct OutReg { uint32_t n_all = 0, n_uniq = 0; std::vector<OutAln> aln; bool mapped = false; };
inline uint32_t clip_mapq(double x) { uint32_t v = h_d2u32(x); return std::min<uint32_t>(v, 60 * 16); }

void post_map(const mm_align_t *a, const ReadState &rs, Root *res, uint64_t *bin, const AlnRec *alns, OutReg &out)
{
	uint32_t n_res = rs.n_res;
	out.mapped = n_res > 0;
	if(!n_res) return;
	sort_res((ResEnt *)res, n_res);                                       /* radix_sort_64x, minialign.c:4452 */
	/* mm_prune_regs */
	uint64_t q = n_res;
	uint32_t minv = (uint32_t)h_ofs((int32_t)h_f2u32((float)h_ofs((int32_t)res[0].plen) * a->o.min_ratio));
	while(res[--q].plen > minv) {}
	n_res = (uint32_t)(q + 1);
	uint32_t n_all = n_res;
	auto hdr = [&](uint32_t iid) { return (uint32_t *)&bin[iid]; };        /* { n_aln, plen, lb, ub } */
	/* mm_collect_supp */
	uint64_t p, qq;
	for(p = 1, qq = n_res; p < qq; p++) {
		uint64_t mx = 0;
		for(uint64_t i = p; i < qq; i++) {
			uint32_t *s = hdr(res[i].lid);
			int64_t lb = s[2], ub = s[3], span = ub - lb; bool covered = false;
			for(uint64_t j = 0; j < p; j++) {
				uint32_t *t = hdr(res[j].lid);
				if((int64_t)t[3] < ub) lb = std::max<int64_t>(lb, t[3]); else ub = std::min<int64_t>(ub, t[2]);
				if(1.2 * (double)(ub - lb) < (double)span) { qq--; std::swap(res[i], res[qq]); i--; covered = true; break; }
			}
			if(covered) continue;
			mx = std::max<uint64_t>(mx, ((uint64_t)(2 * (ub - lb) - span) << 32) | i);
		}
		if(mx & 0xffffffff) std::swap(res[p], res[mx & 0xffffffff]);
	}
	p = std::min(p, qq);
	/* mm_post_map */
	int64_t usc = 0, lsc = INT64_MAX, tsc = 0;
	for(uint64_t i = p; i < n_res; i++) { int64_t sc = h_ofs((int32_t)res[i].plen); usc = std::max(usc, sc); lsc = std::min(lsc, sc); tsc += sc; }
	lsc = (lsc == INT32_MAX) ? 0 : lsc;
	double tpc = 1.0, x = a->xcoef, mxc = a->mcoef + a->xcoef;
	for(uint64_t i = 0; i < p; i++) {
		uint32_t score = (uint32_t)h_ofs((int32_t)res[i].plen);
		uint32_t *b = hdr(res[i].lid);
		double pid = 0.0; uint64_t len = 0;
		for(uint32_t j = 0; j < b[0]; j++) { const AlnRec &al = alns[bin[res[i].lid + 2 + j] - 1]; len += al.plen; pid += (double)al.plen * al.identity; }
		pid /= (double)len;
		double ec = 2.0 / (pid * mxc - x);
		double ulen = ec * (double)std::max<int64_t>((int64_t)score - usc, 0), pe = 1.0 / (ulen * ulen + 1);
		b[1] = clip_mapq(-10.0 * 16 * log10(pe));
		tpc *= 1.0 - pe;
	}
	double tpe = std::min(1.0 - tpc, 1.0);
	for(uint64_t i = p; i < n_res; i++) {
		uint32_t *b = hdr(res[i].lid);
		b[1] = clip_mapq(-10.0 * 16 * log10(1.0 - tpe * (double)(int64_t)((int64_t)res[i].plen - lsc + 1) / (double)tsc));
	}
	/* mm_pack_reg */
	for(uint64_t i = 0; i < n_all; i++) {
		uint32_t *b = hdr(res[i].lid);
		for(uint32_t j = 0; j < b[0]; j++) out.aln.push_back(OutAln{ (uint32_t)(bin[res[i].lid + 2 + j] - 1), b[1] });
		if(i == p - 1) out.n_uniq = (uint32_t)out.aln.size();
	}
	out.n_all = (uint32_t)out.aln.size();
}

/* ---- CIGAR from path bits (gaba_parse.h:168-221, reverse parser) ---- */
inline uint64_t path_u64(const uint64_t *ptr, int64_t pos) { int64_t rem = pos & 63; return (ptr[pos >> 6] >> rem) | ((ptr[(pos >> 6) + 1] << (63 - rem)) << 1); }
inline uint64_t lzc(uint64_t x) { return x ? (uint64_t)__builtin_clzll(x) : 64; }
inline void put_num(std::string &s, uint64_t v) { char b[24]; int n = 0; if(!v) b[n++] = '0'; while(v) { b[n++] = (char)('0' + v % 10); v /= 10; } while(n) s.push_back(b[--n]); }
/* the same into a raw buffer (most run lengths have one or two digits) */
inline char *put_num_p(char *p, uint64_t v)
{
	if(v < 10) { *p++ = (char)('0' + v); return p; }
	if(v < 100) { *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); return p; }
	char b[24]; int n = 0; while(v) { b[n++] = (char)('0' + v % 10); v /= 10; } while(n) *p++ = b[--n];
	return p;
}
void cigar_reverse(std::string &out, const uint32_t *path, uint64_t offset, uint64_t len)
{
	const uint64_t *p = (const uint64_t *)((uintptr_t)path & ~(uintptr_t)7);
	uint64_t ofs = (uint64_t)((int64_t)offset + (((uintptr_t)path & 4) ? 32 : 0) - 64), idx = len;
	/* written through a raw pointer into room reserved for the worst case (every path bit its own run: 2 characters per bit and change) */
	const size_t o = out.size(); out.resize(o + 2 * len + 64);
	char *w = &out[o];
	while((int64_t)idx > 0) {
		uint64_t m = lzc(path_u64(p, (int64_t)(ofs + idx))), c = std::min(idx, m - (m > 0));
		idx -= c; if(c) { w = put_num_p(w, c); *w++ = 'D'; }
		m = lzc(~path_u64(p, (int64_t)(ofs + idx))); c = std::min(idx, m);
		idx -= c; if(c) { w = put_num_p(w, c); *w++ = 'I'; }
		uint64_t sidx = idx;
		do { m = lzc(path_u64(p, (int64_t)(ofs + idx)) ^ 0x5555555555555555ull); c = std::min(idx, m) & ~1ull; idx -= c; } while(c == 64);
		if((sidx - idx) >> 1) { w = put_num_p(w, (sidx - idx) >> 1); *w++ = 'M'; }
	}
	out.resize((size_t)(w - out.data()));
}

/* ---- SAM (minialign.c:5127-5198, 5390-5426), default tag set ---- */
void sam_seq(std::string &s, const uint8_t *q, uint32_t n, bool rev)
{
	static const char fw[] = "ACGTN\0\0\0\0\0\0\0\0\0\0\0", rv[] = "TGCAN\0\0\0\0\0\0\0\0\0\0\0";
	size_t o = s.size(); s.resize(o + n);
	char *d = &s[o];
	if(!rev) for(uint32_t i = 0; i < n; i++) d[i] = fw[q[i] & 15];
	else { const uint8_t *e = q + n - 1; for(uint32_t i = 0; i < n; i++) d[i] = rv[e[-(int64_t)i] & 15]; }
}
/* walks the path bits in the order of _parser_loop_rv (gaba_parse.h:168-188); fn(op, count) sees every nonzero run ('D', 'I', 'M') */
template<typename F> void path_walk_reverse(const uint32_t *path, uint64_t offset, uint64_t len, F fn)
{
	const uint64_t *p = (const uint64_t *)((uintptr_t)path & ~(uintptr_t)7);
	uint64_t ofs = (uint64_t)((int64_t)offset + (((uintptr_t)path & 4) ? 32 : 0) - 64), idx = len;
	while((int64_t)idx > 0) {
		uint64_t m = lzc(path_u64(p, (int64_t)(ofs + idx))), c = std::min(idx, m - (m > 0));
		idx -= c; if(c) fn('D', c);
		m = lzc(~path_u64(p, (int64_t)(ofs + idx))); c = std::min(idx, m);
		idx -= c; if(c) fn('I', c);
		uint64_t sidx = idx;
		do { m = lzc(path_u64(p, (int64_t)(ofs + idx)) ^ 0x5555555555555555ull); c = std::min(idx, m) & ~1ull; idx -= c; } while(c == 64);
		if((sidx - idx) >> 1) fn('M', (sidx - idx) >> 1);
	}
}
/* MD:Z (mm_print_sam_md, minialign.c:5243-5301): match counts, the reference base at a mismatch, ^ + reference bases at a deletion.  On the
 * reverse strand the query base is complemented by xor 3, so an N there never equals the reference's N. */
void sam_md(std::string &s, const std::vector<uint8_t> &rcodes, const uint8_t *qseq, uint32_t qlen, const gaba::Segment &sg, const uint32_t *path)
{
	static const char dec[] = "ACGTN\0\0\0\0\0\0\0\0\0\0\0";
	s += "\tMD:Z:";
	const bool rev = (~sg.bid & 1) != 0;
	const uint8_t *rp = rcodes.data() + (rcodes.size() - sg.apos - sg.alen), *rb = rp;
	const uint8_t *qp = rev ? qseq + (qlen - sg.bpos) : qseq + (qlen - sg.bpos - sg.blen);
	path_walk_reverse(path, sg.ppos, (uint64_t)sg.alen + sg.blen, [&](char op, uint64_t c) {
		if(op == 'D') { put_num(s, (uint64_t)(rp - rb)); s.push_back('^'); rb = rp + c; for(uint64_t i = 0; i < c; i++) s.push_back(dec[*rp++ & 15]); }
		else if(op == 'I') { if(rev) qp -= c; else qp += c; }
		else {
			for(uint64_t t = 0; t < c; t++) {
				const uint8_t rc = rp[t], qc = rev ? (uint8_t)(3 ^ qp[-1 - (int64_t)t]) : qp[t];
				if(rc != qc) { put_num(s, (uint64_t)(rp + t - rb)); s.push_back(dec[rc & 15]); rb = rp + t + 1; }
			}
			rp += c; if(rev) qp -= c; else qp += c;
		}
	});
	put_num(s, (uint64_t)(rp - rb));
}
inline void put_int(std::string &s, int64_t v) { if(v < 0) { s.push_back('-'); put_num(s, (uint64_t)-v); } else put_num(s, (uint64_t)v); }
/* mm_print_sam_mapped with its tag printers (minialign.c:5127-5426).  QUIRKS kept: flags and tag bits share one word (-P switches IH on, -T IH omits
 * secondaries); the SA entries name the first reference sequence whatever they hit and carry the raw 16x fixed-point mapping quality; RG:Z prints
 * the whole "ID:..." token. */
void sam_record(const mm_align_t *a, std::string &s, const char *qname, const uint8_t *qseq, uint32_t qlen, const OutReg &reg,
	const AlnRec *alns, const gaba::Segment *segs, const uint32_t *paths, const HSeq *rec, const CigEnt *cig_ent = nullptr, const char *cig_text = nullptr)
{
	/* the run lengths of a segment: the string the device made for its slot (K4, mm_cigar.hpp) or, without one, the walk over the path words (gaba_parse.h:168-221) */
	auto put_cigar = [&](const AlnRec &al, uint32_t slot, const gaba::Segment &g) {
		if(cig_ent) { s.append(cig_text + cig_ent[slot].off, cig_ent[slot].len); }
		else { cigar_reverse(s, paths + al.path_off, g.ppos, (uint64_t)g.alen + g.blen); }
	};
	const uint64_t f = a->o.ptags();
	auto tag = [f](int x) { return ((f >> x) & 1) != 0; };
	const bool has_qual = rec && !rec->qual.empty(), has_co = rec && rec->has_comment;
	if(!reg.mapped || reg.n_all == 0) {
		s += qname; s += "\t4\t*\t0\t0\t*\t*\t0\t0\t"; sam_seq(s, qseq, qlen, false); s.push_back('\t');
		if(has_qual) s.append(rec->qual, 0, qlen); else s.push_back('*');
		if(has_co) { s += "\tCO:Z:"; s += rec->comment; }
		s.push_back('\n');
		return;
	}
	auto edit = [&](const AlnRec &al) { return (uint32_t)((double)al.dcnt * (1.0 - al.identity)) + al.agcnt + al.bgcnt; };
	uint32_t flag = 0;
	const uint32_t n = (f & 0x08) ? reg.n_uniq : reg.n_all;           /* MM_OMIT_REP */
	for(uint32_t i = 0; i < n; i++) {
		if(i >= reg.n_uniq) flag = 0x100;
		const AlnRec &al = alns[reg.aln[i].aln];
		for(uint32_t j = al.slen; j > 0; j--) {
			const gaba::Segment &sg = segs[al.seg_off + j - 1];
			const HSeq &r = a->mi->seq[sg.aid >> 1];
			uint32_t rs = r.blen() - sg.apos - sg.alen;
			uint32_t hl = qlen - sg.bpos - sg.blen, tl = sg.bpos;
			uint32_t qs = (flag & 0x900) ? hl : 0, qe = qlen - ((flag & 0x900) ? tl : 0);
			s += qname; s.push_back('\t'); put_num(s, flag | ((~sg.bid & 1) << 4)); s.push_back('\t');
			s += r.name; s.push_back('\t'); put_num(s, rs + 1); s.push_back('\t'); put_num(s, reg.aln[i].mapq >> 4); s.push_back('\t');
			char clip = (flag & 0x900) ? 'H' : 'S';
			if(hl) { put_num(s, hl); s.push_back(clip); }
			put_cigar(al, al.seg_off + j - 1, sg);
			if(tl) { put_num(s, tl); s.push_back(clip); }
			s += "\t*\t0\t0\t";
			if(sg.bid & 1) sam_seq(s, qseq + qs, qe - qs, false); else sam_seq(s, qseq + (qlen - qe), qe - qs, true);
			s.push_back('\t');
			if(has_qual) {
				if(sg.bid & 1) s.append(rec->qual, qs, qe - qs);
				else { const char *qq = rec->qual.data() + (qlen - qe); for(uint32_t x = qe - qs; x > 0; x--) s.push_back(qq[x - 1]); }
			} else s.push_back('*');
			if(f) {
				if(tag(0)) { s += "\tRG:Z:"; s += a->o.rg_id; }
				if(tag(2)) { s += "\tNH:i:"; put_num(s, reg.n_all); }
				if(tag(3)) { s += "\tIH:i:"; put_num(s, i); }
				if(tag(4)) { s += "\tAS:i:"; put_int(s, al.score); }
				if(tag(6)) { s += "\tNM:i:"; put_num(s, edit(al)); }
				if(tag(8)) sam_md(s, ref_codes(a->mi, sg.aid >> 1), qseq, qlen, sg, paths + al.path_off);
			}
			if(i == 0 && j == al.slen) {
				flag = 0x800;
				bool stop = false;
				if(tag(5)) { s += "\tXS:i:"; put_int(s, reg.n_all > 1 ? alns[reg.aln[1].aln].score : 0); }
				if(tag(7) && (reg.n_uniq > 1 || alns[reg.aln[0].aln].slen > 1)) {
					s += "\tSA:Z:";
					for(uint32_t x = 0; x < reg.n_uniq; x++) {
						const AlnRec &bl = alns[reg.aln[x].aln];
						for(uint32_t y = bl.slen; y > 0; y--) {
							if(x == 0 && y == bl.slen) continue;
							const gaba::Segment &sh = segs[bl.seg_off + y - 1];
							const HSeq &rr = a->mi->seq[sh.aid >> 1];
							s += a->mi->seq[0].name; s.push_back(','); put_num(s, rr.blen() - sh.apos - sh.alen + 1); s.push_back(',');
							s.push_back((sh.bid & 1) ? '+' : '-'); s.push_back(',');
							uint32_t h2 = qlen - sh.bpos - sh.blen, t2 = sh.bpos;
							if(h2) { put_num(s, h2); s.push_back('H'); }
							put_cigar(bl, bl.seg_off + y - 1, sh);
							if(t2) { put_num(s, t2); s.push_back('H'); }
							s.push_back(','); put_num(s, reg.aln[x].mapq); s.push_back(','); put_num(s, edit(bl)); s.push_back(';');
						}
					}
					stop = true;
				}
				if(has_co) { s += "\tCO:Z:"; s += rec->comment; }
				if(stop) { s.push_back('\n'); return; }                 /* the other records are in the SA tag (minialign.c:5418-5420) */
			}
			s.push_back('\n');
		}
		flag = 0x800;
	}
}

/* ---- the other output formats: MAF, BLAST6 (tabular), PAF (minialign.c:5427-5625); nothing is printed for unmapped reads ---- */
void put_fixed(std::string &s, uint32_t n, int c)           /* _putfi, minialign.c:4812: n with a decimal point in front of its last c digits */
{
	char d[24]; int i = 0;
	while(n || i <= c) { d[i++] = (char)('0' + n % 10); n /= 10; }
	for(int j = i; j > c; j--) s.push_back(d[j - 1]);
	s.push_back('.');
	for(int j = c; j > 0; j--) s.push_back(d[j - 1]);
}
void put_pair(std::string &s1, std::string &s2, uint32_t n1, uint32_t n2)       /* _putpi, minialign.c:4847: two numbers right-aligned to one width */
{
	char d1[16], d2[16]; int i = 0;
	while(n1 | n2) { d1[i] = (char)(n1 % 10); d2[i] = (char)(n2 % 10); n1 /= 10; n2 /= 10; i++; }
	if(i == 0) { d1[0] = d2[0] = 0; i = 1; }
	int z1 = 0, z2 = 0;
	for(int j = i; j > 0; j--) {
		z1 |= d1[j - 1] | (j == 1); z2 |= d2[j - 1] | (j == 1);
		s1.push_back((char)(d1[j - 1] + '0' - (z1 ? 0 : 0x10))); s2.push_back((char)(d2[j - 1] + '0' - (z2 ? 0 : 0x10)));
	}
}
void alt_record(const mm_align_t *a, std::string &s, const char *qname, const uint8_t *qseq, uint32_t qlen, const OutReg &reg,
	const AlnRec *alns, const gaba::Segment *segs, const uint32_t *paths)
{
	if(!reg.mapped || reg.n_all == 0) return;
	const uint64_t f = a->o.ptags();
	const uint32_t n = (f & 0x08) ? reg.n_uniq : reg.n_all;
	const size_t l_qname = strlen(qname);
	std::vector<char> buf;
	for(uint32_t i = 0; i < n; i++) {
		const AlnRec &al = alns[reg.aln[i].aln];
		const gaba::Segment &sg = segs[al.seg_off + al.slen - 1], &eg = segs[al.seg_off];
		const HSeq &r = a->mi->seq[sg.aid >> 1]; const uint32_t rl = r.blen();
		const uint32_t dcnt = al.dcnt, mcnt = h_d2u32((double)dcnt * al.identity), gcnt = al.agcnt + al.bgcnt;
		if(a->o.format == 1) {                 /* mm_print_maf_mapped, :5476 */
			for(uint32_t j = al.slen; j > 0; j--) {
				const gaba::Segment &g = segs[al.seg_off + j - 1];
				const HSeq &rr = a->mi->seq[g.aid >> 1]; const uint32_t rrl = rr.blen();
				const uint32_t rs = rrl - g.apos - g.alen, qs = qlen - g.bpos - g.blen;
				const uint64_t plen = (uint64_t)g.alen + g.blen;
				s += "a score="; put_num(s, (uint32_t)al.score); s.push_back('\n');
				const size_t w = std::max(rr.name.size(), l_qname) + 1;
				std::string q2 = "s "; q2 += qname; q2.append(w - l_qname, ' ');
				s += "s "; s += rr.name; s.append(w - rr.name.size(), ' ');
				put_pair(s, q2, rs, qs); s.push_back(' '); q2.push_back(' ');
				put_pair(s, q2, g.alen, g.blen); s.push_back(' '); q2.push_back(' ');
				s += "+ "; q2.push_back((g.bid & 1) ? '+' : '-'); q2.push_back(' ');
				put_pair(s, q2, rrl, qlen); s.push_back(' '); q2.push_back(' ');
				buf.resize(plen + 64);
				uint64_t m = gaba_dump_seq_reverse(buf.data(), buf.size(), GABA_SEQ_A, paths + al.path_off, g.ppos, plen, ref_codes(a->mi, g.aid >> 1).data() + rs, '-');
				s.append(buf.data(), m); s.push_back('\n');
				s += q2;
				m = gaba_dump_seq_reverse(buf.data(), buf.size(), GABA_SEQ_B | ((g.bid & 1) ? GABA_SEQ_FW : GABA_SEQ_RV), paths + al.path_off, g.ppos, plen,
					(g.bid & 1) ? qseq + qs : qseq + (qlen - qs), '-');
				s.append(buf.data(), m); s += "\n\n";
			}
		} else if(a->o.format == 2) {          /* mm_print_blast6_mapped, :5497: qname rname idt len #x #gap qs qe rs re e-value bitscore */
			const uint32_t rs = (sg.bid & 1) ? rl - sg.apos - sg.alen + 1 : rl - eg.apos, re = (sg.bid & 1) ? rl - eg.apos : rl - sg.apos - sg.alen + 1;
			const uint32_t qs = qlen - sg.bpos - sg.blen + 1, qe = qlen - eg.bpos;
			s += qname; s.push_back('\t'); s += r.name; s.push_back('\t');
			put_fixed(s, h_d2u32(1000.0 * al.identity), 3);
			for(uint32_t v : { dcnt + gcnt, dcnt - mcnt, gcnt, qs, qe, rs, re }) { s.push_back('\t'); put_num(s, v); }
			s.push_back('\t');
			const double bit = 1.85 * (double)al.score - 0.02;
			put_fixed(s, h_d2u32(1000.0 * (double)rl * (double)qlen * pow(2.0, -bit)), 3);
			s.push_back('\t'); put_num(s, h_d2u32(bit)); s.push_back('\n');
		} else {                                /* mm_print_paf_mapped, :5549: qname ql qs qe strand rname rl rs re #match block_len mapq [tags] */
			const uint32_t rs = rl - sg.apos - sg.alen, re = rl - eg.apos, qs = qlen - sg.bpos - sg.blen, qe = qlen - eg.bpos;
			s += qname; for(uint32_t v : { qlen, qs, qe }) { s.push_back('\t'); put_num(s, v); }
			s.push_back('\t'); s.push_back((sg.bid & 1) ? '+' : '-'); s.push_back('\t'); s += r.name;
			for(uint32_t v : { rl, rs, re, mcnt, dcnt + gcnt, reg.aln[i].mapq >> 4 }) { s.push_back('\t'); put_num(s, v); }
			if((f >> 4) & 1) { s += "\tAS:i:"; put_num(s, (uint32_t)al.score); }
			if((f >> 10) & 1) { s += "\tID:f:"; put_fixed(s, h_d2u32(al.identity * 10000.0), 4); }
			if((f >> 6) & 1) { s += "\tNM:i:"; put_num(s, (dcnt - mcnt) + gcnt); }
			if((f >> 11) & 1) { s += "\tSQ:i:"; sam_seq(s, qseq, qlen, false); }
			if((f >> 9) & 1) { s += "\tCG:Z:"; cigar_reverse(s, paths + al.path_off, 0, al.plen); }
			s.push_back('\n');
		}
	}
}

bool ensure_shared_slabs(mm_align_t *P, uint32_t max_qlen, uint32_t lanes);
