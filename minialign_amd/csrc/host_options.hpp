/* the option table, presets and checks of the command line (minialign.c:6166-6203, 5703-6160): mm_opt_* -- part of mm_host.hip (included from there at the place it stood; split out in round 6 so that it can be read on its own) */
/* =============================================================================================
 * options
 * ============================================================================================= */
struct mm_opt_s {
	uint32_t k = 15, w = 32, b = 14, n_frq = 3; float frq[8] = { 0.05f, 0.01f, 0.001f, 0 };     /* up to MAX_FRQ_CNT = 7 thresholds, minialign.c:29 */
	uint32_t min_len = 1, help = 0;
	/* output (minialign.c:5880-5967): flag = -P 0x08 and bit 0 when -R is given; tags = bits 1 << MM_xx of -T.  The reference's printer ORs the two
	 * into one word (minialign.c:5677), so -P also switches IH on and -T IH also omits the secondary records: kept */
	uint64_t flag = 0, tags = 0; std::string rg_line, rg_id; bool keep_qual = false;
	uint32_t format = 0;             /* -O: 0 sam, 1 maf, 2 blast6, 5 paf (minialign.c:2543-2549, 5940) */
	bool ava = false;                /* -X (MM_AVA in the mapper's flag word, minialign.c:5965, 6377) */
	bool circ_set = false; std::vector<std::string> circ_names;     /* -c: given at all / names of the circular reference sequences (none: all), minialign.c:2457, 5986 */
	uint64_t ptags() const { return flag | tags; }
	uint32_t wlen = 7000, glen = 7000, min_score = 50; float min_ratio = 0.3f;
	gaba_params_t p;
	uint32_t nth = 1;
	std::string arg_line, fnw;       /* fnw: -d, file the index is dumped to (minialign.c:5979 mm_opt_fnw) */
	mm_opt_s() { memset(&p, 0, sizeof(p)); for(int i = 0; i < 16; i++) p.score_matrix[i] = (i & 3) == (i >> 2) ? 1 : -1; p.gi = 1; p.ge = 1; p.xdrop = 50; }
};
namespace {
int opt_one(mm_opt_t *o, char c, const char *arg);
int opt_line(mm_opt_t *o, const char *s)
{
	while(*s) {
		while(*s == ' ') s++;
		if(*s != '-') break;
		char c = s[1]; s += 2; std::string a; while(*s && *s != ' ') a.push_back(*s++);
		if(opt_one(o, c, a.c_str())) return 1;
	}
	return 0;
}
template<typename F> void split_each(const char *arg, const char *delims, F fn)          /* mm_split_foreach */
{
	int i = 0;
	for(const char *p = arg; ; ) { const char *e = p; while(*e && !strchr(delims, *e)) e++; if(e > p) fn(i++, std::string(p, e)); if(!*e) break; p = e + 1; }
}
/* the preset tree of minialign.c:5853-5878 as data: each name applies its option line, then the next name is looked up among its children
 * (mm_opt_preset, minialign.c:5880-5889); a name that is not there is an error (the reference then tries to read it as a configuration file) */
struct PresetNode { const char *key, *val; const PresetNode *kids; };
#define PN_END { nullptr, nullptr, nullptr }
const PresetNode pn_leaf_r7[] = { { "1d", "", nullptr }, { "2d", "", nullptr }, PN_END };
const PresetNode pn_leaf_1[] = { { "1d", "", nullptr }, { "1dsq", "-b6 -r4,4", nullptr }, { "2d", "-b6 -r4,4", nullptr }, PN_END };
const PresetNode pn_r9_45[] = { { "1", "", pn_leaf_1 }, { "1d", "", nullptr }, { "1dsq", "-b6 -r4,4", nullptr }, { "2d", "-b6 -r4,4", nullptr }, PN_END };
const PresetNode pn_r9[] = { { "4", "-a2", pn_r9_45 }, { "5", "-a2", pn_r9_45 }, { "1d", "", nullptr }, { "1dsq", "-b6 -r4,4", nullptr }, { "2d", "-b6 -r4,4", nullptr }, PN_END };
const PresetNode pn_ont[] = { { "r7", "-b4", pn_leaf_r7 }, { "r9", "", pn_r9 }, { "1d", "-a2", nullptr }, { "1dsq", "-a2 -b6 -r4,4", nullptr }, { "2d", "-a2 -b6 -r4,4", nullptr }, PN_END };
const PresetNode pn_pacbio[] = { { "clr", "", nullptr }, { "ccs", "-b5 -p6 -p2", nullptr }, PN_END };
const PresetNode pn_root[] = { { "pacbio", "-k15 -w10 -a2 -b4 -p4 -q2 -r3,3 -Y50 -s50 -m0.3", pn_pacbio }, { "ont", "-k15 -w10 -a3 -b5 -p6 -q2 -r3,3 -Y50 -s50 -m0.3", pn_ont },
	{ "ava", "-k15 -w5 -a2 -b3 -p0 -q2 -Y50 -s30 -m0.05", nullptr }, PN_END };
int opt_preset(mm_opt_t *o, const char *name)
{
	const PresetNode *c = pn_root; int rc = 0; bool any = false;
	split_each(name, ".:", [&](int, const std::string &t) {
		if(rc) return;
		const PresetNode *q = c; while(q && q->key && t != q->key) q++;
		if(!q || !q->key) { rc = 1; return; }
		if(opt_line(o, q->val)) { rc = 1; return; }
		c = q->kids; any = true;
	});
	return rc || !any;
}
/* the option handlers of minialign.c:5990-6099 with their range checks; a failed check is an error (the reference counts it and exits 1) */
bool opt_fail(const char *msg) { fprintf(stderr, "[E::mm_opt_parse] %s\n", msg); return true; }
int opt_one(mm_opt_t *o, char c, const char *arg)
{
	auto base_of = [](char ch) -> int { switch(ch) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': case 'U': return 4; default: return 0; } };     /* idxaf, minialign.c:232 */
	/* mm_opt_atoi / mm_opt_atof (minialign.c:5745-5768): digits only for the integer options, [0-9-.,eE] for the real ones; anything else is "unparsable number" */
	auto digits = [](const char *t, size_t n) { for(size_t i = 0; i < n && t[i]; i++) if(!isdigit((unsigned char)t[i])) return false; return true; };
	if(strchr("kwBLabpqYstWG12", c) && !digits(arg, strlen(arg))) return opt_fail("unparsable number.");
	if(strchr("rC", c)) { bool bad = false; split_each(arg, ",;:/", [&](int, const std::string &t) { if(!digits(t.c_str(), t.size())) bad = true; }); if(bad) return opt_fail("unparsable number."); }
	if(strchr("fm", c)) { for(const char *t = arg; *t; t++) if(!strchr("0123456789-.,eE", *t) && !(c == 'f' && strchr(";:/", *t))) return opt_fail("unparsable number."); }
	switch(c) {
		case 'x': return opt_preset(o, arg);
		case 'k': o->k = atoi(arg); return !(o->k > 1 && o->k < 32) && opt_fail("k must be inside [1,32).");
		case 'w': o->w = atoi(arg); return !(o->w > 1 && o->w < 32) && opt_fail("w must be inside [1,32).");
		case 'B': o->b = atoi(arg); return !(o->b > 1 && o->b < 32) && opt_fail("b must be inside [1,32).");
		case 'f': {
			bool bad = false; o->n_frq = 0;
			split_each(arg, ",;:/", [&](int i, const std::string &t) {
				if(i >= 7) { bad = true; return; }
				float f = o->frq[o->n_frq++] = (float)atof(t.c_str());
				if(!(f >= 0.0 && f < 1.0) || (i > 0 && !(o->frq[i - 1] > o->frq[i]))) bad = true;
			});
			return (bad || o->n_frq == 0) && opt_fail("frequency thresholds (-f) must be inside [0,1), descending, at most 7.");
		}
		case 'L': o->min_len = atoi(arg); return !(o->min_len > 0) && opt_fail("minimum sequence length must be > 0.");
		case 'a': { int m = atoi(arg); for(int i = 0; i < 16; i++) if((i & 3) == (i >> 2)) o->p.score_matrix[i] = (int8_t)m; return !(m > 0 && m < 7) && opt_fail("match award (-a) must be inside [1,7]."); }
		case 'b': { int x = atoi(arg); for(int i = 0; i < 16; i++) if((i & 3) != (i >> 2)) o->p.score_matrix[i] = (int8_t)-x; return !(x > 0 && x < 7) && opt_fail("mismatch penalty (-b) must be inside [1,7]."); }
		case 'e': {
			bool bad = false;
			split_each(arg, ",;:/", [&](int, const std::string &t) {
				if(t.size() < 3 || !base_of(t[0]) || !base_of(t[1])) { bad = true; return; }
				o->p.score_matrix[(base_of(t[1]) - 1) * 4 + (base_of(t[0]) - 1)] += (int8_t)atoi(t.c_str() + 2);
			});
			return bad && opt_fail("unknown base in score modifier (-e).");
		}
		case 'p': { int gi = atoi(arg); o->p.gi = (int8_t)gi; return !(gi < 32) && opt_fail("gap open penalty (-p) must be inside [0,32]."); }
		case 'q': { int ge = atoi(arg); o->p.ge = (int8_t)ge; return !(ge > 0 && ge < 32) && opt_fail("gap extension penalty (-q) must be inside [1,32]."); }
		case 'r': {
			int g[2] = { 0, 0 };
			split_each(arg, ",;:/", [&](int i, const std::string &t) { if(i == 0) g[0] = g[1] = atoi(t.c_str()); else if(i == 1) g[1] = atoi(t.c_str()); });
			o->p.gfa = (int8_t)g[0]; o->p.gfb = (int8_t)g[1];
			return !(g[0] >= 0 && g[0] < 32 && g[1] >= 0 && g[1] < 32) && opt_fail("short-gap extension penalty (-r) must be inside [0,32].");
		}
		case 'Y': { int x = atoi(arg); o->p.xdrop = (int8_t)x; return !(x > 10 && x < 128) && opt_fail("X-drop cutoff must be inside [10,128]."); }
		case 's': o->min_score = atoi(arg); return !(o->min_score > 0) && opt_fail("minimum alignment score must be > 0.");
		case 'm': o->min_ratio = (float)atof(arg); return !(o->min_ratio > 0.0 && o->min_ratio < 1.0) && opt_fail("minimum alignment score ratio must be inside [0.0,1.0].");
		case 't': o->nth = atoi(arg); return 0;             /* host threads of the reference; the device path sizes its own */
		case 'W': o->wlen = atoi(arg); return 0;
		case 'G': o->glen = atoi(arg); return 0;
		case 'd': o->fnw = arg; return o->fnw.empty();
		case '1': case '2': return 0;                      /* input batch / output buffer sizes of the reference's host pipeline: accepted, no meaning here */
		case 'v': return 0;
		case 'h': o->help = 1; return 0;
		case 'O': {
			static const struct { const char *k; uint32_t v; } t[] = { { "sam", 0 }, { "maf", 1 }, { "blast6", 2 }, { "paf", 5 } };
			for(auto &e : t) if(strcmp(arg, e.k) == 0) { o->format = e.v; return 0; }
			return opt_fail("unknown output format (-O).");
		}
		case 'c': {                      /* mm_opt_circular, minialign.c:5986-5997: no name, `*' or `-' marks every sequence */
			o->circ_set = true;
			split_each(arg, ",;:/", [&](int, const std::string &t) { if(t == "*" || t == "-") o->circ_names.clear(); else o->circ_names.push_back(t); });
			return 0;
		}
		case 'X': o->flag |= 0x01; o->ava = true; return 0;      /* MM_AVA: every file is mapped onto every file (minialign.c:6377); QUIRK kept: the bit is also the RG tag's */
		case 'A': o->flag |= 0x10; return 0;      /* MM_COMP: no effect on the mapping; QUIRK kept: the bit is also the AS tag's */
		case 'C': return 0;                       /* base ids: parsed, unused (minialign.c:3768 pins qid to 0) */
		case 'P': o->flag |= 0x08; return 0;
		case 'Q': o->keep_qual = true; return 0;
		case 'T': {                      /* mm_opt_tags + mm_print_tag2flag, minialign.c:5928, 5631 */
			static const char *const names[] = { "RG", "CO", "NH", "IH", "AS", "XS", "NM", "SA", "MD", "CG", "ID", "SQ" };
			bool bad = false;
			split_each(arg, ",;:/", [&](int, const std::string &t) { if(t.size() != 2) { bad = true; return; } for(int i = 0; i < 12; i++) if(t == names[i]) o->tags |= 1ull << i; });
			return bad && opt_fail("unknown tag (-T).");
		}
		case 'R': {                      /* mm_opt_rg, minialign.c:5890-5921: a backslash turns the next character into a tab */
			o->rg_line.clear(); o->rg_id.clear(); o->flag &= ~1ull;
			std::string line; for(const char *q = arg; *q; q++) { if(*q == '\\') { q++; line.push_back('\t'); if(!*q) break; } else line.push_back(*q); }
			bool found = false;
			split_each(line.c_str(), "\t\r\n", [&](int, const std::string &t) { if(!found && t.compare(0, 3, "ID:") == 0) { o->rg_id = t; found = true; } });
			if(!found) return opt_fail("RG line must start with @RG and contains ID, like `@RG\\tID:1'.");
			o->rg_line = line; o->flag |= 1ull; return 0;
		}
		default: fprintf(stderr, "[E::mm_opt_parse] unsupported option -%c\n", c); return 1;
	}
}
/* mm_opt_check_sanity, minialign.c:6097-6112 */
int opt_check(mm_opt_t *o)
{
	int x = 0; for(int i = 0; i < 16; i++) x = std::max(x, -(int)o->p.score_matrix[i]);
	const int gfa = o->p.gfa, gfb = o->p.gfb, ge = o->p.ge;
	if(!(gfa == 0 || gfa > ge) || !(gfb == 0 || gfb > ge)) return opt_fail("short-gap extension penalty (-r) must be larger than gap extension penalty.");
	if((gfa == 0) != (gfb == 0)) return opt_fail("short-gap extension penalty (-r) must be set for both sides.");
	if(!(gfa == 0 || gfb == 0 || gfa + gfb > x)) return opt_fail("short-gap extension penalty (-r) must not be greater than mismatch penalty.");
	return 0;
}
} /* anonymous */

extern "C" mm_opt_t *mm_opt_init(void) { return new mm_opt_s(); }
extern "C" void mm_opt_destroy(mm_opt_t *o) { delete o; }
extern "C" int mm_opt_parse(mm_opt_t *o, int argc, char const *const *argv, char const **files, int max_files, int *n_files)
{
	int nf = 0;
	o->arg_line.clear();
	for(int i = 0; i < argc; i++) { if(i) o->arg_line += ' '; o->arg_line += argv[i]; }    /* mm_join(argv, ' '), minialign.c:6163 */
	/* the walk of mm_opt_parse_argv (minialign.c:5786-5812): a word that does not start with '-' (or is "-" alone) is positional; behind the dash the boolean
	 * letters (X A P Q h) are eaten one by one, the first other letter is the option, its argument is the rest of the word or -- when the word ends there -- the
	 * next word unless that one looks like an option; a required argument that is missing and an unknown letter are errors */
	auto isarg = [](const char *w) { return w[0] != '-' || w[1] == 0; };
	for(int i = 1; i < argc; i++) {
		const char *q = argv[i];
		if(isarg(q)) { if(nf < max_files) files[nf++] = q; continue; }
		while(*++q && strchr("XAPQh", *q)) { if(opt_one(o, *q, "")) return 1; }
		if(*q == 0) continue;
		const bool req = strchr("xRTOdtkwfBLWGabepqrYsm12", *q) != NULL, optl = strchr("cvC", *q) != NULL;
		if(!req && !optl) { fprintf(stderr, "[E::mm_opt_parse] unknown option `-%c'.\n", *q); return 1; }
		const char *r = q[1] ? q + 1 : ((i + 1 < argc && isarg(argv[i + 1])) ? argv[++i] : NULL);
		if(req && !r) { fprintf(stderr, "[E::mm_opt_parse] missing argument for option `-%c'.\n", *q); return 1; }
		if(opt_one(o, *q, r ? r : "")) return 1;
	}
	if(n_files) *n_files = nf;
	if(opt_check(o)) return 1;
	if(o->w >= 32) o->w = (uint32_t)(int)(2.0 / 3.0 * o->k + .499);       /* minialign.c:6111 */
	return 0;
}
