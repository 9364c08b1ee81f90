"""CPU tests (no GPU) for the mapper-level oracle (oracle/ora_mm.c): byte-identical SAM against the committed golden SAM of
the compiled reference, per-stage vectors (sketch / sorted seeds / chain roots) against committed digests, and -- when
oracle/_ref is present -- live comparison with the reference binary and its stage harness."""
import gzip, hashlib, json, os, subprocess, tempfile
import numpy as np, pytest
import mmlib as M
from golden.make_mm_golden import make_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, 'golden', 'mm_golden.json')))

@pytest.fixture(scope='module')
def workdir():
    with tempfile.TemporaryDirectory() as d:
        yield d

def _strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

@pytest.mark.parametrize('s', META['sets'], ids=[s['name'] for s in META['sets']])
def test_oracle_sam_matches_golden(s, workdir):
    ref, rd = make_inputs(s, workdir)
    assert hashlib.md5(open(ref, 'rb').read()).hexdigest() == s['ref_md5'], 'generator stream changed'
    assert hashlib.md5(open(rd, 'rb').read()).hexdigest() == s['reads_md5']
    got = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-x' + s['preset'], ref, rd],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    want = gzip.open(os.path.join(HERE, 'golden', s['name'] + '.sam.gz')).read()
    assert hashlib.md5(want).hexdigest() == s['sam_md5']
    assert got == want

@pytest.mark.parametrize('s', META['sets'], ids=[s['name'] for s in META['sets']])
def test_oracle_stages_match_golden(s, workdir):
    ref, rd = make_inputs(s, workdir)
    o = M.OracleMM(s['preset'], M.read_fasta(ref))
    assert o.occ() == s['occ']
    reads = dict(M.read_fasta(rd))
    for st in s['stages']:
        q = reads[st['read']]
        sk = o.sketch(q)
        assert len(sk) == st['sketch_n'] and [int(x) for x in sk[:8]] == st['sketch_head']
        assert hashlib.md5(sk.tobytes()).hexdigest() == st['sketch_md5']
        sd = o.seed(q)
        assert len(sd) == st['seed_n'] and hashlib.md5(sd.tobytes()).hexdigest() == st['seed_md5']
        assert [int(x) for x in o.chain()] == st['chain']

@pytest.mark.skipif(not M.RefMM.available(), reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_vs_live_reference_multicontig(workdir):
    """a fresh set with many contigs: exercises the rlen state carried across reads (minialign.c:3864)"""
    s = dict(name='live_mc', preset='pacbio', genome=(201, 300000, 25, 0.05), reads=(202, 0.5, 'pacbio', 'fa', 3000, 1000))
    ref, rd = make_inputs(s, workdir)
    a = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign'), '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    b = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    assert a == b


@pytest.mark.parametrize('fmt', ['fa', 'fq'])
@pytest.mark.parametrize('preset', ['pacbio', 'ont.1dsq'])
def test_oracle_edge_cases_match_reference_golden(preset, fmt, workdir):
    """hand-built edge cases (reads shorter than k, all-N, N runs, unmappable, both strands, chimera, contig ends, a read the
    reference drops from its output altogether, FASTQ input): the golden SAM is the compiled reference's
    (tests/golden/make_edge_golden.py)"""
    import gzip
    from golden.make_edge_golden import make_edge_inputs, strip_pg
    ref, rd = make_edge_inputs(workdir, fmt)
    got = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-x' + preset, ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'edge_%s_%s.sam.gz' % (preset.replace('.', ''), fmt))).read()
    assert got == want


def _opt_lines():
    from golden.make_opt_golden import OPTION_LINES
    return OPTION_LINES

@pytest.mark.parametrize('name,opts', _opt_lines(), ids=[n for n, _ in _opt_lines()])
def test_oracle_option_lines_match_reference_golden(name, opts, workdir):
    """options beyond the presets (sketch, index thresholds / bucket bits / length filter, score matrix incl. modifiers, plain affine gaps,
    chaining windows, X-drop): golden SAM from the compiled reference (tests/golden/make_opt_golden.py)"""
    import gzip
    from golden.make_opt_golden import make_opt_inputs, strip_pg
    ref, rd = make_opt_inputs(workdir)
    got = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'opt_%s.sam.gz' % name)).read()
    assert got == want


def _tag_lines():
    from golden.make_tag_golden import TAG_LINES
    return TAG_LINES

@pytest.mark.parametrize('name,opts', _tag_lines(), ids=[n for n, _ in _tag_lines()])
def test_oracle_optional_sam_fields_match_reference_golden(name, opts, workdir):
    """-T tags (RG CO NH IH AS XS NM SA MD), -R, -Q, -P on FASTQ reads with header comments: golden SAM from the compiled reference
    (tests/golden/make_tag_golden.py), including its quirks (shared flag / tag word, SA names and mapping qualities, tabs in read names)"""
    import gzip
    from golden.make_tag_golden import inputs_for, strip_pg
    ref, rd = inputs_for(name, workdir)
    got = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'tag_%s.sam.gz' % name)).read()
    assert got == want


def _circ_lines():
    from golden.make_circ_golden import CIRC_LINES
    return CIRC_LINES

@pytest.mark.parametrize('name,opts', _circ_lines(), ids=[n for n, _ in _circ_lines()])
def test_oracle_circular_references_match_reference_golden(name, opts, workdir):
    """-c: wrap-around minimizers in the index, chains linked across the origin (mm_circularize), extension continuing into the reference itself, the two
    segments printed as primary + supplementary: golden SAM from the compiled reference (tests/golden/make_circ_golden.py)"""
    import gzip
    from golden.make_circ_golden import make_circ_inputs, strip_pg
    ref, rd = make_circ_inputs(workdir)
    got = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'circ_%s.sam.gz' % name)).read()
    assert got == want


def test_oracle_reader_matches_reference_on_oddly_formatted_files(workdir):
    """the reader's view of oddly formatted FASTA / FASTQ read files (names, bases, qualities, comments of the -- mostly unmapped -- records, and the runs
    the reference gives up on): golden output of the compiled reference (tests/golden/make_parse_golden.py)"""
    import gzip, json
    from golden.make_parse_golden import make_parse_inputs, strip_pg, OPTS
    gold = json.loads(gzip.open(os.path.join(HERE, 'golden', 'parse_cases.json.gz')).read())
    ref, files = make_parse_inputs(workdir)
    for name, p in files.items():
        r = subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + OPTS + [ref, p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60)
        assert (r.returncode != 0) == (gold[name][0] != 0), name
        assert strip_pg(r.stdout).decode('latin1') == gold[name][1], name
