"""GPU parity for the whole mapper path: SAM from the HIP pipeline (minialign_amd/minialign, i.e. mm_main in
libminialign_amd.so) must be byte-identical to (1) the committed golden SAM of the compiled reference, (2) the CPU oracle,
and (3) -- when it travelled with the snapshot -- the reference binary itself, on seeded synthetic sets."""
import gzip, hashlib, json, os, subprocess, tempfile
import pytest
import mmlib as M
from golden.make_mm_golden import make_inputs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, 'golden', 'mm_golden.json')))
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')

def _strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def _run(exe, preset, ref, rd):
    r = subprocess.run([exe, '-x' + preset, ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return _strip_pg(r.stdout)

def _first_diff(a, b):
    la, lb = a.split(b'\n'), b.split(b'\n')
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return 'line %d:\n  got  %r\n  want %r' % (i, x[:300], y[:300])
    return 'lengths differ: %d vs %d lines' % (len(la), len(lb))

@pytest.fixture(scope='module')
def workdir():
    with tempfile.TemporaryDirectory() as d:
        yield d

@pytest.mark.parametrize('s', META['sets'], ids=[s['name'] for s in META['sets']])
def test_sam_matches_golden(s, workdir):
    ref, rd = make_inputs(s, workdir)
    got = _run(CLI, s['preset'], ref, rd)
    want = gzip.open(os.path.join(HERE, 'golden', s['name'] + '.sam.gz')).read()
    assert got == want, _first_diff(got, want)

@pytest.mark.parametrize('s', [
    dict(name='g_multi', preset='pacbio', genome=(301, 600000, 30, 0.10), reads=(302, 1.0, 'pacbio', 'fa', 4000, 1500)),
    dict(name='g_rep', preset='pacbio', genome=(311, 250000, 2, 0.50), reads=(312, 1.0, 'pacbio', 'fa', 5000, 2000)),
    dict(name='g_ont', preset='ont.1dsq', genome=(321, 500000, 6, 0.10), reads=(322, 1.0, 'ont', 'fa')),
    # the shape of BASELINE configs[2] (dm6: many contigs) scaled down: 300 contigs, the carried reference length changes all the time
    dict(name='g_dm6like', preset='pacbio', genome=(341, 12000000, 300, 0.08), reads=(342, 0.5, 'pacbio', 'fa', 6000, 2000)),
    dict(name='g_dm6like_ont', preset='ont.1dsq', genome=(351, 6000000, 150, 0.08), reads=(352, 0.5, 'ont', 'fa')),
], ids=lambda s: s['name'])
def test_sam_matches_oracle(s, workdir):
    ref, rd = make_inputs(s, workdir)
    got = _run(CLI, s['preset'], ref, rd)
    want = _run(os.path.join(M.ROOT, 'oracle', 'ora_minialign'), s['preset'], ref, rd)
    assert got == want, _first_diff(got, want)
    refbin = os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')
    if os.path.exists(refbin):
        assert got == _run(refbin, s['preset'], ref, rd)


@pytest.mark.parametrize('fmt', ['fa', 'fq'])
@pytest.mark.parametrize('preset', ['pacbio', 'ont.1dsq'])
def test_edge_cases_match_reference_golden(preset, fmt, workdir):
    """the hand-built edge-case set of tests/golden/make_edge_golden.py (reads shorter than k, all-N, N runs, unmappable,
    chimera, contig ends, FASTQ, ...) through the HIP pipeline against the compiled reference's SAM"""
    from golden.make_edge_golden import make_edge_inputs
    ref, rd = make_edge_inputs(workdir, fmt)
    got = _run(CLI, preset, ref, rd)
    want = gzip.open(os.path.join(HERE, 'golden', 'edge_%s_%s.sam.gz' % (preset.replace('.', ''), fmt))).read()
    assert got == want, _first_diff(got, want)


def test_many_batches_on_two_lanes_give_the_same_sam(workdir):
    """the command-line program cuts its input into batches that alternate between two lanes of the device context (pack / upload /
    run of one batch overlaps D2H / SAM of the previous one); the carried reference length is handed from batch to batch.  With
    MM_BATCH_BASES the batches are made tiny: dozens of hand-overs on a 25-contig reference where the carried value keeps changing."""
    s = dict(name='g_lanes', preset='pacbio', genome=(331, 400000, 25, 0.10), reads=(332, 1.5, 'pacbio', 'fa', 3000, 1000))
    ref, rd = make_inputs(s, workdir)
    want = _run(CLI, s['preset'], ref, rd)
    for bb in ('20000', '150000'):
        r = subprocess.run([CLI, '-x' + s['preset'], ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_BATCH_BASES=bb))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = _strip_pg(r.stdout)
        assert got == want, _first_diff(got, want)
    assert want == _run(os.path.join(M.ROOT, 'oracle', 'ora_minialign'), s['preset'], ref, rd)


def test_the_first_read_of_a_batch_starts_from_what_the_batch_in_front_predicts(workdir):
    """DESIGN.md 5, "the value across the batch border": with 25 contigs the value the stream had when a batch was taken (three or four batches back) is the wrong
    contig's length for almost every batch, and each time that flips the first read's `apos >= rlen` test the read is mapped again behind the check.  The lane in front
    posts what its batch expects to leave (PredBoard) and the next batch starts from that: with some 80 batches the re-runs stay a handful (the program's closing line
    counts them) -- and the records are the oracle's, whatever was guessed."""
    import re
    s = dict(name='g_border', preset='pacbio', genome=(371, 1500000, 25, 0.05), reads=(372, 2.0, 'pacbio', 'fa', 5000, 1500))
    ref, rd = make_inputs(s, workdir)
    r = subprocess.run([CLI, '-x' + s['preset'], ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_BATCH_BASES='40000', MM_VERBOSE='1'))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    err = r.stderr.decode(errors='replace')
    batches = len(set(re.findall(r'batch (\d+) \(device', err)))
    m = re.search(r'(\d+) re-run\(s\)', err)
    assert m and batches >= 40, (batches, err[-600:])
    assert int(m.group(1)) <= batches // 10, 'batches %d, re-runs %s' % (batches, m.group(1))          # (82 batches, 0 re-runs on round 6's tree; a first read starting from a stale value is mapped again in a third of the batches)
    want = _run(os.path.join(M.ROOT, 'oracle', 'ora_minialign'), s['preset'], ref, rd)
    got = _strip_pg(r.stdout)
    assert got == want, _first_diff(got, want)


def test_very_long_reads_with_a_small_workspace_budget(workdir):
    """the DP workspace of a persistent wave grows with the longest read of the batch; MM_SLAB_GB caps what a lane may take, in which
    case fewer waves are launched.  Reads of tens of kilobases under a 1 GB cap against the oracle."""
    s = dict(name='g_long', preset='pacbio', genome=(341, 900000, 2, 0.10), reads=(342, 1.2, 'pacbio', 'fa', 40000, 8000))
    ref, rd = make_inputs(s, workdir)
    r = subprocess.run([CLI, '-x' + s['preset'], ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_SLAB_GB='1'))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = _strip_pg(r.stdout)
    assert got == _run(CLI, s['preset'], ref, rd)
    want = _run(os.path.join(M.ROOT, 'oracle', 'ora_minialign'), s['preset'], ref, rd)
    assert got == want, _first_diff(got, want)


def test_prebuilt_index_file_gives_the_same_sam(workdir):
    """`minialign -d idx.mai ref.fa` then `minialign idx.mai reads.fa` (main_index / the .mai branch of main_align, minialign.c:6293-6436):
    same records as the on-the-fly index; reads from stdin when no query file is named; two blocks give two headers and both sets of records"""
    s = dict(name='g_mai', preset='pacbio', genome=(371, 250000, 3, 0.10), reads=(372, 0.8, 'pacbio', 'fa', 3000, 1000))
    ref, rd = make_inputs(s, workdir)
    mai = os.path.join(workdir, 'g_mai.mai')
    want = _run(CLI, s['preset'], ref, rd)
    assert subprocess.run([CLI, '-x' + s['preset'], '-d', mai, ref]).returncode == 0
    assert _run(CLI, s['preset'], mai, rd) == want
    r = subprocess.run([CLI, '-x' + s['preset'], mai], stdin=open(rd, 'rb'), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and _strip_pg(r.stdout) == want
    assert subprocess.run([CLI, '-x' + s['preset'], '-d', mai, ref, ref]).returncode == 0
    assert _run(CLI, s['preset'], mai, rd) == want + want
    # gzip-compressed reference and reads
    open(ref + '.gz', 'wb').write(gzip.compress(open(ref, 'rb').read())); open(rd + '.gz', 'wb').write(gzip.compress(open(rd, 'rb').read()))
    assert _run(CLI, s['preset'], ref + '.gz', rd + '.gz') == want


def _opt_lines():
    from golden.make_opt_golden import OPTION_LINES
    return OPTION_LINES

@pytest.mark.parametrize('name,opts', _opt_lines(), ids=[n for n, _ in _opt_lines()])
def test_option_lines_match_reference_golden(name, opts, workdir):
    """the command-line options beyond the presets through the HIP pipeline against the compiled reference's SAM (tests/golden/make_opt_golden.py):
    other k / w, bucket bits, 3 and 5 occurrence thresholds (= rescue rounds), the length filter on both sides, plain affine gaps (-r0), an
    asymmetric score matrix (-e), chaining windows and X-drop"""
    from golden.make_opt_golden import make_opt_inputs
    ref, rd = make_opt_inputs(workdir)
    r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = _strip_pg(r.stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'opt_%s.sam.gz' % name)).read()
    assert got == want, _first_diff(got, want)


def test_options_the_reference_rejects_are_rejected(workdir):
    for bad in (['-k40'], ['-w1'], ['-a9'], ['-r1,1'], ['-xpacbio', '-r3,0'], ['-Y5'], ['-f0.1,0.2'], ['-m1.5'], ['-xnosuch'], ['-eAZ1']):
        r = subprocess.run([CLI] + bad + ['/dev/null', '/dev/null'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and r.stdout == b'', bad


def _tag_lines():
    from golden.make_tag_golden import TAG_LINES
    return TAG_LINES

@pytest.mark.parametrize('name,opts', _tag_lines(), ids=[n for n, _ in _tag_lines()])
def test_optional_sam_fields_match_reference_golden(name, opts, workdir):
    """-T tags (RG CO NH IH AS XS NM SA MD), -R read group, -Q qualities, -P through the HIP pipeline against the compiled reference's SAM
    (tests/golden/make_tag_golden.py)"""
    from golden.make_tag_golden import inputs_for
    ref, rd = inputs_for(name, workdir)
    r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = _strip_pg(r.stdout)
    want = gzip.open(os.path.join(HERE, 'golden', 'tag_%s.sam.gz' % name)).read()
    assert got == want, _first_diff(got, want)


def _circ_lines():
    from golden.make_circ_golden import CIRC_LINES
    return CIRC_LINES

@pytest.mark.parametrize('name,opts', _circ_lines(), ids=[n for n, _ in _circ_lines()])
def test_circular_references_match_reference_golden(name, opts, workdir):
    """-c through the HIP pipeline: reads across the origin of circular references on both strands, a read longer than the circle, by-name selection;
    also through an index file, which keeps the circular flags"""
    from golden.make_circ_golden import make_circ_inputs
    ref, rd = make_circ_inputs(workdir)
    want = gzip.open(os.path.join(HERE, 'golden', 'circ_%s.sam.gz' % name)).read()
    r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = _strip_pg(r.stdout)
    assert got == want, _first_diff(got, want)
    mai = os.path.join(workdir, 'circ_%s.mai' % name)
    assert subprocess.run([CLI] + opts + ['-d', mai, ref], stderr=subprocess.DEVNULL).returncode == 0
    r = subprocess.run([CLI] + [o for o in opts if not o.startswith('-c') and o != 'plasmid'] + [mai, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and _strip_pg(r.stdout) == want


def test_oddly_formatted_read_files_match_reference_golden(workdir):
    """wrapped / CRLF / lower-case / IUPAC / blank lines / missing final newline / tabs in headers / empty records / a delimiter inside a sequence line /
    FASTQ variants, read with -Q -T CO: records as the compiled reference prints them; where the reference gives up (exit 1 after the header), so does this"""
    import json
    from golden.make_parse_golden import make_parse_inputs, OPTS
    gold = json.loads(gzip.open(os.path.join(HERE, 'golden', 'parse_cases.json.gz')).read())
    ref, files = make_parse_inputs(workdir)
    for name, p in files.items():
        r = subprocess.run([CLI] + OPTS + [ref, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert (r.returncode != 0) == (gold[name][0] != 0), (name, r.stderr.decode()[-500:])
        assert _strip_pg(r.stdout).decode('latin1') == gold[name][1], name


def test_traceback_that_leaves_the_band_is_dropped_like_the_reference(workdir):
    """-xava on a read whose weak secondary hits drift out of the band: one of the up-extensions walks back across the lower band edge (q = -1, masks read
    with the wrapped lane as the reference's shifts do) and never returns -- the reference drops it (gaba_dp_trace NULL, gaba.c:3324).  Regression: the
    cached mask words used 0xffffffff as their "none" mark, which is also q = -1."""
    ref = os.path.join(workdir, 'oob.ref.fa'); rd = os.path.join(workdir, 'oob.reads.fa'); one = os.path.join(workdir, 'oob.one.fa')
    M.gensim('genome', 114, 2000000, 6, 0.25, out=ref); M.gensim('reads', 115, ref, 25, 'pacbio', 'fa', 15000, 4000, out=rd)
    keep = False
    with open(rd, 'rb') as f, open(one, 'wb') as g:
        for line in f:
            if line.startswith(b'>'): keep = line.startswith(b'>r519_') or line.startswith(b'>r52')
            if keep: g.write(line)
    got = _run(CLI, 'ava', ref, one)
    want = _run(os.path.join(M.ROOT, 'oracle', 'ora_minialign'), 'ava', ref, one)
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize('opts', [[], ['-TSA,AS,NM'], ['-P']], ids=['default-tags', 'SA-tag', 'omit-secondaries'])
def test_cigar_strings_made_on_the_device_are_the_hosts(opts, workdir):
    """K4 (csrc/mm_cigar.hpp): for SAM without MD tags the run lengths of every segment are made on the device and only their text comes back; MM_HOST_CIGAR takes the
    form of rounds 1-5 (path words back, the host walks them).  Same bytes -- on a repeat-rich multi-contig set (secondary and supplementary records, SA entries) and on
    the oracle's records; with -T MD the host walks the paths whatever the switch says (the third run)"""
    s = dict(name='g_cig', preset='pacbio', genome=(361, 2000000, 8, 0.40), reads=(362, 1.5, 'pacbio', 'fa', 7000, 2000))
    ref, rd = make_inputs(s, workdir)
    def run(env, extra=[]):
        r = subprocess.run([CLI, '-x' + s['preset']] + opts + extra + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_BATCH_BASES='250000', MM_VERBOSE='1', **env))          # (K4 runs for streams of two batches per lane and more)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return _strip_pg(r.stdout), r.stderr
    dev, err_d = run({}); host, err_h = run(dict(MM_HOST_CIGAR='1'))
    assert dev == host, _first_diff(dev, host)
    assert b'CIGAR text made on the device' in err_d and b'CIGAR text made on the device' not in err_h
    if not opts:
        r = subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-x' + s['preset'], ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE); assert r.returncode == 0
        want = _strip_pg(r.stdout); assert dev == want, _first_diff(dev, want)
        md, err_m = run({}, ['-TMD']); assert b'CIGAR text made on the device' not in err_m and md.count(b'MD:Z:') > 100
