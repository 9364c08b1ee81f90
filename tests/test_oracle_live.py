"""The oracle program against the compiled reference, live, on a few small seeded sets of varied shape and option line (tools/oracle_soak.sh).
Runs where oracle/_ref/ exists (the build container and any box the built tree travelled to); the committed goldens cover the rest."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle/_ref/minialign')), reason='compiled reference not present')
def test_oracle_matches_reference_on_fresh_sets(tmp_path):
	for f in ('oracle/ora_minialign', 'tools/gensim'):
		assert os.path.exists(os.path.join(ROOT, f)), f + ' missing: run __graft_entry__.build()'
	out = tmp_path / 'osk.txt'
	r = subprocess.run(['bash', 'tools/oracle_soak.sh', str(out), '7300', '6'], cwd=ROOT, capture_output=True, text=True, timeout=600)
	sys.stdout.write(r.stdout[-2000:])
	assert 'mismatches 0 of 6' in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle/_ref/minialign')), reason='compiled reference not present')
def test_primed_part_runs_reproduce_the_single_stream(tmp_path):
	"""The method the full-size GPU tests use to have the compiled reference map a WHOLE set in processes side by side (tests/test_headline_gpu.py
	_reference_by_parts): a process maps consecutive parts, fed the last reads of the part in front first, and tools/samcheck --parts gives a digest per part.  On a
	many-contig set -- where the carried reference length differs from read to read -- the digests must be those of ONE -t1 run over the whole set."""
	import json
	sys.path.insert(0, os.path.join(ROOT, 'tests'))
	import mmlib as M, test_headline_gpu as H
	d = str(tmp_path); ref = os.path.join(d, 'ref.fa'); n_parts = 6
	M.gensim('genome', 9301, 8000000, 60, 0.1, out=ref)
	whole = os.path.join(d, 'rd.fa'); spans = []; at = 0
	with open(whole, 'wb') as g:
		for p in range(n_parts):
			b = subprocess.run([H.GENSIM, 'reads', '9302', ref, '1.5', 'pacbio', 'fa', '6000', '2500', str(p), str(n_parts)], capture_output=True, check=True).stdout
			g.write(b); spans.append((at, len(b))); at += len(b)
	one = subprocess.run('%s -xpacbio -t1 %s %s 2>/dev/null | %s --parts' % (H.REFBIN, ref, whole, H._samcheck()), shell=True, capture_output=True, text=True, check=True)
	want = [tuple(x) for x in json.loads(one.stdout.strip().splitlines()[-1])['parts']]
	assert len(want) == n_parts and all(x[0] > 0 for x in want)
	for group in (1, 2, 3):
		out = os.path.join(d, 'g%d' % group)
		assert H._reference_by_parts('pacbio', ref, whole, spans, out, threads=4, primer=4, window=1 << 20, group=group).wait(timeout=300) == 0
		assert H._parts_of(out, n_parts, group=group) == want, group


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle/_ref/minialign')), reason='compiled reference not present')
def test_oracle_matches_reference_on_hard_repeats(tmp_path):
	"""the oracle against the compiled reference on a reference with mammalian repeat structure (tools/gensim.c genomehard: SINE / LINE-like families, satellite arrays,
	segmental duplications, N gaps): reads with dozens of chains, rescue rounds, secondary records"""
	sys.path.insert(0, os.path.join(ROOT, 'tests'))
	import mmlib as M
	d = str(tmp_path); ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
	M.gensim('genomehard', 9401, 20000000, 4, 0.45, out=ref); M.gensim('reads', 9402, ref, 0.06, 'pacbio', 'fa', 8000, 3000, out=rd)
	strip = lambda b: b''.join(l for l in b.splitlines(True) if not l.startswith(b'@PG'))
	a = subprocess.run([os.path.join(ROOT, 'oracle/_ref/minialign'), '-xpacbio', '-t1', ref, rd], capture_output=True, check=True).stdout
	b = subprocess.run([os.path.join(ROOT, 'oracle/ora_minialign'), '-xpacbio', ref, rd], capture_output=True, check=True).stdout
	assert strip(a) == strip(b) and a.count(b'\n') > 150
	assert sum(1 for l in a.splitlines() if not l.startswith(b'@') and int(l.split(b'\t')[1]) & 0x100) > 20          # secondary records: the repeats bite
