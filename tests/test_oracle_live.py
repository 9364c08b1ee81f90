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
