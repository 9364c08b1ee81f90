"""GPU: a short soak -- eight seeded sets of unusual shape (tiny and very long reads, a 0.9 repeat fraction, 2 000 contigs, small k / w, permissive thresholds,
plain affine gaps, a circular reference smaller than its reads) through the command-line program and the compiled reference, whole outputs compared
(tools/soak.sh; the long form of this run is profiles/round1_i_soak.txt).  Skipped where oracle/_ref did not travel."""
import os, subprocess, tempfile
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu

def test_short_soak_against_the_compiled_reference():
    if not os.path.exists(os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')): pytest.skip('oracle/_ref not built')
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(['bash', os.path.join(M.ROOT, 'tools', 'soak.sh'), d, '4100', '8', '12'], cwd=M.ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        log = r.stdout.decode()
        assert 'mismatches: 0 of 8' in log, log[-3000:]
