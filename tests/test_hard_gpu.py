"""GPU: a reference with the repeat structure of a mammalian genome (tools/gensim.c genomehard: SINE / LINE-like families of 10^4 .. 10^5 copies at this size, two hundred
middle-sized families, segmental duplications, satellite arrays, long N gaps -- 45 % repeat content, where the other sets plant 5 %) x 20 000 PBSIM-like reads, through the
command-line program, EVERY record against the compiled reference at -t1 (its parts in processes side by side, primed as in tests/test_headline_gpu.py).  This is where
the occurrence thresholds cut, where reads carry dozens of chains and secondary records, where the pools' demand per base is several times that of the random sets
(README.md:48-51: the real human genome maps 3.3 x slower per base than E.coli).  Skipped where oracle/_ref did not travel."""
import json, os, shutil, subprocess, sys, tempfile
import pytest
import mmlib as M
import test_headline_gpu as H

pytestmark = pytest.mark.gpu

def test_hard_repeats_20000_reads_equal_the_reference():
    if not os.path.exists(H.REFBIN): pytest.skip('oracle/_ref not built')
    d = tempfile.mkdtemp(prefix='mmhard_')
    try:
        ref, rd, spans = H._generate(d, 'hard', H.SIZES['hard_genome'], H.SIZES['hard_reads'], keep_parts=True, hard=True)
        bg = H._reference_by_parts('pacbio', ref, rd, spans, os.path.join(d, 'hard_ref'), group=2)          # (a 400 Mb index: 2.5 GB per process)
        s, err, sec = H._map_through_samcheck([H.CLI, '-xpacbio', ref, rd], rd, 0, os.devnull, timeout=300)
        assert s['error'] == '' and s['reads'] == s['primary'] and s['reads'] >= H.SIZES['hard_min_reads'], s
        assert bg.wait(timeout=900) == 0, open(os.path.join(d, 'hard_ref.idx.err')).read()[-2000:]
        want = H._parts_of(os.path.join(d, 'hard_ref'), H.PARTS, group=2); got = [tuple(x) for x in s['parts']]
        bad = [p for p in range(H.PARTS) if got[p] != want[p]]
        assert not bad, 'hard-repeat set: parts %r differ from the compiled reference (ours %r, reference %r)' % (bad, [got[p] for p in bad], [want[p] for p in bad])
        assert sum(x[0] for x in want) == s['records']
        sys.stderr.write('[hard] %d reads, %d records (%.2f per read: %d secondary, %d supplementary), mapped %d, %.1f s with the index build; %s\n' % (
            s['reads'], s['records'], s['records'] / max(1, s['reads']), s['secondary'], s['supplementary'], s['mapped'], sec, err.decode().strip().splitlines()[-1][:300]))
    finally:
        shutil.rmtree(d, ignore_errors=True)
