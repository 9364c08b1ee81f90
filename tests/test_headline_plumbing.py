"""CPU: the plumbing of the long GPU tests (tests/test_headline_gpu.py, tests/test_hard_gpu.py) run small, with the ORACLE's command-line program standing where the device
program stands in them -- set generation in parts, byte spans, the compiled reference over every part in primed processes side by side, tools/samcheck's per-part digests, every
assertion.  Those tests take minutes of a GPU box each and cannot be tried here; a slip in their own code should not be what a box is spent on.  (It also says once more, on
four small sets of three kinds, that the oracle's records are the compiled reference's.)"""
import os, sys, shutil, tempfile
import pytest
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mmlib as M
import test_headline_gpu as H
import test_hard_gpu as HH

SMALL = dict(dm6_genome=(0x5eed0001, 2000000, 52, 0.05), dm6_reads=(0x5eed0002, 2.5, 'pacbio'), dm6_min_bases=2e6,
             hg38_genome=(0x5eed0001, 4000000, 25, 0.05), pb_reads=(0x5eed0002, 1.5, 'pacbio'), pb_min_bases=3e6, pb_min_bytes=1e6, pb_min_reads=100,
             ont_reads=(0x5eed0003, 1.0, 'ont'), ont_min_bases=1e6, ont_min_reads=10,
             hg38hard_genome=(0x5eed0001, 4000000, 25, 0.45), hg38hard_reads=(0x5eed0002, 1.0, 'pacbio'), hg38hard_min_bases=2e6, hg38hard_min_reads=100,
             hard_genome=(0x5eed0011, 3000000, 4, 0.45), hard_reads=(0x5eed0012, 0.7, 'pacbio'), hard_min_reads=50, index_threads=4)

@pytest.mark.skipif(not os.path.exists(H.REFBIN), reason='compiled reference not present')
def test_the_long_gpu_tests_run_small_with_the_oracle_as_the_mapper(monkeypatch):
	monkeypatch.setattr(H, 'SIZES', SMALL)
	monkeypatch.setattr(H, 'CLI', os.path.join(M.ROOT, 'oracle', 'ora_minialign'))
	monkeypatch.setattr(H, 'AT_SCALE_ON_ONE_GPU', False)
	work = tempfile.mkdtemp(prefix='mmplumb_')
	try:
		sets = H._hg38_sets(work); fx = next(sets)
		H.test_dm6_size_x20_whole_sam_equals_the_reference(work, fx)
		H.test_hg38_size_x3_whole_set_equals_the_reference_and_device_contexts(work, fx)
		H.test_hg38_size_ont_like_whole_set_equals_the_reference(work, fx)
		H.test_hg38_size_hard_repeat_whole_set_equals_the_reference(work, fx)
		with pytest.raises(StopIteration): next(sets)
		HH.test_hard_repeats_20000_reads_equal_the_reference()
	finally:
		shutil.rmtree(work, ignore_errors=True)
