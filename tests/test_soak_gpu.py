"""GPU: a short soak -- eight seeded sets of unusual shape (tiny and very long reads, a 0.9 repeat fraction, 2 000 contigs, small k / w, permissive thresholds,
plain affine gaps, a circular reference smaller than its reads) through the command-line program and the compiled reference, whole outputs compared
(tools/soak.sh; the long form of this run is profiles/round1_i_soak.txt).  Skipped where oracle/_ref did not travel."""
import ctypes, os, subprocess, tempfile, threading, time
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu

def test_short_soak_against_the_compiled_reference():
    if not os.path.exists(os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')): pytest.skip('oracle/_ref not built')
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(['bash', os.path.join(M.ROOT, 'tools', 'soak.sh'), d, '4100', '8', '12'], cwd=M.ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        log = r.stdout.decode()
        assert 'mismatches: 0 of 8' in log, log[-3000:]


def test_a_context_left_idle_maps_the_same_bytes_when_it_resumes():
    """a device context that has mapped a long-tailed set (the workspace ladder, jobs, helper waves, recycled device buffers all in play), then sits idle for a while (twice eight seconds) while
    the host churns through memory (page cache and anonymous pages come and go: what a neighbouring CPU job does to a box), then maps again -- twice: same bytes every time,
    and the oracle's.  (Round 3 saw one `Memory access fault by GPU ... address (nil)' in a bench process whose device was idle under heavy host memory pressure.)"""
    from minialign_amd import multi
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', 9101, 5000000, 3, 0.3, out=ref); M.gensim('reads', 9102, ref, 2.0, 'ont', 'fa', out=rd)
        want = b''.join(l for l in subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-xont.1dsq', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.splitlines(True) if not l.startswith(b'@'))
        keep = {k: os.environ.get(k) for k in ('MM_SLAB_GB', 'MM_LANES', 'MM_BATCH_BASES')}
        os.environ.update(MM_SLAB_GB='8', MM_LANES='3', MM_BATCH_BASES='4000000')
        try:
            L = multi.load_library(); assert L.mm_set_device(0) == 0
            o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 4)(b'minialign', b'-xont.1dsq', ref.encode(), rd.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
            assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
            mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
            text = open(rd, 'rb').read(); addr = ctypes.cast(ctypes.c_char_p(text), ctypes.c_void_p).value
            def once(): return multi.ShardMapper(L, al, None, 0, 0, lanes=3, text=(addr, len(text))).map(0).col.text()
            assert once() == want
            stop = threading.Event()
            def churn():          # 4 GB at a time, touched and dropped: bounded, nowhere near the box's memory
                while not stop.is_set():
                    b = bytearray(4 << 30)
                    for i in range(0, len(b), 1 << 20): b[i] = 1
                    del b; time.sleep(0.2)
            t = threading.Thread(target=churn); t.start()
            for _ in range(2):
                time.sleep(8)
                assert once() == want
            stop.set(); t.join()
            L.mm_align_destroy(al); L.mm_idx_destroy(mi)
        finally:
            for k, v in keep.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
