"""GPU, BASELINE.json configs[2..4] at FULL size, through the command-line program (the drop-in), against the compiled reference (oracle/_ref/minialign -t1:
with several threads the reference's own output depends on which thread buffer a read lands in):

  (i)   D.melanogaster dm6-size reference (143.7 Mb, 1 870 contigs) x PBSIM-like x20 (2.87 Gb): the WHOLE SAM byte for byte, and the same set split over 8 ranks of
        minialign_amd.multi (all on cuda:0) identical to the single stream;
  (ii)  human hg38-size reference (3.1 Gb, 25 contigs) x PBSIM-like x3 (9.3 Gb, the headline set): size-independent properties of the whole 13 GB stream
        (tools/samcheck.c: one primary record per read in input order, every CIGAR adds up to its read and stays inside its contig, ...), the records of the
        first 45 000 reads (more than three 300 Mb batches on four lanes) byte for byte, and the same set split over 2 ranks of minialign_amd.multi on cuda:0
        (shards by bytes, carried value settled over gloo, every rank writing its own records in rank order) identical to the single stream;
  (iii) the hg38-size reference x ONT-like reads (3.1 Gb) with -xont.1dsq: properties of the whole stream, the records of the first 20 000 reads byte for byte.

The reference runs (index files, then -t1 over the sample) go on in the background on host cores while the device maps.  Skipped where the compiled reference
did not travel with the snapshot (it does with gpurun; /root/reference itself is never read here)."""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')
REFBIN = os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')
GENSIM = os.path.join(M.ROOT, 'tools', 'gensim')
PARTS = 16

def _samcheck():
    exe = os.path.join(M.ROOT, 'tools', 'samcheck'); src = exe + '.c'
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src): subprocess.check_call(['gcc', '-O2', '-o', exe, src])
    return exe

def _generate(d, tag, genome, reads, rd_tag='rd'):
    """reference (made once per tag) + read set (16 parts side by side, then one file); returns (ref.fa, reads.fa)"""
    ref = os.path.join(d, tag + '_ref.fa'); rd = os.path.join(d, '%s_%s.fa' % (tag, rd_tag))
    if not os.path.exists(ref): M.gensim('genome', *genome, out=ref)
    seed, depth, kind = reads
    procs = []
    for p in range(PARTS):
        f = open('%s.%02d' % (rd, p), 'wb')
        procs.append((subprocess.Popen([GENSIM, 'reads', str(seed), ref, str(depth), kind, 'fa', '20000', '2000', str(p), str(PARTS)], stdout=f), f))
    for pr, f in procs:
        assert pr.wait() == 0; f.close()
    with open(rd, 'wb') as g:
        for p in range(PARTS):
            with open('%s.%02d' % (rd, p), 'rb') as f: shutil.copyfileobj(f, g, 64 << 20)
            os.unlink('%s.%02d' % (rd, p))
    return ref, rd

def _head_fasta(rd, n, out):
    k = 0
    with open(rd, 'rb') as f, open(out, 'wb') as g:
        for line in f:
            if line.startswith(b'>'):
                k += 1
                if k > n: break
            g.write(line)
    return out

def _reference_in_background(preset, ref, sample, out, threads=32):
    """index file with `threads` threads, then the sample at -t1; returns the Popen (records land in `out`, header stripped later)"""
    mai = out + '.mai'
    cmd = '%s -x%s -t%d -d %s %s 2> %s.idx.err && %s -x%s -t1 %s %s > %s 2> %s.err; rc=$?; rm -f %s; exit $rc' % (REFBIN, preset, threads, mai, ref, out, REFBIN, preset, mai, sample, out, out, mai)
    return subprocess.Popen(['bash', '-c', cmd])

def _md5_records(path):
    h = hashlib.md5(); n = 0
    with open(path, 'rb') as f:
        for line in f:
            if not line.startswith(b'@'): h.update(line); n += 1
    return h.hexdigest(), n

def _map_through_samcheck(cmd, rd, n_head, head_out, env=None, cwd=None, timeout=900):
    """cmd's standard output through tools/samcheck; returns (summary dict, stderr of cmd, seconds)"""
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=cwd)
    c = subprocess.Popen([_samcheck(), rd, str(n_head), head_out], stdin=p.stdout, stdout=subprocess.PIPE)
    p.stdout.close()
    out, _ = c.communicate(timeout=timeout); err = p.stderr.read(); rc = p.wait(timeout=60)
    assert rc == 0, err.decode()[-3000:]
    return json.loads(out.decode().strip().splitlines()[-1]), err, time.time() - t0

@pytest.fixture(scope='module')
def work():
    if not os.path.exists(REFBIN): pytest.skip('oracle/_ref not built')
    d = tempfile.mkdtemp(prefix='mmheadline_')
    yield d
    shutil.rmtree(d, ignore_errors=True)

def test_dm6_size_x20_whole_sam_equals_the_reference(work):
    ref, rd = _generate(work, 'dm6', (0x5eed0001, 143700000, 1870, 0.05), (0x5eed0002, 20.0, 'pacbio'))
    want = os.path.join(work, 'dm6_ref.sam')
    bg = _reference_in_background('pacbio', ref, rd, want)
    s, err, sec = _map_through_samcheck([CLI, '-xpacbio', ref, rd], rd, 1 << 30, os.path.join(work, 'dm6_ours.sam'))
    assert s['error'] == '' and s['reads'] == s['primary'] and s['mapped'] > 0.98 * s['reads'], s
    assert s['bases_mapped'] > 2.7e9, s                                             # the full x20 set (2.87 Gb), not a sample
    assert bg.wait(timeout=900) == 0, open(want + '.err').read()[-2000:]
    got = _md5_records(os.path.join(work, 'dm6_ours.sam')); ref_md5 = _md5_records(want)
    assert got == ref_md5, 'dm6-size x20: SAM differs from the compiled reference'
    assert got[1] == s['records']
    # the same set over EIGHT ranks (all on cuda:0; MM_LANES=1 each): 1 870 contigs, so the carried value differs at nearly every shard boundary -- checks, window re-maps
    # and the rank-after-rank writers all have work -- and the stream must still be the single stream's
    env = dict(os.environ, MM_MULTI_SAME_DEVICE='1', MM_SLAB_GB='6', MM_LANES='1', MM_HOST_THREADS='24', PYTHONPATH=M.ROOT)
    port = 29900 + os.getpid() % 1000
    s8, err8, sec8 = _map_through_samcheck([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port', str(port),
                                            '-m', 'minialign_amd.multi', '-xpacbio', ref, rd], rd, 0, os.devnull, env=env, cwd=M.ROOT, timeout=1200)
    assert s8['error'] == '' and s8['digest'] == s['digest'] and s8['records'] == s['records'] and s8['bytes'] == s['bytes'], (s, s8, err8.decode()[-1500:])
    sys.stderr.write('[headline] dm6-size x20: single stream %.1f s, eight ranks on one GPU %.1f s (index builds included)\n' % (sec, sec8))
    for f in ('dm6_ours.sam', 'dm6_ref.sam', 'dm6_rd.fa', 'dm6_ref.fa'): os.unlink(os.path.join(work, f))

@pytest.fixture(scope='module')
def hg38(work):
    """the headline set and, running in the background, the reference's records for its first 45 000 reads (and for the first 20 000 of the ONT-like set)"""
    genome = (0x5eed0001, 3100000000, 25, 0.05)
    ref, rd = _generate(work, 'hg38', genome, (0x5eed0002, 3.0, 'pacbio'))
    pb_sample = _head_fasta(rd, 45000, os.path.join(work, 'pb_sample.fa'))
    bg_pb = _reference_in_background('pacbio', ref, pb_sample, os.path.join(work, 'pb_ref.sam'))
    _, ont_rd = _generate(work, 'hg38', genome, (0x5eed0003, 1.0, 'ont'), rd_tag='ont')
    ont_sample = _head_fasta(ont_rd, 20000, os.path.join(work, 'ont_sample.fa'))
    bg_ont = _reference_in_background('ont.1dsq', ref, ont_sample, os.path.join(work, 'ont_ref.sam'))
    yield dict(ref=ref, rd=rd, ont=ont_rd, bg_pb=bg_pb, bg_ont=bg_ont)
    for b in (bg_pb, bg_ont):
        if b.poll() is None: b.kill()

def test_hg38_size_x3_properties_head_parity_and_two_ranks(work, hg38):
    ref, rd = hg38['ref'], hg38['rd']
    head = os.path.join(work, 'pb_head.sam')
    s, err, sec = _map_through_samcheck([CLI, '-xpacbio', ref, rd], rd, 45000, head)
    assert s['error'] == '' and s['reads'] == s['primary'] and s['contigs'] == 25, s
    assert s['bases_mapped'] > 8.8e9 and s['mapped'] > 0.98 * s['reads'] and s['bytes'] > 12e9, s          # the whole 9.3 Gb set
    # the same set over two ranks (both on cuda:0): byte shards, carried value settled, records written rank after rank == the single stream
    env = dict(os.environ, MM_MULTI_SAME_DEVICE='1', MM_SLAB_GB='24', MM_LANES='2', MM_HOST_THREADS='48', PYTHONPATH=M.ROOT)
    port = 29700 + os.getpid() % 1500
    s2, err2, sec2 = _map_through_samcheck([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
                                            '-m', 'minialign_amd.multi', '-xpacbio', ref, rd], rd, 0, os.devnull, env=env, cwd=M.ROOT, timeout=1200)
    assert s2['error'] == '' and s2['digest'] == s['digest'] and s2['records'] == s['records'] and s2['bytes'] == s['bytes'], (s, s2, err2.decode()[-1500:])
    # the first 45 000 reads against the compiled reference at -t1
    assert hg38['bg_pb'].wait(timeout=900) == 0, open(os.path.join(work, 'pb_ref.sam.err')).read()[-2000:]
    got = _md5_records(head); want = _md5_records(os.path.join(work, 'pb_ref.sam'))
    assert got == want and got[1] >= 45000, 'hg38-size x3: the records of the first 45 000 reads differ from the compiled reference'
    sys.stderr.write('[headline] hg38-size x3: single stream %.1f s, two ranks %.1f s (index builds included)\n' % (sec, sec2))

def test_hg38_size_ont_like_properties_and_head_parity(work, hg38):
    ref, rd = hg38['ref'], hg38['ont']
    head = os.path.join(work, 'ont_head.sam')
    s, err, sec = _map_through_samcheck([CLI, '-xont.1dsq', ref, rd], rd, 20000, head)
    assert s['error'] == '' and s['reads'] == s['primary'] and s['contigs'] == 25, s
    assert s['bases_mapped'] > 2.5e9, s                                             # the whole 3.1 Gb set
    assert hg38['bg_ont'].wait(timeout=900) == 0, open(os.path.join(work, 'ont_ref.sam.err')).read()[-2000:]
    got = _md5_records(head); want = _md5_records(os.path.join(work, 'ont_ref.sam'))
    assert got == want and got[1] >= 20000, 'hg38-size ONT-like set: the records of the first 20 000 reads differ from the compiled reference'
