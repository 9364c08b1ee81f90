"""GPU: the streaming engine (several batches in flight, carried reference length verified in batch order) and the sharded multi-process path
(minialign_amd/multi.py: one read set split over ranks, the carried value settled between them, records merged in rank order) must give the bytes the
single stream gives -- on many-contig references, the case where the carried value (minialign.c:3864) bites."""
import ctypes, os, subprocess, sys, tempfile
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')

def _strip_pg(sam): return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))
def _body(sam): return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@'))

@pytest.fixture(scope='module')
def data():
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', 7301, 3000000, 25, 0.08, out=ref)                          # 25 contigs of very different lengths
        M.gensim('reads', 7302, ref, 1.2, 'pacbio', 'fa', 5000, 2000, out=rd)       # about 700 reads
        want = subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        yield ref, rd, _strip_pg(want)

@pytest.mark.parametrize('lanes,batch', [(1, 400000), (3, 250000), (4, 60000)])
def test_batches_in_flight_give_the_single_stream(data, lanes, batch):
    """many small batches on 1 / 3 / 4 lanes: every batch runs ahead with a predicted carried value and is verified in order"""
    ref, rd, want = data
    env = dict(os.environ, MM_BATCH_BASES=str(batch), MM_LANES=str(lanes), MM_SLAB_GB='6')
    r = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _strip_pg(r.stdout) == want

@pytest.mark.parametrize('n_ctx,lanes,batch,chunk', [(2, 2, 250000, None), (3, 1, 90000, 200000), (4, 2, 60000, 65536), (4, 1, None, 1 << 20)])
def test_batches_dealt_over_device_contexts_give_the_single_stream(data, n_ctx, lanes, batch, chunk):
    """the command-line program over several device contexts in ONE process (mm_align_init spans the visible devices; here MM_DEVICE_CONTEXTS puts 2 / 3 / 4 of them
    on the one GPU of the box): pieces of the text go to the devices round robin, the scan finds the records in order, batches are dealt to device x lane, the carried
    value is verified in batch order across devices, one writer -- the bytes of the single stream"""
    ref, rd, want = data
    env = dict(os.environ, MM_DEVICE_CONTEXTS=str(n_ctx), MM_LANES=str(lanes), MM_SLAB_GB='4')
    if batch: env['MM_BATCH_BASES'] = str(batch)
    if chunk: env['MM_CHUNK_BYTES'] = str(chunk)
    r = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _strip_pg(r.stdout) == want

def test_device_contexts_through_the_c_abi(data):
    """mm_map_text (text in host memory -> sink in input order) and mm_map_reads (parsed reads, any batch to any device) over three device contexts"""
    from minialign_amd import multi
    ref, rd, want = data
    env_keep = {k: os.environ.get(k) for k in ('MM_DEVICE_CONTEXTS', 'MM_SLAB_GB', 'MM_BATCH_BASES')}
    os.environ.update(MM_DEVICE_CONTEXTS='3', MM_SLAB_GB='4', MM_BATCH_BASES='200000')
    try:
        L = multi.load_library(); assert L.mm_set_device(0) == 0
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 4)(b'minialign', b'-xpacbio', ref.encode(), rd.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        assert L.mm_align_devices(al) == 3
        text = open(rd, 'rb').read(); body = _body(want)
        sm = multi.ShardMapper(L, al, None, 0, 0, lanes=2, text=(ctypes.cast(ctypes.c_char_p(text), ctypes.c_void_p).value, len(text))).map(0)
        assert sm.col.text() == body
        reads = ctypes.c_void_p(L.mm_reads_load(rd.encode())); n = L.mm_reads_count(reads)
        sm = multi.ShardMapper(L, al, reads, 0, n, lanes=2).map(0)
        assert sm.col.text() == body
        L.mm_reads_free(reads); L.mm_align_destroy(al); L.mm_idx_destroy(mi)
    finally:
        for k, v in env_keep.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

@pytest.mark.parametrize('world', [2, 3])
def test_one_read_set_split_over_ranks_on_one_gpu(data, world):
    """world_size 2 and 3 with every rank on cuda:0 (gloo group): shards mapped with a guessed carried value, settled with the true one (checks, window
    re-maps), merged in rank order == the single stream"""
    ref, rd, want = data
    env = dict(os.environ, MM_MULTI_SAME_DEVICE='1', MM_SLAB_GB='4', MM_LANES='2', MM_BATCH_BASES='600000', PYTHONPATH=M.ROOT)
    port = 29600 + os.getpid() % 1500 + world
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1', '--master-port', str(port),
                        '-m', 'minialign_amd.multi', '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=M.ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert _body(r.stdout) == _body(want), r.stderr.decode()[-1500:]

def test_settling_a_wrong_guess_remaps_only_the_head(data):
    """a shard started with a deliberately wrong carried value: mm_carry_check names the first read that decides differently (or says that nothing does),
    the window re-map splices the new records in, and the text equals a run that started with the right value"""
    from minialign_amd import multi
    ref, rd, want = data
    os.environ.setdefault('MM_SLAB_GB', '6')
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 4)(b'minialign', b'-xpacbio', ref.encode(), rd.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
    mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
    reads = ctypes.c_void_p(L.mm_reads_load(rd.encode())); n = L.mm_reads_count(reads)
    body = _body(want)
    tried = 0
    for first in (0, 97, 211, 330):
        # the truth at `first`: what a stream over reads [0, first) leaves behind
        sm0 = multi.ShardMapper(L, al, reads, 0, first, lanes=2).map(0); truth = sm0.carry_out
        right = multi.ShardMapper(L, al, reads, first, n - first, lanes=2).map(truth).col.text()
        assert sm0.col.text() + right == body
        for guess in (0, 1, L.mm_idx_max_len(mi)):
            sm = multi.ShardMapper(L, al, reads, first, n - first, lanes=2, guess=guess).map()
            sm.settle(None, 0, 1, truth)
            assert sm.col.text() == right, (first, guess, sm.stats)
            tried += sm.stats['remapped_reads'] > 0
    assert tried > 0, 'no case exercised the window re-map'

def test_bench_line_with_two_ranks_on_one_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), here with both ranks on cuda:0 and a gloo group: one read set
    split over the ranks, the carried value settled, ONE JSON line from rank 0 with the strong-scaling fields, and the records of the first reads identical to
    the CPU reference / oracle (--check)"""
    import json
    env = dict(os.environ, MM_BENCH_SAME_DEVICE='1', MM_SLAB_GB='8', PYTHONPATH=M.ROOT)
    port = 29800 + os.getpid() % 1500
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(M.ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--workload', 'dm6', '--genome-len', '6000000', '--contigs', '40', '--depth', '8', '--lanes', '2', '--check', '--check-reads', '300', '--baseline-reads', '600'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=M.ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['value'] > 0 and d['unit'] == 'Gbases/s'
    assert d['sam_identical'] is True, d.get('sam_check')
    assert d['roofline']['achieved'] > 0 and 'cpu_baseline' not in d

@pytest.fixture(scope='module')
def repeat_rich():
    """750 reads, many with dozens of chains (the chain jobs of the default run have work), on a repeat-rich reference; the records the oracle gives (made once for all the schedules)"""
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', 7401, 1500000, 6, 0.45, out=ref); M.gensim('reads', 7402, ref, 3.0, 'pacbio', 'fa', 6000, 2500, out=rd)
        opts = ['-xpacbio', '-f0.2,0.05,0.002']
        want = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        yield ref, rd, opts, want

@pytest.mark.parametrize('env', [dict(MM_TEST_SPLIT='1'), dict(MM_K3_NO_JOBS='1'), dict(MM_K3_NO_RETRY_JOBS='1'), dict(MM_K3_NO_ROUND_JOBS='1'), dict(MM_K3_NO_JOBS='1', MM_K3_NO_RETRY_JOBS='1', MM_K3_NO_ROUND_JOBS='1'),
                                 dict(MM_HOST_INDEX='1'), dict(MM_K3_JOB_CAP='7'), dict(MM_DEVICE_CONTEXTS='2', MM_SLAB_GB='14'), dict(MM_NO_CARRY_DEPS='1'), dict(MM_HOST_CIGAR='1'), dict(MM_SLAB_GB='1', MM_LANES='3'),
                                 dict(MM_TEST_K3_HANG='40', MM_K3_WATCHDOG_MS='1500')],
                         ids=['batch-split-on-pool-exhaustion', 'no-chain-jobs', 'no-retry-jobs', 'later-round-chains-on-the-own-wave', 'no-jobs-of-any-kind',
                              'host-index', 'more-chain-jobs-than-slots', 'two-device-contexts', 'carried-value-by-prediction-only', 'cigar-strings-by-the-host', 'few-workspaces-three-lanes',
                              'a-launch-called-off-by-the-watchdog'])
def test_alternative_schedules_give_the_same_bytes(env, repeat_rich):
    """the forms kept behind environment switches -- the fallback for a batch the device pools cannot hold (its reads in halves, down to fewer than 8), the extension
    launch without its chain / retry / round jobs, without the carried value taken from its source inside the launch, the host's index build and CIGAR walk, several device
    contexts, a workspace budget so small that waves find none on offer, a launch that the watchdog has to call off -- on a repeat-rich set with a high seed threshold, where
    many reads need the rescue rounds (what round 5 still kept behind switches and measured slower is in HISTORY.md, not in the library)"""
    ref, rd, opts, want = repeat_rich
    r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **dict(dict(MM_SLAB_GB='32', MM_BATCH_BASES='3000000'), **env)), timeout=600)          # (32 GB: a workspace for every resident wave at this read length, so that the chain and retry jobs of the default schedule run)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _strip_pg(r.stdout) == want
    assert b're-run' in r.stderr

@pytest.mark.parametrize('fanout', ['1', '4'])
def test_index_replicas_are_copied_device_to_device(fanout):
    """the replica path of the N-device engine on a one-GPU box (MM_TEST_REPLICA): the contexts behind the first take their index and packed reference through
    idx_replica's copy path -- hipMalloc, hipMemcpyPeer from a holder (fan-out 1: a chain, each replica made from the one before; 4: side by side from the originals),
    adoption by the context, release with the index -- and map through the copies: the records are those of the single stream"""
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', 7411, 4000000, 25, 0.3, out=ref); M.gensim('reads', 7412, ref, 2.0, 'pacbio', 'fa', 5000, 2000, out=rd)
        opts = ['-xpacbio']
        want = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                           env=dict(os.environ, MM_DEVICE_CONTEXTS='4', MM_TEST_REPLICA='1', MM_REPLICA_FANOUT=fanout, MM_SLAB_GB='8', MM_LANES='2', MM_BATCH_BASES='400000', MM_VERBOSE='1'))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert _strip_pg(r.stdout) == want
        made = [l for l in r.stderr.decode().splitlines() if 'index replica on device' in l]
        assert len(made) == 3 and not any('FAILED' in l for l in made), made          # (test hook: a replica per context behind the first)
        if fanout == '1': assert any('(a replica)' in l for l in made), made          # with one copy per holder at a time the third is served by the first replica

def test_output_to_a_regular_file_is_written_at_its_offsets():
    """`minialign ... > out.sam`: where the standard output is a regular file the batches' text is written by several drain threads, each batch at the offset the sizes of the
    batches in front of it give (stream_map, pos_fd) -- the file must be what a pipe receives from the one ordered writer; many small batches over two device contexts, a header in
    front of the records (the stream's position when mapping starts), and a file opened for appending (which takes the ordered writer: pwrite ignores offsets there)"""
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa'); out = os.path.join(d, 'out.sam'); app = os.path.join(d, 'app.sam')
        M.gensim('genome', 7421, 3000000, 12, 0.2, out=ref); M.gensim('reads', 7422, ref, 3.0, 'pacbio', 'fa', 4000, 1500, out=rd)
        env = dict(os.environ, MM_DEVICE_CONTEXTS='2', MM_SLAB_GB='8', MM_LANES='2', MM_BATCH_BASES='300000')
        pipe = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert pipe.returncode == 0, pipe.stderr.decode()[-2000:]
        with open(out, 'wb') as f: r = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=f, stderr=subprocess.PIPE, env=dict(env, MM_VERBOSE='1'), timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert b'at its place in the file' in r.stderr          # the drain threads ran
        assert _strip_pg(open(out, 'rb').read()) == _strip_pg(pipe.stdout)
        with open(app, 'wb') as f: f.write(b'@CO\tin front\n')
        with open(app, 'ab') as f: r = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=f, stderr=subprocess.PIPE, env=dict(env, MM_VERBOSE='1'), timeout=600)
        assert r.returncode == 0 and b'at its place in the file' not in r.stderr
        assert _strip_pg(open(app, 'rb').read()) == b'@CO\tin front\n' + _strip_pg(pipe.stdout)

@pytest.fixture(scope='module')
def long_tailed():
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', 7501, 5000000, 3, 0.3, out=ref); M.gensim('reads', 7502, ref, 2.0, 'ont', 'fa', out=rd)          # 858 reads to 166 kb: 43 above 32 kb, 5 above 64 kb, 2 above 128 kb
        opts = ['-xont.1dsq', '-f0.2,0.05,0.002']
        want = _strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        yield ref, rd, opts, want

@pytest.mark.parametrize('env', [dict(), dict(MM_SLAB_GB='64'), dict(MM_LANES='4', MM_BATCH_BASES='2500000'), dict(MM_K3_NO_RETRY_JOBS='1', MM_K3_NO_JOBS='1'), dict(MM_ONE_SLAB_CLASS='1')],
                         ids=['four-workspaces-per-class-and-xcd', 'full-ladder', 'four-lanes-one-workspace-each', 'no-jobs', 'one-class'])
def test_workspace_ladder_with_jobs_gives_the_same_bytes(long_tailed, env):
    """a long-tailed read set on the ladder of DP workspace classes (32 k / 64 k / 128 k / longest), with so small a budget that the classes above the ordinary one have four
    workspaces per XCD, shared and the lanes' own: chain jobs and retry jobs take their workspaces without waiting (acquire), the work list is by class -- and the forms kept behind switches; a
    hang here is the failure the first version had (helpers waiting for workspaces held by the waves that waited for them)"""
    ref, rd, opts, want = long_tailed
    r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **dict(dict(MM_SLAB_GB='2', MM_LANES='2', MM_BATCH_BASES='6000000'), **env)), timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _strip_pg(r.stdout) == want
