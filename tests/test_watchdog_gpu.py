"""GPU: the watchdog of the extension launches (mm_host.hip: k3_watchdog_main).  The test hook makes one wave of the first launch wait for something that never
comes (MM_TEST_K3_HANG = its place in the work list).  The launch must be called off within the deadline, the census must name the wave and what it waits for, the batch
must run again in the safe mode, and the records must be the bytes of an undisturbed run -- through the command line (the streaming engine, several lanes) and through the
per-batch entry (mm_align_batch via the small-set path of the command line with one lane)."""
import os, subprocess, tempfile
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')

def _run(args, env_extra, timeout=300):
    env = dict(os.environ); env.update(env_extra); env.setdefault('MM_DEVICES', '1')
    r = subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
    body = b''.join(l for l in r.stdout.splitlines(True) if not l.startswith(b'@PG'))
    return r.returncode, body, r.stderr.decode(errors='replace')

@pytest.fixture(scope='module')
def small_set():
    with tempfile.TemporaryDirectory(prefix='mmwd_') as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'reads.fa')
        M.gensim('genome', 9101, 3000000, 6, 0.30, out=ref); M.gensim('reads', 9102, ref, 4.0, 'pacbio', 'fa', 6000, 2500, out=rd)
        rc, want, err = _run(['-xpacbio', ref, rd], {})
        assert rc == 0 and want.count(b'\n') > 1000, err[-2000:]
        assert 'watchdog' not in err, err[-3000:]
        yield ref, rd, want

@pytest.mark.parametrize('lanes,batch', [(4, 2000000), (1, 0), (3, 700000)])
def test_a_launch_that_does_not_end_is_called_off_and_the_batches_run_again(small_set, lanes, batch):
    ref, rd, want = small_set
    env = {'MM_TEST_K3_HANG': '7', 'MM_K3_WATCHDOG_MS': '1500', 'MM_LANES': str(lanes)}
    if batch: env['MM_BATCH_BASES'] = str(batch)
    rc, got, err = _run(['-xpacbio', ref, rd], env)
    assert rc == 0, err[-3000:]
    assert 'watchdog: an extension launch' in err and 'TEST HOOK' in err and 'safe mode' in err, err[-3000:]
    assert 'giving up' not in err
    assert got == want, 'records differ after a launch was called off (%d against %d bytes)' % (len(got), len(want))

def test_the_watchdog_stays_silent_and_costs_nothing_visible(small_set):
    ref, rd, want = small_set
    rc, got, err = _run(['-xpacbio', ref, rd], {'MM_K3_WATCHDOG_MS': '30000', 'MM_LANES': '4', 'MM_BATCH_BASES': '1500000'})
    assert rc == 0 and got == want and 'watchdog' not in err, err[-2000:]
