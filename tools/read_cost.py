#!/usr/bin/env python3
"""Summary of a MM_DUMP_READ_COST file (one batch; mm_host.hip:batch_fetch): the per-read cost of mm_extend_kernel, what the heaviest reads are, and how far the
heaviest is from a wave's fair share of the launch.  Columns of the file: read, length, seeds, chains found, wave ticks (profiling build only), DP vectors of the owning
wave, fill ticks, trace ticks, chains that pass the length test, their summed length, chains walked, trials, trials taken from a job, alignments, chain jobs."""
import sys
rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1]) if l.strip()]
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n = len(rows); vec = sorted(r[5] for r in rows); tot = sum(vec); bases = sum(r[1] for r in rows)
print('reads %d, bases %d, DP vectors %d (mean %.0f per read = %.2f per base)' % (n, bases, tot, tot / max(1, n), tot / max(1, bases)))
for p in (50, 90, 99, 99.9, 100):
    v = vec[min(n - 1, int(n * p / 100))]; print('  percentile %5.1f: %9d vectors' % (p, v))
share = tot / waves
print('per-wave share of the launch at %d waves: %.0f vectors; heaviest read: %d = %.1f x that' % (waves, share, vec[-1], vec[-1] / max(1.0, share)))
print('%8s %8s %6s %8s %7s %6s %6s %5s %10s %10s' % ('read', 'length', 'n_pass', 'w_pass', 'chains', 'trials', 'jobs', 'alns', 'vectors', 'vec/base'))
for r in sorted(rows, key=lambda r: -r[5])[:16]:
    print('%8d %8d %6d %8d %7d %6d %6d %5d %10d %10.1f' % (r[0], r[1], r[8], r[9], r[10], r[11], r[12], r[13], r[5], r[5] / max(1, r[1])))
# by length class
for lo, hi in ((0, 32768), (32768, 65536), (65536, 131072), (131072, 262144), (262144, 1 << 30)):
    sel = [r for r in rows if lo <= r[1] < hi]
    if sel: print('length %7d..%-9d: %6d reads, %5.1f %% of the bases, %5.1f %% of the vectors, %.2f vectors per base, %.2f trials per read' % (lo, hi, len(sel), 100.0 * sum(r[1] for r in sel) / bases, 100.0 * sum(r[5] for r in sel) / max(1, tot), sum(r[5] for r in sel) / max(1, sum(r[1] for r in sel)), sum(r[11] for r in sel) / len(sel)))

if any(r[4] for r in rows):
    # profiling build: wave ticks (s_memtime, 100 MHz) a read held its wave for, with the ticks inside the fills and the tracebacks
    tk = sorted(rows, key=lambda r: -r[4])
    print('\nby wave time (shader clock cycles at 2.1 GHz; total %d ms over all reads):' % (sum(r[4] for r in rows) / 2.1e6))
    CLK = 2.1e6          # cycles per ms (s_memtime runs at the shader clock)
    t00 = min(r[15] for r in rows if r[4]) if len(rows[0]) > 15 else 0
    print('%8s %8s %6s %6s %6s %5s %10s %9s %9s %9s %9s %9s %9s' % ('read', 'length', 'chains', 'trials', 'jobs', 'alns', 'vectors', 'ms', 'fill ms', 'trace ms', 'wait ms', 'start ms', 'end ms'))
    def line(r):
        st = ((r[15] - t00) & 0xffffffff) * 256 / CLK if len(r) > 15 else 0.0
        print('%8d %8d %6d %6d %6d %5d %10d %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f' % (r[0], r[1], r[10], r[11], r[12], r[13], r[5], r[4] / CLK, r[6] / CLK, r[7] / CLK, (r[16] / CLK) if len(r) > 16 else 0.0, st, st + r[4] / CLK))
    for r in tk[:12]: line(r)
    if len(rows[0]) > 15:
        print('the reads that end last:')
        for r in sorted(rows, key=lambda r: -(((r[15] - t00) & 0xffffffff) * 256 + r[4]))[:12]: line(r)
        w = sum(r[16] for r in rows); print('waiting for a workspace: %.0f ms summed over the reads (%.1f %% of the wave time of the reads)' % (w / CLK, 100.0 * w / max(1, sum(r[4] for r in rows))))
