#!/usr/bin/env python3
"""trial_stats.py -- what the extension trials of a read set cost and how far they could be spread (CPU analysis over the oracle's trial log).
usage: OM_DUMP_TRIALS=trials.txt oracle/ora_minialign -xpacbio ref.fa reads.fa > /dev/null; tools/trial_stats.py trials.txt
A line of the log: read, round, chain, trial of the chain, ns of the downward pass + maximum search, ns of the upward pass + traceback, outcome
(z = maximum 0, d = duplicate, s = score too low / no path, r = recorded, R = recorded and the chain's walk ends)."""
import sys, collections
reads = collections.defaultdict(list)
for l in open(sys.argv[1]):
    f = l.split('\t'); reads[int(f[0])].append((int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), f[6].strip()))
tot = sum(t[3] + t[4] for r in reads.values() for t in r)
per = sorted(((sum(t[3] + t[4] for t in r), k) for k, r in reads.items()), reverse=True)
print('%d reads with trials, %d trials, %.2f s of trial time (the oracle, one thread)' % (len(reads), sum(len(r) for r in reads.values()), tot * 1e-9))
oc = collections.Counter(t[5] for r in reads.values() for t in r)
print('outcomes:', dict(oc), ' trials in round 0: %d, later rounds: %d' % (sum(1 for r in reads.values() for t in r if t[0] == 0), sum(1 for r in reads.values() for t in r if t[0] > 0)))
cum = 0
for frac in (0.01, 0.05, 0.10):
    n = max(1, int(len(per) * frac)); print('the heaviest %4.0f %% of the reads (%d): %.1f %% of the trial time' % (frac * 100, n, 100.0 * sum(p[0] for p in per[:n]) / tot))
med = per[len(per) // 2][0]
print('median read: %.2f ms; heaviest: %.1f ms (%.0f x the median)' % (med * 1e-6, per[0][0] * 1e-6, per[0][0] / max(1, med)))
print()
print('the ten heaviest reads: trial time, trials (chains), of them in later rounds; duplicates / low score / recorded; longest single trial; time if every first trial of a chain ran elsewhere')
for cost, k in per[:10]:
    r = reads[k]; chains = len(set((t[0], t[1]) for t in r)); late = sum(1 for t in r if t[0] > 0)
    longest = max(t[3] + t[4] for t in r)
    # spread: first trials of the chains run as jobs side by side (the owner consumes the results in order: free), later trials of a chain stay on the owner
    own = sum(t[3] + t[4] for t in r if t[2] > 0)
    print('  read %6d: %8.1f ms, %4d trials (%3d chains), %4d in later rounds; d %3d  s %3d  r %3d; longest %6.1f ms; spread: %6.1f ms (%.0f x)' % (
        k, cost * 1e-6, len(r), chains, late, sum(1 for t in r if t[5] == 'd'), sum(1 for t in r if t[5] == 's'), sum(1 for t in r if t[5] in 'rR'), longest * 1e-6, (own + longest) * 1e-6, cost / max(1, own + longest)))
# the whole set: critical path of a launch = the heaviest read; with the first trials spread = max over reads of (own + longest)
crit = per[0][0]; crit_spread = max(sum(t[3] + t[4] for t in r if t[2] > 0) + max(t[3] + t[4] for t in r) for r in reads.values())
print()
print('one wave per read: the launch lasts as long as its heaviest read, %.1f ms of %.2f s (%.0f waves of work); first trials spread: %.1f ms' % (crit * 1e-6, tot * 1e-9, tot / crit, crit_spread * 1e-6))
dup_up = sum(t[4] for r in reads.values() for t in r)
dup_down = sum(t[3] for r in reads.values() for t in r if t[5] == 'd')
print('work a speculative full trial adds: the upward pass + traceback of the trials that end as duplicates after the downward pass -- today they stop there: %d of %d trials, their downward passes are %.1f %% of the trial time' % (oc['d'], sum(oc.values()), 100.0 * dup_down / tot))
