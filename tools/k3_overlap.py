#!/usr/bin/env python3
"""k3_overlap.py -- from the MM_VERBOSE log of a stream (bench.py 2> log): per step, how many extension launches were in flight for how long, and where a lane's
time per batch goes (kernels, the host's turns between them, upload, waiting for the carried value, D2H).  usage: k3_overlap.py log [steps = 3]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines(); last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = []; cur = []
for l in lines:
    if 'batch 0 (device' in l and 'taken' in l:
        if cur: steps.append(cur)
        cur = []
    cur.append(l)
steps.append(cur)
for st in steps[-last:]:
    iv = []; agg = collections.defaultdict(float); n = 0
    for l in st:
        m = re.search(r'batch (\d+) \(device (\d+) lane (\d+)\): run ([\d.]+) ms \(at ([\d.]+)\): sketch ([\d.]+), sort \+ chain ([\d.]+), extension ([\d.]+)', l)
        if m:
            end = float(m.group(5)); z = float(m.group(8)); iv.append((end - z - 2, end - 2)); n += 1
            agg['run'] += float(m.group(4)); agg['sketch'] += float(m.group(6)); agg['sort + chain'] += float(m.group(7)); agg['extension'] += z
        m = re.search(r'\): (pack \+ upload|carry wait \+ verify|D2H) ([\d.]+) ms', l)
        if m: agg[m.group(1)] += float(m.group(2))
    if not iv: continue
    ev = []
    for s, e in iv: ev += [(s, 1), (e, -1)]
    ev.sort(); d = 0; t0 = ev[0][0]; hist = collections.defaultdict(float)
    for t, x in ev: hist[d] += t - t0; t0 = t; d += x
    end_all = max(e for s, e in iv)
    print('step of %d batches: extension launches in flight 0 / 1 / 2 / 3 / 4+ for %s ms between the first launch (at %.0f ms) and the end of the last (at %.0f ms)' % (
        n, ' / '.join('%.0f' % hist.get(k, 0) if k < 4 else '%.0f' % sum(v for q, v in hist.items() if q >= 4) for k in range(5)), iv and min(s for s, e in iv), end_all))
    print('  per batch on its lane: pack + upload %.0f, sketch %.0f, sort + chain %.0f, extension %.0f, the host\'s turns between the kernels %.0f, carried-value wait + verify %.0f, D2H %.0f ms' % (
        agg['pack + upload'] / n, agg['sketch'] / n, agg['sort + chain'] / n, agg['extension'] / n, (agg['run'] - agg['sketch'] - agg['sort + chain'] - agg['extension']) / n, agg['carry wait + verify'] / n, agg['D2H'] / n))
