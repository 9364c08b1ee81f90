#!/usr/bin/env python3
"""What the lanes of the streaming engine did, batch by batch, from the stderr of a run with MM_VERBOSE=1 (last engine run in the file): per batch the duration and
end time of pack + upload, run (K1 .. K3 with the predicted carried value), wait for the carried value + verification, wait for the writer, D2H.
Usage: tools/lane_trace.py <stderr file>"""
import re, sys, collections
ev = []
for l in open(sys.argv[1]):
    m = re.match(r'\[minialign_amd\] batch (\d+) \((?:device \d+ )?lane (\d+)\): (.*?) ([\d.]+) ms \(at ([\d.]+)\)', l)
    if m: ev.append((int(m.group(1)), int(m.group(2)), m.group(3), float(m.group(4)), float(m.group(5))))
idx = max(i for i, e in enumerate(ev) if e[0] == 0 and e[2].startswith('pack'))
run = ev[idx:]; by = {}
for k, li, what, ms, at in run: by.setdefault(k, {'lane': li})[what.split()[0]] = (ms, at)
print('# batch lane | <stage> <ms>@<end, ms since the engine started>')
for k in sorted(by):
    b = by[k]; print('%3d %d  ' % (k, b['lane']) + '  '.join('%s %4.0f@%-5.0f' % (w, b[w][0], b[w][1]) for w in ('pack', 'run', 'carry', 'wait', 'D2H') if w in b))
tot = collections.Counter()
for k, li, what, ms, at in run: tot[what.split()[0]] += ms
print('# lane time summed: ' + ', '.join('%s %.0f ms' % kv for kv in tot.items()))
