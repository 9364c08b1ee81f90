#!/usr/bin/env python3
"""How the kernels of a bench run share the device: from a rocprofv3 --kernel-trace CSV, per kernel kind the summed and the merged (union) busy time inside the
timed step (the second half of the sketch launches), and how often which kinds run side by side.  Usage: tools/trace_concurrency.py <..._kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sk = [r for r in rows if 'sketch' in r['Kernel_Name']]
mid = int(sk[len(sk) // 2]['Start_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) >= mid]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
def kind(r):
    n = r['Kernel_Name']; g = int(r['Grid_Size_X'])
    if 'sketch' in n: return 'K1 sketch+lookup'
    if 'mm_sort_kernel' in n: return 'K2s sort'
    if 'chain_scan' in n: return 'K2p window scans'
    if 'mm_chain_kernel' in n: return 'K2c chain sweep'
    if 'sort_chain_lds' in n: return 'K2a (large reads)'
    if 'sort_chain_kernel' in n: return 'K2 serial (rescue rounds)'
    if 'extend' in n: return 'K3 extend (round 0)' if g > 100000 else 'K3 extend (rescue rounds)'
    return 'copies'
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
kinds = {}
for r in rows: kinds.setdefault(kind(r), []).append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
print('timed step: %.1f ms, %d kernel launches' % ((t1 - t0) / 1e6, len(rows)))
for k, iv in sorted(kinds.items()):
    print('%-28s n=%5d  summed %8.1f ms  mean %7.2f ms  busy (union) %8.1f ms' % (k, len(iv), sum(e - s for s, e in iv) / 1e6, sum(e - s for s, e in iv) / 1e6 / len(iv), union(iv) / 1e6))
print('any kernel: %.1f ms' % (union([x for iv in kinds.values() for x in iv]) / 1e6))
ev = []
for k, iv in kinds.items():
    for s, e in iv: ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort(); cur = {}; last = ev[0][0]; hist = {}
for t, d, k in ev:
    key = (cur.get('K3 extend (round 0)', 0), 1 if cur.get('K2c chain sweep', 0) > 0 else 0)
    hist[key] = hist.get(key, 0) + (t - last); last = t
    cur[k] = cur.get(k, 0) + d
tot = sum(hist.values())
for key, v in sorted(hist.items(), key=lambda x: -x[1])[:8]:
    print('  %d x K3 round 0 in flight, chain sweep %s: %6.0f ms (%4.1f %%)' % (key[0], 'running' if key[1] else 'idle   ', v / 1e6, 100 * v / tot))
