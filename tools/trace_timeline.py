#!/usr/bin/env python3
"""Per-queue timeline of the timed step of a bench run from a rocprofv3 --kernel-trace CSV: every launch over 1 ms (start, duration, kind, queue), and per 50 ms
slice how many extension launches are in flight.  Usage: tools/trace_timeline.py <..._kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
sk = [r for r in rows if 'sketch' in r['Kernel_Name']]; mid = int(sk[len(sk) // 2]['Start_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) >= mid]; t0 = int(rows[0]['Start_Timestamp'])
def kind(n):
    for k, v in (('sketch', 'K1'), ('mm_sort_kernel', 'K2s'), ('chain_scan', 'K2p'), ('mm_chain_kernel', 'K2c'), ('sort_chain', 'K2a'), ('extend', 'K3'), ('text_codes', 'K0'), ('codes_pack', 'K0')):
        if k in n: return v
    return 'copy'
qs = {}
for r in rows: qs.setdefault(r.get('Queue_Id', '?'), len(qs))
print('# start ms, duration ms, kind, queue; launches over 1 ms')
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 1.0: print('%9.1f %8.1f  %-4s q%-2d grid %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, d, kind(r['Kernel_Name']), qs[r.get('Queue_Id', '?')], r['Grid_Size_X']))
