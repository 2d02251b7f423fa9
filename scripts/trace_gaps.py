"""Kernel-trace timeline of a repeating update: per kernel of the sequence, median duration and median gap to the previous
kernel's end (rocprofv3 --kernel-trace CSV).  usage: python scripts/trace_gaps.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_ingest'
short = lambda n: n.replace('void orcvio_amd::', '').replace('orcvio_amd::', '').split('(')[0]
seqs, cur = [], None
for r in rows:
    name = short(r['Kernel_Name'])
    if first in name:
        if cur: seqs.append(cur)
        cur = []
    if cur is not None:
        cur.append((name, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
if cur: seqs.append(cur)
from collections import Counter
shape = Counter(tuple(n for n, _, _ in s) for s in seqs).most_common(3)
for names, cnt in shape:
    sel = [s for s in seqs if tuple(n for n, _, _ in s) == names][5:]
    if not sel: continue
    print(f'--- sequence seen {cnt} times ({len(names)} kernels)')
    tot = []
    for i, nm in enumerate(names):
        dur = statistics.median(s[i][2] - s[i][1] for s in sel) / 1e3
        gap = statistics.median(s[i][1] - s[i - 1][2] for s in sel) / 1e3 if i else 0.0
        print(f'{nm:40s} gap {gap:7.2f} us   dur {dur:7.2f} us   ends at {statistics.median(s[i][2] - s[0][1] for s in sel) / 1e3:7.2f}')
    print('span first start -> last end: %.2f us' % statistics.median((s[-1][2] - s[0][1]) / 1e3 for s in sel))
    starts = [s[0][1] for s in sel]
    if len(starts) > 2: print('period (start to start, median): %.2f us' % statistics.median((b - a) / 1e3 for a, b in zip(starts, starts[1:])))
