"""From a rocprofv3 kernel trace (csv) of bench.py: the timeline of one replayed update in the steady state -- start and duration of every
launch relative to the step's first kernel, and the gaps between them (median over the timed steps).
usage: python scripts/step_timeline.py <kernel_trace.csv> [first kernel name prefix = k_front]"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_front'
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0].replace('orcvio_amd::', '').replace('void ', '')) for r in rows), key=lambda t: t[0])
steps, cur = [], None
for s, e, n in ks:
    if n.startswith(first):
        if cur: steps.append(cur)
        cur = []
    if cur is not None: cur.append((s, e, n))
if cur: steps.append(cur)
sig = max(set(tuple(n for _, _, n in sp) for sp in steps), key=lambda q: sum(1 for sp in steps if tuple(n for _, _, n in sp) == q))
sel = [sp for sp in steps if tuple(n for _, _, n in sp) == sig]
sel = sel[len(sel) // 2:]   # (the later half: the timed region)
print(f'{len(sel)} steps of {len(sig)} launches')
prev_end = None
for i, n in enumerate(sig):
    start = st.median(sp[i][0] - sp[0][0] for sp in sel) / 1e3
    dur = st.median(sp[i][1] - sp[i][0] for sp in sel) / 1e3
    gap = st.median(sp[i][0] - sp[i - 1][1] for sp in sel) / 1e3 if i else 0.0
    print(f'  {n:28s} start {start:7.2f} us  gap before {gap:6.2f}  duration {dur:6.2f}')
per = [b[0][0] - a[0][0] for a, b in zip(sel, sel[1:])]
print('step period (first kernel to first kernel): median %.2f us; last kernel end -> next first start: %.2f us' % (st.median(per) / 1e3, st.median(b[0][0] - a[-1][1] for a, b in zip(sel, sel[1:])) / 1e3))
