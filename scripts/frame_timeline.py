"""Per-kernel timeline of the one-call frame from a rocprofv3 kernel trace of scripts/gpu_frame_trace.py: for every kernel of a frame
(k_ingest of the features' arena ... k_finish_sqrt of the object solve) the median start and end, microseconds after the frame's first
kernel, and how much of the objects' compression ran while the feature update was still busy.
usage: python scripts/frame_timeline.py <kernel_trace.csv> <out.json>"""
import csv, json, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('void orcvio_amd::', '').replace('orcvio_amd::', '').split('(')[0]
ev = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# a frame starts at a k_ingest that is followed (within a few kernels) by k_front and contains k_object_rows_batch
frames, cur = [], None
for name, a, b in ev:
    if name == 'k_potrf_reg<16>' or name == 'k_fac_flip':   # the prefactorisation between frames
        if cur: frames.append(cur)
        cur = None
        continue
    if cur is None:
        if name != 'k_ingest': continue
        cur = []
    cur.append((name, a, b))
if cur: frames.append(cur)
frames = [f for f in frames if any(n in ('k_object_rows_batch', 'k_obj_fused') for n, _, _ in f) and any(n.startswith('k_front') for n, _, _ in f)][5:]
sig = max(set(tuple(n for n, _, _ in f) for f in frames), key=lambda s: sum(1 for f in frames if tuple(n for n, _, _ in f) == s))
sel = [f for f in frames if tuple(n for n, _, _ in f) == sig]
out = dict(frames=len(sel), kernels=[])
for i, name in enumerate(sig):
    st = statistics.median((f[i][1] - f[0][1]) / 1e3 for f in sel)
    en = statistics.median((f[i][2] - f[0][1]) / 1e3 for f in sel)
    out['kernels'].append(dict(kernel=name, start_us=round(st, 2), end_us=round(en, 2)))
names = list(sig)
feat_end = next(k['end_us'] for k in out['kernels'] if k['kernel'] == 'k_epilogue')
obj = [k for k in out['kernels'] if k['kernel'] in ('k_object_rows_batch', 'k_obj_front', 'k_obj_fused', 'k_gemm_objA') or k['kernel'].startswith('k_obj_border')]
gemmA = None
# the k_gemm that follows the border kernel is A' = sum B - Y^T Y
for i, k in enumerate(out['kernels']):
    if k['kernel'].startswith('k_obj_border'):
        gemmA = out['kernels'][i + 1]
    if k['kernel'] == 'k_gemm_objA':   # (round 5: the one-launch compression is followed by its own product kernel)
        gemmA = k
comp_start = min(k['start_us'] for k in obj)
comp_end = gemmA['end_us'] if gemmA else max(k['end_us'] for k in obj)
out['feature_half_done_us'] = feat_end
out['object_compression_us'] = [comp_start, comp_end]
out['compression_overlapped_with_the_feature_update_us'] = round(max(0.0, min(comp_end, feat_end) - comp_start), 2)
out['frame_device_span_us'] = out['kernels'][-1]['end_us']
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}))
for k in out['kernels']: print('%-40s %8.2f %8.2f' % (k['kernel'], k['start_us'], k['end_us']))
