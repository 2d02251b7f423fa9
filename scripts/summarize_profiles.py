"""Turn the raw rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/<tag>_*) into the committed summaries under
profiles/: <tag>_kernel_stats.csv (per-kernel duration statistics), <tag>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE per
dispatch, median, KB), <tag>_bench.json."""
import csv, glob, json, os, shutil, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out')
dst = os.environ.get('PROFILES_DST') or os.path.join(ROOT, 'profiles')   # (on the GPU box: a directory under gpurun_out/, the raw traces stay there)
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.replace('void orcvio_amd::', '').replace('orcvio_amd::', '')
    return name.split('(')[0]


stats = find(f'{tag}_trace/**/*kernel_stats.csv')
if stats:
    shutil.copy(stats, os.path.join(dst, f'{tag}_kernel_stats.csv'))
    print('kernel stats ->', f'profiles/{tag}_kernel_stats.csv')
else:   # no --stats table: build it from the kernel trace
    trace = find(f'{tag}_trace/**/*kernel_trace.csv')
    if trace:
        dur = {}
        for row in csv.DictReader(open(trace)):
            dur.setdefault(row['Kernel_Name'], []).append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
        tot = sum(sum(v) for v in dur.values())
        with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 2), min(v), max(v),
                            round(statistics.pstdev(v), 1)])
        print('kernel stats (from trace) ->', f'profiles/{tag}_kernel_stats.csv')

traffic = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = find(f'{tag}_pmc_{counter}/**/*counter_collection.csv')
    if not f:
        continue
    per = {}
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] != counter:
            continue
        per.setdefault(short(row['Kernel_Name']), []).append(float(row['Counter_Value']))
    for k, v in per.items():
        if k.startswith('__amd'):
            continue
        traffic.setdefault(k, {})[f'{counter}_KB_median'] = statistics.median(v)
        traffic[k]['dispatches'] = len(v)
f = find(f'{tag}_pmc_MFMA/**/*counter_collection.csv')
if f:
    per = {}
    for row in csv.DictReader(open(f)):
        per.setdefault((short(row['Kernel_Name']), row['Counter_Name']), []).append(float(row['Counter_Value']))
    for (k, c), v in per.items():
        if k.startswith('__amd'):
            continue
        traffic.setdefault(k, {})[f'{c}_median'] = statistics.median(v)
# calibration of FETCH_SIZE / WRITE_SIZE on THIS access pattern (MI355X_MICROARCH.md: only 16 B-per-lane streams are calibrated):
# k_fac_commit and k_cov_remove read and write a known number of bytes with 8 B per lane, coalesced
calib = {}
for k, known in (('k_fac_commit', None), ('k_gram_reduce', None)):
    if k in traffic:
        calib[k] = {kk: vv for kk, vv in traffic[k].items() if kk.endswith('_KB_median')}
def build_tag():
    """sha256 (first 16 hex digits) of the library the counters were taken on: bench.py quotes it beside `roofline.traffic`."""
    import hashlib
    so = os.path.join(ROOT, 'orcvio_amd', 'lib', 'liborcvio_msckf.so')
    try:
        sys.path.insert(0, ROOT)
        from orcvio_amd import build as _b
        return dict(tag=tag, liborcvio_msckf_sha16=hashlib.sha256(open(so, 'rb').read()).hexdigest()[:16], source_sha16=_b.source_sha16())
    except OSError:
        return dict(tag=tag)


if traffic:
    json.dump({'build': build_tag(), 'calibration_kernels': calib, 'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains) around bench.py --steps 20; '
                       'KB per dispatch, median over the dispatches of each kernel. gfx950 caveat (MI355X_MICROARCH.md): '
                       'FETCH_SIZE under-reports wide coalesced reads by 2x; these kernels read 8 B per lane. SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = '
                       'FP64 flops executed on the matrix cores per dispatch; SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs.',
               'kernels': traffic}, open(os.path.join(dst, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
    print('pmc traffic ->', f'profiles/{tag}_pmc_traffic.json')
b = os.path.join(src, f'{tag}_bench.json')
if os.path.exists(b) and os.path.getsize(b) > 10:
    shutil.copy(b, os.path.join(dst, f'{tag}_bench.json'))
    print('bench ->', f'profiles/{tag}_bench.json')
    # the per-configuration table of the same line on its own (<tag>_configs.json): configs, the config-3 frame legs, the config-1 stream
    try:
        d = os.path.join(src, f'{tag}_bench_detail.json')   # the side measurements live in the detail file since round 5
        if os.path.exists(d):
            shutil.copy(d, os.path.join(dst, f'{tag}_bench_detail.json'))
            print('bench detail ->', f'profiles/{tag}_bench_detail.json')
        line = json.load(open(d if os.path.exists(d) else b))
        if line.get('configs'):
            obj = line.get('objects_update') or {}
            frames = {k: obj[k] for k in ('frame_config3', 'frame_config3_prefactored', 'frame_config3_one_call', 'frame_config3_one_call_prefactored') if k in obj}
            json.dump({'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'latency': line.get('latency'),
                       'configs': line['configs'], 'config3_frame_legs': frames, 'stream_config1': line.get('stream_config1'), 'stream_config5': line.get('stream_config5'),
                       'sequential_updates_per_s': line.get('sequential_updates_per_s'), 'block_ms_per_step': line.get('block_ms_per_step'),
                       'cpu_baseline': line.get('cpu_baseline')}, open(os.path.join(dst, f'{tag}_configs.json'), 'w'), indent=1)
            print('configs ->', f'profiles/{tag}_configs.json')
    except Exception as e:
        print('configs: skipped', e)
