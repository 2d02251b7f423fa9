"""bench.py's launcher logic, on CPU: `--gpus N` without a launcher must try to start N ranks (and say so when the node
has fewer GPUs) instead of silently running one (VERDICT r1 / ADVICE r1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_is_honoured_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0
        assert '--gpus 2' in p.stderr and 'GPU(s)' in p.stderr


def _canned_detail():
    """A full detail record of a real run (round 4's 20.5 KB line, which the round driver could not parse)."""
    import json
    path = os.path.join(ROOT, 'profiles', 'r4_final_bench_driver_style.json')
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_contract_line_is_compact_and_parses_from_the_tail_of_stdout(tmp_path, capsys, monkeypatch):
    """the LAST stdout line must be a JSON object < 4 KB carrying metric / value / ms_per_step / roofline /
    cpu_baseline, parseable from the last 8 KB of stdout whatever was printed before it."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    out = _canned_detail()
    out['roofline']['traffic_source'] = dict(file='profiles/r4b_pmc_traffic.json', build=None, measured_in_this_run=False)
    out['roofline']['whole_step_executed_frac'] = 0.027
    out['roofline']['kernel_us'] = 40.4
    block_dt = [v * 1e-3 * out['steps'] for v in out['block_ms_per_step']]
    line = bench.compact_line(out, block_dt, 'bench_detail.json')
    assert len(line) < 4096 and '\n' not in line
    got = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline', 'timed_blocks', 'timed_region_s', 'sequential_updates_per_s', 'host_visible_ms',
              'config3_frame_ms'):
        assert k in got, k
    assert got['value'] == out['value'] and got['ms_per_step'] == out['ms_per_step'] and got['steps'] == 20
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'kernel', 'whole_step_executed_frac')) <= set(got['roofline'])
    assert abs(got['roofline']['frac'] - got['roofline']['achieved'] / got['roofline']['peak']) < 1e-5
    assert set(('value', 'unit', 'cores', 'kind', 'sample', 'host_cores', 'all_cores')) <= set(got['cpu_baseline'])
    assert got['cpu_baseline']['all_cores']['cores'] == 32
    assert 'workload' in got['config'] and 'model' not in got['config']
    assert abs(got['timed_region_s'] - sum(block_dt)) < 1e-4
    assert got['config3_frame_ms'] == round(out['objects_update']['frame_config3_one_call']['median_ms'], 5)
    # the whole emission: detail to a file, the compact line last on stdout; parse it the way the driver must -- from the tail
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    print('some library banner ' * 800)   # 16 KB of noise in front
    bench.emit(out, block_dt)
    cap = capsys.readouterr()
    tail = cap.out[-8192:]
    last = json.loads(tail.strip().splitlines()[-1])
    assert 'roofline' in last and 'cpu_baseline' in last and last['value'] == out['value']
    detail = json.load(open(tmp_path / 'bench_detail.json'))
    assert detail['configs'] and detail['stream_config5'] and detail['latency']   # nothing lost: it moved
    assert last['detail'] == 'bench_detail.json'


def test_contract_line_stays_small_with_no_side_measurements_and_at_n_gt_1():
    """Ranks > 1 (no latency / objects / CPU legs) and the comm record: still the contract keys, still < 4 KB."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    out = _canned_detail()
    for k in ('objects_update', 'configs', 'stream_config1', 'stream_config5', 'cpu_baseline'):
        out[k] = None
    out['latency'] = {'device_resident': out['latency']['device_resident']}
    out['n_gpus'] = 2
    out['comm'] = dict(transport='rccl', world=2, ranks_seen=2, exchange_us=21.0, replicated_solve_us=57.0, model_us=126.0)
    got = json.loads(bench.compact_line(out, [0.002] * 6, None))
    assert got['comm']['world'] == 2 and got['cpu_baseline'] is None and got['host_visible_ms'] is None and got['n_gpus'] == 2
    assert len(json.dumps(got)) < 4096
