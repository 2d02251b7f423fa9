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
