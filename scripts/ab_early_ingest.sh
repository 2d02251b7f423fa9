for i in 1 2; do for m in 1 0; do ORCVIO_EARLY_INGEST=$m python bench.py --no-configs --no-cpu-baseline --latency-updates 400 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('early=$m', {k:round(v['median_ms'],4) for k,v in d['latency'].items()}, round(d['objects_update']['frame_config3_one_call']['median_ms'],4))"; done; done
