# A/B of the finish inside the solve launch (LaFin) on one box: bench.py's contract figures for each setting, alternating
# (2: every update with a look-ahead solve; 1: the chained frame call only -- the default; 0: never)
for v in 2 1 0 2 1 0; do
ORCVIO_FUSE_FINISH=$v timeout 600 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse=$v', {k: d.get(k) for k in ('value','ms_per_step','host_visible_ms','config3_frame_ms','config3_frame_unchained_ms','config3_object_update_ms')}, d['roofline']['frac'], d['roofline']['kernel_us'])"
done
