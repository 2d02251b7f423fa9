# A/B on one box: U = [A; b^T] L_a inside k_front (ORCVIO_FRONT_U=1: opt-in) against the k_gemm_asmA launch behind it (0, the default), alternating
for v in 1 0 1 0; do
ORCVIO_FRONT_U=$v timeout 600 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('front_u=$v', {k: d.get(k) for k in ('value','ms_per_step','host_visible_ms','config3_frame_ms','config3_object_update_ms','configs_device_resident_ms')}, d['roofline']['frac'], d['roofline']['kernel_us'], d['roofline'].get('kernel_ms'))"
done
