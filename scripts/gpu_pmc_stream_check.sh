#!/bin/bash
# On the GPU box: the C++ stream harness (one call per frame) under a counter pass of rocprofv3, which serialises dispatches -- the
# polled joins of the frame call's side stream can then never succeed; the call must notice once, repair that frame and go on at
# full speed.  Prints the wall time of the pass, the harness's own rate and its count of repaired updates.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/gpu_stream_bench.sh pmc0 2>&1 | grep '"step"' | cut -c1-220
cd /tmp && export TMPDIR=/tmp
S=$(date +%s.%N)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/orcvio_pmc_check -o p -- $R/gpurun_out/stream_pmc0/stream_bench --stream $R/gpurun_out/stream_pmc0/config1.bin --mode step --frames 200 --warmup 16 > $R/gpurun_out/stream_pmc0/pmc.log 2>&1
E2=$(date +%s.%N)
python3 -c "print('pmc pass over 216 step frames: %.1f s' % ($E2 - $S))"
grep -o '"front_fallbacks": [0-9]*' $R/gpurun_out/stream_pmc0/pmc.log | head -2
grep -o '"frames_per_s": [0-9.]*' $R/gpurun_out/stream_pmc0/pmc.log | head -2
cd $R
timeout 500 python -m pytest tests/test_gpu_stream.py -q 2>&1 | grep -E "passed|failed"
