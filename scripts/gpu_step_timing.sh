#!/bin/bash
# On the GPU box: the C++ stream harness linked against the DIAGNOSTICS build with ORCVIO_TIMING=1 -- orcvio_msckf_io_step_frame prints
# the host's wall time at its stations (mean of 128 calls, microseconds after entry).
set -u
OUT=gpurun_out/stream_timing
mkdir -p $OUT
g++ -O2 -std=c++17 -DORCVIO_HAVE_STEP_FRAME -o $OUT/stream_bench_dbg tests/cpp/stream_bench.cpp -L orcvio_amd/lib -lorcvio_msckf_dbg -Wl,-rpath,$PWD/orcvio_amd/lib || exit 1
python - <<PY
from orcvio_amd import synth
fl = synth.Flags(use_larvio=1)
fr, P0 = synth.make_stream(fl)
synth.write_stream('$OUT/config1.bin', fr, P0, fl)
PY
ORCVIO_TIMING=1 $OUT/stream_bench_dbg --stream $OUT/config1.bin --mode step --frames 480 2>&1 | tail -5 | cut -c1-420
