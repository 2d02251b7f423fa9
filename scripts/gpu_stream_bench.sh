#!/bin/bash
# The C++ stream harness on the GPU box: build, generate the two streams, run every mode, optionally under a kernel trace.
# usage: bash scripts/gpu_stream_bench.sh [tag] [trace]
set -u
TAG=${1:-s}
OUT=gpurun_out/stream_$TAG
mkdir -p $OUT
DEFS=""
grep -q orcvio_msckf_io_step_frame include/orcvio_msckf.h && DEFS="-DORCVIO_HAVE_STEP_FRAME"
g++ -O2 -std=c++17 $DEFS -o $OUT/stream_bench tests/cpp/stream_bench.cpp -L orcvio_amd/lib -lorcvio_msckf -Wl,-rpath,$PWD/orcvio_amd/lib || exit 1
python - <<PY
from orcvio_amd import synth
fl = synth.Flags(use_larvio=1)
fr, P0 = synth.make_stream(fl)
synth.write_stream('$OUT/config1.bin', fr, P0, fl)
fl5 = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1)
fr, P0 = synth.make_stream(fl5, sigma_px=0.008)
synth.write_stream('$OUT/config5.bin', fr, P0, fl5)
PY
for cfg in config1 config5; do
  for rep in 1 2; do
    $OUT/stream_bench --stream $OUT/$cfg.bin --mode calls --frames 480 | tee -a $OUT/$cfg.jsonl
    $OUT/stream_bench --stream $OUT/$cfg.bin --mode calls --no-prefactor --frames 480 | tee -a $OUT/$cfg.jsonl
    if [ -n "$DEFS" ]; then
      $OUT/stream_bench --stream $OUT/$cfg.bin --mode step --frames 480 | tee -a $OUT/$cfg.jsonl
    fi
  done
done
if [ "${2:-}" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  M=calls; [ -n "$DEFS" ] && M=step
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_$M -o t -- $GRAFT_REPO_ROOT/$OUT/stream_bench --stream $GRAFT_REPO_ROOT/$OUT/config1.bin --mode $M --frames 64 --warmup 16 > $GRAFT_REPO_ROOT/$OUT/trace_$M.log 2>&1
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_calls_nopf -o t -- $GRAFT_REPO_ROOT/$OUT/stream_bench --stream $GRAFT_REPO_ROOT/$OUT/config1.bin --mode calls --no-prefactor --frames 64 --warmup 16 > $GRAFT_REPO_ROOT/$OUT/trace_calls_nopf.log 2>&1
  cd $GRAFT_REPO_ROOT
  find $OUT -name "*.csv" | head; du -sh $OUT
fi
