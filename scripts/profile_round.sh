#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + two PMC passes of bench.py; raw outputs under gpurun_out/<tag>_*.
# usage: bash scripts/profile_round.sh r1d
set -u
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ORCVIO_BENCH_DETAIL=${TAG}_bench_detail.json python3 $ROOT/bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
export ORCVIO_BENCH_DETAIL=${TAG}_scratch_detail.json   # (the profiled runs below must not overwrite the detail of the plain run)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -o p -- python3 $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1
done
# matrix-core activity: FP64 MFMA operations (x512 = flops), busy cycles (summed over the SIMDs) and the wall cycles of the dispatch
# (GRBM_GUI_ACTIVE: summed over the 8 XCDs, MI355X_MICROARCH.md)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_MFMA -o p -- python3 $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_MFMA.log 2>&1
find $OUT/${TAG}_trace $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_MFMA -name "*.csv" | head -20
# the summaries (what gets committed under profiles/) next to the raw outputs; the raw traces are too large to travel back
PROFILES_DST=$OUT/profiles_${TAG} python3 $ROOT/scripts/summarize_profiles.py ${TAG}
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_MFMA
