#!/bin/bash
# kernel trace of the config-3 object update alone (per-kernel durations)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/objtrace -o t -- python3 $ROOT/scripts/gpu_object_stages.py > $ROOT/gpurun_out/objtrace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/objtrace/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    print(row['Name'][:60].replace('orcvio_amd::','').replace('void ',''), row['Calls'], round(float(row['AverageNs'])/1e3,1))
PY
