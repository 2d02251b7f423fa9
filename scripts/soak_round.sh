#!/bin/bash
# Run on the GPU box (through gpurun): the randomised parity soaks on the current build, a couple of minutes each; one JSON per soak
# under gpurun_out/<tag>_soak_*.json and their summary <tag>_soaks.json.  usage: bash scripts/soak_round.sh r5 [seconds per soak]
TAG=${1:-rX}
SEC=${2:-120}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python3 scripts/gpu_soak_loop.py $SEC 170000 > $OUT/${TAG}_soak_loop.json 2>/dev/null
python3 scripts/gpu_soak.py $SEC 171000 > $OUT/${TAG}_soak_features.json 2>/dev/null
python3 scripts/gpu_soak_hybrid_full.py $SEC 172000 > $OUT/${TAG}_soak_hybrid_full.json 2>/dev/null
python3 scripts/gpu_soak_hybrid.py $SEC 174000 > $OUT/${TAG}_soak_hybrid.json 2>/dev/null
python3 scripts/gpu_soak_sharded.py $SEC 173000 > $OUT/${TAG}_soak_sharded.json 2>/dev/null
python3 scripts/gpu_soak_objects.py $SEC 175000 mix > $OUT/${TAG}_soak_objects.json 2>/dev/null
python3 scripts/gpu_soak_triangulate.py $SEC 176000 > $OUT/${TAG}_soak_triangulate.json 2>/dev/null
python3 scripts/gpu_soak_ipc.py $SEC 177000 2 > $OUT/${TAG}_soak_ipc.json 2>/dev/null
python3 - <<PY
import json, os
out = dict(what='randomised parity soaks on the round\'s final build (scripts/soak_round.sh ${TAG} ${SEC}): each compares the C-ABI results with the oracle on random inputs and lists every failing seed')
for name in ('loop', 'features', 'hybrid_full', 'hybrid', 'sharded', 'objects', 'triangulate', 'ipc'):
    p = os.path.join('$OUT', '${TAG}_soak_%s.json' % name)
    try:
        t = open(p).read()
        out[name] = json.loads(t[t.index('{'):])
    except Exception as e:
        out[name] = dict(error=repr(e))
json.dump(out, open(os.path.join('$OUT', '${TAG}_soaks.json'), 'w'), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if a not in ('failures', 'ranks')}, 'failures:', len(v.get('failures', [])) if isinstance(v.get('failures'), list) else v.get('failures'))
PY
