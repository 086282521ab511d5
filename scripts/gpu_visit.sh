#!/bin/bash
# One GPU visit: the whole GPU suite, then a short bench of the line of record.  usage: scripts/gpu_visit.sh <tag> [bench args]
set -u
TAG=${1:-visit}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" | tail -60) > $OUT/pytest.log
tail -25 $OUT/pytest.log
(timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>> $OUT/bench.err | tail -1) > $OUT/bench.json
python - <<PY
import json
try:
    r = json.load(open("$OUT/bench.json"))
    print("images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:12]])
    print(json.dumps(r.get("parity_check"))[:400])
except Exception as e:
    print("bench FAILED", e)
PY
tail -3 $OUT/bench.err
