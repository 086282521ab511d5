#!/bin/bash
set -u
TAG=${1:-r03k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q -v 2>&1 | grep -E "FAILED|ERROR|passed|failed|Fatal|test_.*(PASSED|FAILED)" | tail -400) > $OUT/pytest.log
tail -5 $OUT/pytest.log
(ASYRP_QUAD8=0 timeout 200 python scripts/conv_bench.py 32 quad 2>&1 | grep -v amdgpu.ids | tail -10) > $OUT/ab_quad_off.txt
(timeout 200 python scripts/conv_bench.py 32 quad 2>&1 | grep -v amdgpu.ids | tail -10) > $OUT/ab_quad_on.txt
cat $OUT/ab_quad_off.txt $OUT/ab_quad_on.txt
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
for rnd in 1 2; do
  (timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_default_$rnd.json
  (ASYRP_QUAD8=0 timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_quad_off_$rnd.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r = json.load(open(f))
        small = [(x["kernel"][-34:], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r.get("kernel_families", []) if "8, 8, 8" in x["kernel"] or "true" in x["kernel"] or "2, 2, 1, 1, 3, 1" in x["kernel"]]
        print(f.split("/")[-1], "images/s %.3f" % r["value"], small)
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids $OUT/ab.err | tail -n 5
