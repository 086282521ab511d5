#!/bin/bash
# round 3, visit q: gemm1x1.hip with the XCD-aware map: op tests, tile sweep with the map on / off, whole-edit A/B
set -u
TAG=${1:-r03q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "barrier_free" 2>&1 | tail -5) > $OUT/pytest_ops.log
cat $OUT/pytest_ops.log
(timeout 300 python scripts/conv_bench.py 32 onebyone 2>&1 | grep -v amdgpu.ids | tail -14) > $OUT/onebyone.txt
(ASYRP_XCD_MAP=0 timeout 300 python scripts/conv_bench.py 32 onebyone 2>&1 | grep -v amdgpu.ids | tail -14) > $OUT/onebyone_nomap.txt
cat $OUT/onebyone.txt $OUT/onebyone_nomap.txt
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
for rnd in 1 2; do
  (ASYRP_GEMM1X1=0 timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_g1_off_$rnd.json
  (timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_g1_128_$rnd.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r = json.load(open(f))
        small = [(x["kernel"][-40:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r.get("kernel_families", [])
                 if "gemm1x1" in x["kernel"] or "2, 2, 2, 2, 1, 1>" in x["kernel"]]
        print(f.split("/")[-1], "images/s %.3f" % r["value"], small)
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids $OUT/ab.err | tail -n 5
