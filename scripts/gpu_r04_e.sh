#!/bin/bash
# round 4, visit e: conv_out barrier fix: determinism probe, new test, bench with the parity check
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
(timeout 120 python scripts/batch_invariance_probe.py 32 2>&1 | grep -v amdgpu.ids) > $OUT/probe.txt
cat $OUT/probe.txt
(timeout 300 python -m pytest tests/test_gpu_unet.py -q -x 2>&1 | tail -5) > $OUT/pytest.log
cat $OUT/pytest.log
(timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>> $OUT/ab.err | tail -1) > $OUT/base.json
python - <<PY
import json
r = json.load(open("$OUT/base.json"))
print("images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:12]])
print(json.dumps(r.get("parity_check"))[:600])
PY
tail -3 $OUT/ab.err
