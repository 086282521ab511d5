#!/bin/bash
# prep branch, visit d: conv_out with two chunks in flight: its tests + one whole-edit line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prep_d
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.wt/r4prep
(timeout 40 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "conv_out" 2>&1 | tail -4) > $OUT/pytest.log
cat $OUT/pytest.log
(timeout 50 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check 2>> $OUT/ab.err | tail -1) > $OUT/conv_out_prefetch2.json
python - <<PY
import json
r = json.load(open("$OUT/conv_out_prefetch2.json"))
print("images/s %.3f" % r["value"], [(x["kernel"][-40:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4), round(x["algorithmic_GBps"])) for x in r["kernel_families"] if "conv_in" in x["kernel"] or "conv_out" in x["kernel"]])
PY
