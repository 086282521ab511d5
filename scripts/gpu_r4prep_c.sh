#!/bin/bash
# prep branch, visit c: the first-convolution stencil kernel: tests + one whole-edit line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prep_c
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.wt/r4prep
(timeout 60 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "first_convolution" 2>&1 | tail -6) > $OUT/pytest.log
cat $OUT/pytest.log
(timeout 60 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check 2>> $OUT/ab.err | tail -1) > $OUT/conv_in_on.json
python - <<PY
import json
r = json.load(open("$OUT/conv_in_on.json"))
print("images/s %.3f" % r["value"], [(x["kernel"][-40:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4), round(x["algorithmic_GBps"])) for x in r["kernel_families"] if "conv_in" in x["kernel"] or "4, 1, 2, 4, 3, 1" in x["kernel"] or "conv_out" in x["kernel"]])
PY
