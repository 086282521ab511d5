#!/bin/bash
# prep branch, visit a (run from the worktree copy on the GPU box): split-K 16x16 + small batch class: tests, A/B
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prep_a
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.wt/r4prep
(timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch_class.py -m gpu -q -k "splitk_16 or quad_form or batch_class or small_class" 2>&1 | tail -15) > $OUT/pytest.log
cat $OUT/pytest.log
B="--steps 1 --warmup 0 --no-cpu-baseline --no-parity-check"
(ASYRP_SPLITK16=0 timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_sk16_off.json
(timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_sk16_on.json
(timeout 100 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check 2>> $OUT/ab.err | tail -1) > $OUT/b1_class32.json
(timeout 100 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --nominal-batch 1 2>> $OUT/ab.err | tail -1) > $OUT/b1_class1.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        r = json.load(open(f))
        fam = [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r.get("kernel_families", [])[:7]]
        print(f.split("/")[-1], "images/s %.3f  ms/step %.1f" % (r["value"], r["ms_per_step"]), fam)
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids $OUT/ab.err | tail -n 8
