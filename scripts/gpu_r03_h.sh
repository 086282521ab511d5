#!/bin/bash
set -u
TAG=${1:-r03h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fast_mode.py tests/test_gpu_unet.py tests/test_gpu_iddpm.py -m gpu -q 2>&1 | tail -8) > $OUT/pytest_subset.log
tail -4 $OUT/pytest_subset.log
python scripts/attn_phases.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_phases.txt
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
for rnd in 1 2; do
  (timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_default_$rnd.json
  (ASYRP_ATTN=old timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_attn_old_$rnd.json
done
(timeout 200 python bench.py --config imagenet $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_imagenet_default.json
(ASYRP_ATTN=old timeout 200 python bench.py --config imagenet $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_imagenet_attn_old.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r = json.load(open(f))
        att = r.get("roofline_attention", {})
        per = [(x["kernel"][-26:], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r.get("kernel_families", []) if "attn" in x["kernel"]]
        print(f.split("/")[-1], "images/s %.3f" % r["value"], "attention TF %.1f frac %.3f share %.4f" % (att.get("achieved", 0), att.get("frac", 0), att.get("share_of_step", 0)), per)
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids $OUT/ab.err | tail -n 5
