#!/bin/bash
# Round 3, visit D: GPU tests; deep weight ring for the 16x16 / 8x8 layers and the re-pipelined split-plane attention: per-layer
# micro-benchmarks and same-box whole-edit A/Bs through the library's switches.  usage: scripts/gpu_r03_d.sh <tag>
set -u
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/pytest.log
tail -6 $OUT/pytest.log
(ASYRP_WEIGHT_REGS=0 timeout 200 python scripts/conv_bench.py 32 small 2>&1 | tail -20) > $OUT/ab_small_layers_ldsdma.txt
(timeout 200 python scripts/conv_bench.py 32 small 2>&1 | tail -20) > $OUT/ab_small_layers_regs.txt
cat $OUT/ab_small_layers_ldsdma.txt $OUT/ab_small_layers_regs.txt
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
for rnd in 1 2; do
  (timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_default_$rnd.json
  (ASYRP_ATTN=old timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_attn_old_$rnd.json
  (ASYRP_WEIGHT_REGS=0 timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_weight_regs_off_$rnd.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r = json.load(open(f))
        att = r.get("roofline_attention", {})
        fam = {x["kernel"]: x for x in r.get("kernel_families", [])}
        small = [(k[-22:], round(v["tflops"], 1), round(v["share_of_step"], 4)) for k, v in fam.items() if "8, 8, 8" in k or "<8, 4" in k]
        print(f.split("/")[-1], "images/s %.3f" % r["value"], "attention TF %.1f frac %.3f share %.4f" % (att.get("achieved", 0), att.get("frac", 0), att.get("share_of_step", 0)), small)
    except Exception as e:
        print(f, "ERR", e)
PY
tail -n 3 $OUT/*.err
