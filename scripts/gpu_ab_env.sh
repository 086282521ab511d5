#!/bin/bash
# Same-box A/B of library switches on the line of record: scripts/gpu_ab_env.sh <tag> "<ENV=.. ...>" ["<ENV2=..>" ...]; baseline interleaved
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --no-cpu-baseline"
i=0
for rnd in 1 2; do
  for e in "X=0" "$@"; do
    i=$((i+1))
    (env $e timeout 200 python bench.py $B 2>> $OUT/err.txt | tail -1) > $OUT/run_$i.json
    python - <<PY
import json
try:
    r = json.load(open("$OUT/run_$i.json"))
    print("%-28s images/s %.3f  invariance %s " % ("$e", r["value"], r.get("parity_check", {}).get("batch_invariance_bitwise")), [(x["kernel"][-22:], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:7]])
except Exception as ex:
    print("$e", "FAILED", ex)
PY
  done
done
tail -3 $OUT/err.txt | grep -v amdgpu.ids
