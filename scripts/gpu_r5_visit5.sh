#!/bin/bash
# Round 5, visit 5: conv_in as one K = 32 MFMA step (A/B against the fp32 stencil), attention with LDS-DMA Q staging + deferred
# normalisation (phase stamps), op / UNet tests.   usage: scripts/gpu_r5_visit5.sh <tag>
set -u
TAG=${1:-r05e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_fast_mode.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15) > $OUT/pytest_ops_unet.log
tail -6 $OUT/pytest_ops_unet.log
(timeout 200 python scripts/attn_phases.py 2>&1 | grep -v amdgpu.ids) > $OUT/attn_phases.txt
head -14 $OUT/attn_phases.txt
for M in 1 0; do
  (ASYRP_CONV_IN_MFMA=$M timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check 2>$OUT/bench_cin$M.err | tail -1) > $OUT/bench_cin$M.json
  python - <<PY
import json
r = json.load(open("$OUT/bench_cin$M.json"))
print("ASYRP_CONV_IN_MFMA=$M images/s %.3f" % r["value"], [(x["kernel"][-30:], round(x["share_of_step"] * r["ms_per_step"] / max(x["launches_per_step"], 1) * 1e3, 1), "us", round(x.get("algorithmic_GBps", 0))) for x in r["kernel_families"] if "conv_in" in x["kernel"] or "attn" in x["kernel"]], r["roofline_attention"]["frac"])
PY
done
