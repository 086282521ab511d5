#!/bin/bash
# new tests + full GPU suite + tile 1 vs tile 6 (8-wave) A/B, micro and whole-bench.  usage: scripts/gpu_w8.sh <tag>
set -u
TAG=${1:-w8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 400 python scripts/conv_bench.py 32 > $OUT/conv_bench.txt 2>&1
grep -A 12 "8-wave plain-loop" $OUT/conv_bench.txt
for r in 1 2; do
  timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_main_$r.json 2>$OUT/bench_main_$r.err
  ASYRP_MAIN_TILE=6 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_w8_$r.json 2>$OUT/bench_w8_$r.err
done
python - <<PY
import json
for n in ("main_1","w8_1","main_2","w8_2"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],3), "img/s", round(d["roofline"]["achieved"],1), "TF", d["roofline"]["kernel"])
    except Exception as e:
        print(n, "failed", e)
PY
