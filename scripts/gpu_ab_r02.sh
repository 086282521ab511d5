#!/bin/bash
# interleaved same-box A/B: XCD-aware tile map on/off x {8-wave, 16-wave} main tile; then whole-edit bench lines
OUT=gpurun_out/${1:-r02c}
mkdir -p $OUT
for r in 1 2 3; do
  ASYRP_XCD_MAP=0 python scripts/ab_r02.py >> $OUT/ab_layers.txt 2>/dev/null
  ASYRP_XCD_MAP=1 python scripts/ab_r02.py >> $OUT/ab_layers.txt 2>/dev/null
done
cat $OUT/ab_layers.txt
for r in 1 2; do
  for cfg in "0 6" "1 6" "1 7"; do
    set -- $cfg
    echo "xmap=$1 main_tile=$2" >> $OUT/ab_bench.txt
    ASYRP_XCD_MAP=$1 ASYRP_MAIN_TILE=$2 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['value'], r['roofline']['achieved'], r['roofline']['kernel'], r.get('roofline_attention',{}).get('achieved'), r.get('roofline_attention',{}).get('share_of_step'))" >> $OUT/ab_bench.txt
  done
done
cat $OUT/ab_bench.txt
