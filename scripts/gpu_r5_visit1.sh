#!/bin/bash
# Round 5, visit 1: (a) loop-shape calibration of the candidate tilings, (b) the 2.19 vs 2.48 us/step
# K-loop question: dispatch map, forced stagger of a CU's two workgroups, sclk / power per variant, product kernel under the stagger
# switches, (c) the round-5 parity tests.   usage: scripts/gpu_r5_visit1.sh <tag>
set -u
TAG=${1:-r05a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scripts/gpuclk.py > $OUT/gpuclk_probe.txt 2>&1
(timeout 200 scripts/calib/bin/loop_shapes_r5) > $OUT/calib_loop_shapes_r5.txt 2>&1
tail -12 $OUT/calib_loop_shapes_r5.txt
(timeout 200 python scripts/k32_phases.py 32 map) > $OUT/k32_dispatch_map.txt 2>&1
(timeout 300 python scripts/k32_phases.py 32 stagger) > $OUT/k32_stagger_stamps.txt 2>&1
(timeout 200 python scripts/k32_phases.py 32 clk) > $OUT/k32_clk_per_variant.txt 2>&1
for M in 0 1 2; do
  (ASYRP_STAGGER=$M timeout 200 python scripts/k32_phases.py 32 prod) > $OUT/k32_prod_stagger$M.txt 2>&1
done
(ASYRP_STAGGER=1 ASYRP_STAGGER_ROUNDS=2 timeout 200 python scripts/k32_phases.py 32 prod) > $OUT/k32_prod_stagger1_rounds2.txt 2>&1
(ASYRP_STAGGER=0 timeout 200 python scripts/k32_phases.py 32 prod) > $OUT/k32_prod_stagger0_again.txt 2>&1
grep -v amdgpu.ids $OUT/k32_dispatch_map.txt | tail -4
grep -E "^##|per launch|K loop per step|beside" $OUT/k32_stagger_stamps.txt | grep -v amdgpu.ids
grep -v amdgpu.ids $OUT/k32_clk_per_variant.txt
paste -d'|' <(grep -E "@" $OUT/k32_prod_stagger0.txt | cut -c1-60) <(grep -E "@" $OUT/k32_prod_stagger1.txt | cut -c32-60) <(grep -E "@" $OUT/k32_prod_stagger2.txt | cut -c32-60) <(grep -E "@" $OUT/k32_prod_stagger1_rounds2.txt | cut -c32-60) <(grep -E "@" $OUT/k32_prod_stagger0_again.txt | cut -c32-60)
(timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" | tail -40) > $OUT/pytest_round5.log
tail -30 $OUT/pytest_round5.log
