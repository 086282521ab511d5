#!/bin/bash
# prep branch, visit b: the whole GPU suite on the prep build (defaults changed: 16x16 split-K on) + the small class on AFHQ
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prep_b
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.wt/r4prep
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest.log
cat $OUT/pytest.log
