#!/bin/bash
# Round 6, call 3: the four-wavefront (kQuad) integrators -- parity, then the small-ensemble A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_integrate.py tests/test_gpu_rhs.py tests/test_gpu_weno.py tests/test_gpu_lean.py -q -x > $O/tests.log 2>&1
tail -30 $O/tests.log
timeout 600 python profiles/tools/small_ensemble_ab.py > $O/small_ensembles.txt 2> $O/small.err
cat $O/small_ensembles.txt; tail -3 $O/small.err
