#!/bin/bash
# Round 6, call 4: adaptive-gap decomposition, WENO driver wall clock, remaining parity tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_weno.py tests/test_gpu_exact_solvers.py tests/test_gpu_full_size.py -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python profiles/tools/adaptive_gap.py > $O/adaptive_gap.txt 2> $O/gap.err; cat $O/adaptive_gap.txt; tail -3 $O/gap.err
timeout 600 python profiles/tools/weno_exact_bench.py > $O/weno_exact.txt 2> $O/weno.err; cat $O/weno_exact.txt; tail -3 $O/weno.err
