#!/bin/bash
# Round 6, call 1: the new WENO kernels -- parity tests, then throughput against the generic kernel.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_weno.py tests/test_gpu_exact_solvers.py -x -q > $O/tests.log 2>&1
tail -25 $O/tests.log
timeout 600 python profiles/tools/weno_exact_bench.py > $O/weno_exact.txt 2> $O/weno_exact.err
cat $O/weno_exact.txt; tail -5 $O/weno_exact.err
