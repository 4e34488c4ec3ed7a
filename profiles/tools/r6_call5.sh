#!/bin/bash
# Round 6, call 5: the hoisted 256-row adaptive kernels (KdV / KS) -- parity, then the gap table again.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_evaluation.py tests/test_gpu_full_size.py -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python profiles/tools/adaptive_gap.py > $O/adaptive_gap.txt 2> $O/gap.err; cat $O/adaptive_gap.txt; tail -3 $O/gap.err
