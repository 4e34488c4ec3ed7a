#!/bin/bash
# Round 6, call 6: phase trace of the adaptive kernels (probe library), rest of the GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f; rm -rf $O; mkdir -p $O
timeout 600 python profiles/tools/adaptive_phase_trace.py > $O/adaptive_phase_trace.txt 2> $O/trace.err; cat $O/adaptive_phase_trace.txt; tail -5 $O/trace.err
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_full_size.py > $O/tests.log 2>&1; tail -8 $O/tests.log
