#!/bin/bash
# Round-4 final: the whole GPU test tier, then the measurement run (collect_r4.sh).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4final/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4final/pytest.log
tail -4 gpurun_out/r4final/pytest.log
bash profiles/tools/collect_r4.sh
