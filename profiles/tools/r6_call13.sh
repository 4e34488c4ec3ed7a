#!/bin/bash
# Round 6, call 13: two samples per lane in the lean kernels (rhs_lean2.h) -- tests, bench legs; then the whole GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lean.py -x -q > $O/lean_tests.log 2>&1; tail -12 $O/lean_tests.log
timeout 600 python bench.py --cpu-seconds 0 --secondary-batch 0 --configs one_layer_b4096,kdv_n64_b4096,tower_f16_b4096 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6m/bench.json'))
print('headline %.4f' % d['roofline']['frac'])
for k, v in d['configs'].items():
  print(k, '%.4e' % v['value'], '%.4f' % v['frac'], v.get('kernel'))
PY
( time timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1 ) 2> $O/tests.time; tail -6 $O/tests.log; cat $O/tests.time
