#!/bin/bash
# Round 6, call 11: issue priority in the run-time-parameterised kernels (mode 0 = product, variants rt2 / rt3).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
L="--cpu-seconds 0 --secondary-batch 0 --configs tower_k7_b4096,tower_f64_b4096,tower_k3_b4096,wide_ks_g9_b4096,tower_k7f64_b4096"
timeout 600 python bench.py $L > $O/bench_rt0.json 2> $O/bench_rt0.err
timeout 600 python bench.py $L --library rt2 > $O/bench_rt2.json 2> $O/bench_rt2.err
timeout 600 python bench.py $L --library rt3 > $O/bench_rt3.json 2> $O/bench_rt3.err
python - <<'PY'
import json
for tag in ('rt0', 'rt2', 'rt3'):
  d = json.load(open('gpurun_out/r6k/bench_%s.json' % tag))
  row = [tag]
  for k, v in d['configs'].items():
    row.append('%s %.4f' % (k, v['frac']))
  print(' | '.join(row))
PY
