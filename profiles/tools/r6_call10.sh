#!/bin/bash
# Round 6, call 10: the block-diagonal tower of nets with <= 16 filters (tests, bench leg), priority by
# equation (product) against mode 4 (forcing phases lowered), the ring tests again.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_half_tower.py tests/test_gpu_ring.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_rhs.py -x -q -k "smaller_towers or nan" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
L="--cpu-seconds 0 --secondary-batch 1024 --configs adaptive_rk23,kdv_n64_b4096,tower_f16_b4096"
timeout 600 python bench.py $L > $O/bench_product.json 2> $O/bench_product.err
timeout 600 python bench.py $L --library prio4 > $O/bench_prio4.json 2> $O/bench_prio4.err
python - <<'PY'
import json
for tag in ('product', 'prio4'):
  d = json.load(open('gpurun_out/r6j/bench_%s.json' % tag))
  row = [tag, 'headline %.4f' % d['roofline']['frac'], 'b1024 %.4f' % d['secondary']['frac']]
  for k, v in d['configs'].items():
    row.append('%s %.4f (%s)' % (k, v['frac'], v.get('kernel')))
  print(' | '.join(row))
PY
