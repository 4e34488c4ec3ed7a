#!/bin/bash
# Round 6, call 14: priority hooks in the four-wavefront groups' hidden layer (DDD_QUAD_PRIO=3): q2 = on top of the
# product's modes, q3 = with every per-equation kernel at mode 3.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
L="--cpu-seconds 0 --secondary-batch 0 --configs burgers_b256,burgers_b512,adaptive_rk23_b256"
for v in product q2 q3 product; do
  if [ $v = product ]; then lib=""; else lib="--library $v"; fi
  timeout 600 python bench.py $L $lib > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open('gpurun_out/r6n/bench_%s.json' % tag))
row = [tag]
for k, v in d['configs'].items():
  row.append('%s %.4f' % (k, v['frac']))
print(' | '.join(row))
PY
done
