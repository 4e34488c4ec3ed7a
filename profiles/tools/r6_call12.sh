#!/bin/bash
# Round 6, call 12: output-layer operand groups in flight (DDD_FIN4_AHEAD 2 = product, 3, 4), priority mode 5
# (output layer's middle one level up) against the product, Burgers at mode 3 (b3).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
L="--cpu-seconds 0 --secondary-batch 1024 --configs adaptive_rk23,kdv_n64_b4096,ks_n256_b8192,adaptive_kdv_n64_b4096"
for v in product ahead3 ahead4 prio5 b3 product; do
  if [ $v = product ]; then lib=""; else lib="--library $v"; fi
  timeout 600 python bench.py $L $lib > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open('gpurun_out/r6l/bench_%s.json' % tag))
row = [tag, 'headline %.4f' % d['roofline']['frac'], 'b1024 %.4f' % d['secondary']['frac']]
for k, v in d['configs'].items():
  row.append('%s %.4f' % (k, v['frac']))
print(' | '.join(row))
PY
done
