#!/bin/bash
# Round-5 GPU call 4: adaptive kernels after the controller slimming (stages 2 / 3 without controller
# load / store / vote, cheap butterfly, next output time requested one evaluation ahead) -- cos / sin table
# in LDS / L2 (product) against resident (variant adtrig0); the whole GPU test tier incl. the new lean tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
A="--secondary-batch 0 --cpu-seconds 0 --steps 20 --warmup 5 --min-timed-ms 50 --preheat-ms 50 --configs adaptive_rk23,adaptive_kdv_n64_b4096,adaptive_ks_n256_b1024"
python bench.py $A > $O/adaptive_product.json 2>>$O/err.log
python bench.py $A --library adtrig0 > $O/adaptive_adtrig0.json 2>>$O/err.log
python - <<'PY'
import json
for tag in ('product', 'adtrig0'):
  try:
    r = json.load(open('gpurun_out/r5d/adaptive_%s.json' % tag))
  except Exception as e:
    print(tag, 'FAILED', e); continue
  for k, v in r['configs'].items():
    if isinstance(v, dict):
      print('{:10s} {:26s} {:.4e} {}  frac {}'.format(tag, k, v['value'], v.get('unit'), (v.get('roofline') or {}).get('frac')))
PY
tail -3 $O/err.log
