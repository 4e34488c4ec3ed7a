#!/bin/bash
# Round-5 GPU call 2: hazard-safe forms of the eval_rhs diet + the lean kernel (rhs_lean.h): the whole GPU
# test tier and the default bench line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5b/bench_default.json'))
print('headline', r['value'], r['roofline']['frac'], 'secondary', (r.get('secondary') or {}).get('roofline', {}).get('frac'))
for k, v in r.get('configs', {}).items():
  if isinstance(v, dict):
    print(' ', k, v.get('kernel'), v.get('value'), (v.get('roofline') or {}).get('frac'))
PY
