#!/bin/bash
# Round-5 GPU call 6: the whole GPU test tier on the current tree, the default bench line, the matrix of
# kernels outside it (profiles/tools/bench_matrix_r5.sh).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5f
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5f/bench_default.json'))
s = r.get('secondary') or {}
print('headline %.4e %.4f | secondary %.4e %s' % (r['value'], r['roofline']['frac'], s.get('value', 0), s.get('frac')))
for k, v in r.get('configs', {}).items():
  if isinstance(v, dict):
    print('  %-26s %-16s %.4e  frac %s' % (k, v.get('kernel'), v.get('value', 0), v.get('frac')))
PY
bash profiles/tools/bench_matrix_r5.sh
