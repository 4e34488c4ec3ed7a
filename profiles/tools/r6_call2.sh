#!/bin/bash
# Round 6, call 2: whole GPU suite (NaN-through-relu, truth-ratio bounds, WENO kernels) + the default bench line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6b; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_full_size.py > $O/tests.log 2>&1 ) 2> $O/tests.time
tail -40 $O/tests.log; cat $O/tests.time
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6b/bench_default.json'))
print('headline', d['value'], d['roofline']['frac'], 'secondary', d.get('secondary',{}).get('roofline',{}).get('frac'))
for k,v in d.get('configs',{}).items():
    print('%-28s %10.3e %s frac %s issued %s'%(k, v.get('value',0), v.get('unit',''), v.get('frac'), v.get('frac_issued')))
PY
