#!/bin/bash
# Round-5 GPU call 8: the run-time-parameterised kernels after the derivative-at-a-time epilogue:
# whole GPU test tier + the matrix of kernels outside the default line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5h
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
bash profiles/tools/bench_matrix_r5.sh
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"model_target": "space_derivatives"}' > gpurun_out/r5m/rt_space_head.json 2>>gpurun_out/r5m/err.log
python bench.py $common --hparams '{"polynomial_accuracy_order": 0}' > gpurun_out/r5m/rt_pao0.json 2>>gpurun_out/r5m/err.log
python bench.py $common --hparams '{"polynomial_accuracy_order": 3}' > gpurun_out/r5m/rt_pao3.json 2>>gpurun_out/r5m/err.log
python - <<'PY'
import json
for f in ('rt_space_head', 'rt_pao0', 'rt_pao3'):
  r = json.load(open('gpurun_out/r5m/%s.json' % f))
  print('{:30s} {:16s} {:9.3e} pts/s {:5.1f} % finite={}'.format(f, r['config']['kernel'], r['value'], 100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
