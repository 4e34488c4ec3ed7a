#!/bin/bash
# Round 5, call 9: the wide flavour with its output layer always folded (no projection in the
# epilogue), strides / dilation in the layer API.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5i; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_ops.py tests/test_gpu_adaptive.py -q -m gpu -x > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -5 $out/tests.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $out/wide_ks_cgms9.json 2>$out/err.log
python bench.py $common --equation ks --hparams '{"polynomial_accuracy_order": 0}' > $out/wide_ks_pao0.json 2>>$out/err.log
python bench.py $common --equation kdv --hparams '{"coefficient_grid_min_size": 10}' > $out/wide_kdv_cgms10.json 2>>$out/err.log
python bench.py $common --equation burgers --hparams '{"coefficient_grid_min_size": 12}' > $out/wide_burgers_cgms12.json 2>>$out/err.log
python bench.py $common --equation ks --num-points 256 --batch 4096 --hparams '{"coefficient_grid_min_size": 9}' > $out/wide_ks256_cgms9.json 2>>$out/err.log
python bench.py $common --hparams '{"model_target": "time_derivative"}' > $out/rt_time_head.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5i/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
