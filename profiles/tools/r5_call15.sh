#!/bin/bash
# Round 5, call 15: issue priority by phase (s_setprio 3 through the matrix layers, 0 through the VALU phases).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5q; rm -rf $out; mkdir -p $out
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common > $out/headline.json 2>$out/err.log
python bench.py $common --batch 1024 > $out/b1024.json 2>>$out/err.log
python bench.py $common --equation kdv > $out/kdv.json 2>>$out/err.log
python bench.py $common --equation ks --num-points 256 --batch 8192 > $out/ks256.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep > $out/persub.json 2>>$out/err.log
python bench.py $common --launch-mode per_step > $out/perstep.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 7}' > $out/k7.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 3}' > $out/k3.json 2>>$out/err.log
python bench.py $common --hparams '{"model_target": "time_derivative"}' > $out/time_head.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $out/wide.json 2>>$out/err.log
python bench.py --cpu-seconds 0 --secondary-batch 0 --steps 20 --warmup 5 --min-timed-ms 50 --configs adaptive_rk23,adaptive_kdv_n64_b4096,adaptive_ks_n256_b1024 > $out/adaptive.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5q/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  if f.endswith('adaptive.json'):
    for k, v in r['configs'].items():
      print('{:30s} {:9.3e} {:5.1f} %'.format(k, v['value'], 100 * v['roofline']['frac']))
    continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
