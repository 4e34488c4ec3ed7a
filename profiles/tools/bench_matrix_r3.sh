#!/bin/bash
# Round-3 matrix of the kernels that are NOT the headline: run-time-parameterised
# MFMA kernels (default models through the probe library's no_spec switch, and
# non-default nets), the wide flavour, the generic kernel next to them.
#   bash profiles/tools/bench_matrix_r3.sh  ->  gpurun_out/r3m/*.json
set -u
out=gpurun_out/r3m; mkdir -p $out
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --debug-option no_spec=1 > $out/rt_burgers.json 2>$out/err.log
python bench.py $common --debug-option no_spec=1 --equation kdv > $out/rt_kdv.json 2>>$out/err.log
python bench.py $common --debug-option no_spec=1 --equation ks > $out/rt_ks.json 2>>$out/err.log
python bench.py $common --hparams '{"nonlinearity": "tanh", "num_layers": 4}' > $out/rt_tanh4.json 2>>$out/err.log
python bench.py $common --hparams '{"model_target": "time_derivative"}' > $out/rt_time_head.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $out/wide_ks_cgms9.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"polynomial_accuracy_order": 0}' > $out/wide_ks_pao0.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"coefficient_grid_min_size": 9}' --kernel generic --batch 1024 > $out/generic_ks_cgms9.json 2>>$out/err.log
# smaller towers embedded with zero weights in the 5-tap x 32-channel MFMA layers; larger ones on the generic kernel
python bench.py $common --hparams '{"kernel_size": 3}' > $out/embedded_kernel3.json 2>>$out/err.log
python bench.py $common --hparams '{"filter_size": 16}' > $out/embedded_filter16.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 7}' --batch 1024 --steps 50 > $out/generic_kernel7.json 2>>$out/err.log
python bench.py $common --hparams '{"filter_size": 64}' --batch 1024 --steps 50 > $out/generic_filter64.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3m/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:28s} {:14s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
