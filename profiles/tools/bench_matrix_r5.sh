#!/bin/bash
# Round-5 matrix of the kernels that are NOT in the default bench line: the run-time-parameterised
# MFMA kernels on the default tower (4 relu layers, 4 tanh layers, time-derivative head), the wide
# flavour, ensemble sizes around the headline, the fused streaming step after its scratch frame went.
#   bash profiles/tools/bench_matrix_r5.sh  ->  gpurun_out/r5m/*.json
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5m; rm -rf $out; mkdir -p $out
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"num_layers": 4}' > $out/rt_relu4.json 2>$out/err.log
python bench.py $common --hparams '{"nonlinearity": "tanh", "num_layers": 4}' > $out/rt_tanh4.json 2>>$out/err.log
python bench.py $common --hparams '{"model_target": "time_derivative"}' > $out/rt_time_head.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $out/wide_ks_cgms9.json 2>>$out/err.log
python bench.py $common --equation ks --hparams '{"polynomial_accuracy_order": 0}' > $out/wide_ks_pao0.json 2>>$out/err.log
python bench.py $common --non-conservative > $out/burgers_plain.json 2>>$out/err.log
python bench.py $common --batch 2048 > $out/b2048.json 2>>$out/err.log
python bench.py $common --batch 8192 > $out/b8192.json 2>>$out/err.log
python bench.py $common --scheme bs3 > $out/bs3.json 2>>$out/err.log
python bench.py $common --state-dtype float64 > $out/f64state.json 2>>$out/err.log
python bench.py $common --equation ks --num-points 64 > $out/ks64.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --batch 8192 > $out/persub_b8192.json 2>>$out/err.log
python bench.py $common --launch-mode per_step --batch 8192 > $out/perstep_b8192.json 2>>$out/err.log
python bench.py $common --equation kdv --baseline-stencils --batch 4096 --steps 1000 > $out/fixed_persistent_kdv.json 2>>$out/err.log
python bench.py $common --equation burgers --baseline-stencils --batch 4096 --steps 1000 > $out/fixed_persistent_burgers.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5m/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % (hbm {:6.1f} GB/s) finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['roofline']['hbm_gbps'], r['config']['finite']))
PY
tail -3 $out/err.log
