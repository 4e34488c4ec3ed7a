#!/bin/bash
# Round-4 GPU call 1: the GPU test tier, the full bench line, and the first A/B
# (SLP vectorizer off in the MFMA units: libddd1d_noslp.so).  -> gpurun_out/r4a/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
for lib in product noslp; do
  L=""; [ $lib = noslp ] && L="--library noslp"
  python bench.py $common $L > $O/ab_${lib}_headline.json 2>>$O/ab.err
  python bench.py $common $L --batch 1024 > $O/ab_${lib}_b1024.json 2>>$O/ab.err
  python bench.py $common $L --equation ks --num-points 256 --batch 8192 > $O/ab_${lib}_ks256.json 2>>$O/ab.err
  python bench.py $common $L --hparams '{"nonlinearity": "tanh", "num_layers": 4}' > $O/ab_${lib}_rt_tanh4.json 2>>$O/ab.err
  python bench.py $common $L --hparams '{"model_target": "time_derivative"}' > $O/ab_${lib}_rt_time_head.json 2>>$O/ab.err
  python bench.py $common $L --hparams '{"num_layers": 4}' > $O/ab_${lib}_rt_relu4.json 2>>$O/ab.err
  python bench.py $common $L --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $O/ab_${lib}_wide_ks_cgms9.json 2>>$O/ab.err
  python bench.py $common $L --equation ks --hparams '{"polynomial_accuracy_order": 0}' > $O/ab_${lib}_wide_ks_pao0.json 2>>$O/ab.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4a/ab_*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:36s} {:14s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % wall {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], 100 * r['roofline']['frac_wall'], r['config']['finite']))
PY
