#!/bin/bash
# Round-4 GPU call 4: the towers with streamed weights (7 taps, 64 filters, 3 taps).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_adaptive.py -m gpu -q -x -k "towers or generic_only or n128" -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 100 --warmup 20"
python bench.py $common --hparams '{"kernel_size": 7}' > $O/k7.json 2>>$O/err.log
python bench.py $common --hparams '{"filter_size": 64}' > $O/f64.json 2>>$O/err.log
python bench.py $common --hparams '{"filter_size": 64}' --batch 1024 > $O/f64_b1024.json 2>>$O/err.log
python bench.py $common --hparams '{"kernel_size": 3}' > $O/k3.json 2>>$O/err.log
python bench.py $common --hparams '{"filter_size": 16}' > $O/f16_embedded.json 2>>$O/err.log
python bench.py $common --hparams '{"kernel_size": 7}' --equation ks --num-points 256 --batch 2048 > $O/k7_ks256.json 2>>$O/err.log
python bench.py $common --hparams '{"filter_size": 64}' --equation ks --num-points 256 --batch 1024 > $O/f64_ks256.json 2>>$O/err.log
python bench.py $common --hparams '{"kernel_size": 7}' --kernel generic --batch 1024 --steps 20 > $O/k7_generic.json 2>>$O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4d/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:24s} {:14s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -5 $O/err.log
