#!/bin/bash
# Round 5, call 13: streamed towers after the stream prologue was fenced off (weights now really arrive two groups ahead).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5o; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_adaptive.py -q -m gpu -x -k "towers" > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -4 $out/tests.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"kernel_size": 3}' > $out/k3.json 2>$out/err.log
python bench.py $common --hparams '{"kernel_size": 7}' > $out/k7.json 2>>$out/err.log
python bench.py $common --hparams '{"filter_size": 64}' > $out/c64.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 7, "filter_size": 64}' > $out/k7c64.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 3, "num_layers": 4}' > $out/k3_l4.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 7, "num_layers": 4}' > $out/k7_l4.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --hparams '{"kernel_size": 7}' > $out/k7_persub.json 2>>$out/err.log
python bench.py $common --equation ks --num-points 256 --batch 4096 --hparams '{"kernel_size": 7}' > $out/k7_ks256.json 2>>$out/err.log
python bench.py $common --equation ks --num-points 256 --batch 2048 --hparams '{"filter_size": 64}' > $out/c64_ks256.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5o/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
