#!/bin/bash
# Round 5, call 12: the 7 taps x 64 filters tower (hidden layers rolled over the taps).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5n; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_adaptive.py -q -m gpu -x -k "towers or generic_only" > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -5 $out/tests.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"kernel_size": 7, "filter_size": 64}' > $out/k7c64.json 2>$out/err.log
python bench.py $common --equation ks --num-points 256 --batch 2048 --hparams '{"kernel_size": 7, "filter_size": 64}' > $out/k7c64_ks256.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 3, "filter_size": 64}' > $out/k3c64_in_5x64.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5n/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
