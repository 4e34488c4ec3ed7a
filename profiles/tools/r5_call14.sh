#!/bin/bash
# Round 5, call 14: per-launch kernels with their kernel-argument lines requested together at the top.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5p; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_integrate.py tests/test_gpu_reference_integrate_suite.py -q -m gpu -x > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -4 $out/tests.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common > $out/persistent.json 2>$out/err.log
python bench.py $common --launch-mode per_substep > $out/persub_b4096.json 2>>$out/err.log
python bench.py $common --launch-mode per_step > $out/perstep_b4096.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --batch 8192 > $out/persub_b8192.json 2>>$out/err.log
python bench.py $common --launch-mode per_step --batch 8192 > $out/perstep_b8192.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --batch 2048 > $out/persub_b2048.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --equation kdv > $out/persub_kdv.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --equation ks --num-points 256 --batch 4096 > $out/persub_ks256.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5p/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
