#!/bin/bash
# Round 5, call 16: forcing batches (phases 1 + 2 of the harmonic sums once per three evaluations, on 60 / 36 lanes instead of 20 / 12).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5s; rm -rf $out; mkdir -p $out
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common > $out/headline.json 2>$out/err.log
python bench.py $common --batch 1024 > $out/b1024.json 2>>$out/err.log
python bench.py $common --scheme bs3 > $out/bs3.json 2>>$out/err.log
python bench.py $common --scheme rk4 > $out/rk4.json 2>>$out/err.log
python bench.py $common --non-conservative > $out/plain.json 2>>$out/err.log
python bench.py $common --state-dtype float64 > $out/f64.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5s/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
timeout 900 python -m pytest tests/test_gpu_integrate.py tests/test_gpu_reference_fixtures.py -q -m gpu -x > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -4 $out/tests.log
