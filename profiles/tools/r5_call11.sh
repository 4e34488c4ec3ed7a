#!/bin/bash
# Round 5, call 11: 3-tap tower, three wavefronts per SIMD (single activation buffer, 168 VGPRs).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r5l; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_integrate.py -q -m gpu -x -k "kernel_size or tower or big or k3 or filter" > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -5 $out/tests.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"kernel_size": 3}' > $out/k3.json 2>$out/err.log
python bench.py $common --batch 6144 --hparams '{"kernel_size": 3}' > $out/k3_b6144.json 2>>$out/err.log
python bench.py $common --batch 8192 --hparams '{"kernel_size": 3}' > $out/k3_b8192.json 2>>$out/err.log
python bench.py $common --batch 3072 --hparams '{"kernel_size": 3}' > $out/k3_b3072.json 2>>$out/err.log
python bench.py $common --launch-mode per_substep --hparams '{"kernel_size": 3}' > $out/k3_persub.json 2>>$out/err.log
python bench.py $common --hparams '{"kernel_size": 3, "num_layers": 4}' > $out/k3_l4.json 2>>$out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5l/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $out/err.log
