#!/bin/bash
# Round-4 GPU call 6: split integrators (two 32-row wavefronts per sample) -- full GPU tier + B=1024 A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4f
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 400 --steps 1000 --warmup 100"
for b in 256 512 1024; do
  python bench.py $common --batch $b > $O/split_b$b.json 2>>$O/err.log
  python bench.py $common --batch $b --kernel mfma64 > $O/onewave_b$b.json 2>>$O/err.log
done
python bench.py $common --batch 1024 --equation kdv > $O/split_kdv_b1024.json 2>>$O/err.log
python bench.py $common --batch 1024 --equation kdv --kernel mfma64 > $O/onewave_kdv_b1024.json 2>>$O/err.log
python bench.py $common --batch 2048 --num-points 32 > $O/split_n32_b2048.json 2>>$O/err.log
python bench.py $common --batch 2048 --num-points 32 --kernel mfma64 > $O/onewave_n32_b2048.json 2>>$O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4f/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:28s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
