#!/bin/bash
# Round-4 GPU call 7: per-equation integrators compiled for ONE wavefront per SIMD (512 registers), at B = 1024.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h
rm -rf $O; mkdir -p $O
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 400 --steps 1000 --warmup 100"
for lib in product onewave; do
  L=""; [ $lib != product ] && L="--library $lib"
  python bench.py $common $L --batch 1024 > $O/${lib}_b1024.json 2>>$O/err.log
  python bench.py $common $L --batch 1024 --equation kdv > $O/${lib}_kdv_b1024.json 2>>$O/err.log
  python bench.py $common $L --batch 4096 > $O/${lib}_b4096.json 2>>$O/err.log
  python bench.py $common $L --batch 2048 --equation ks --num-points 256 --steps 400 > $O/${lib}_ks256_b2048.json 2>>$O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4h/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:28s} {:16s} {:9.3e} pts/s {:7.1f} TF {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], r['roofline']['fp32_tflops'],
      100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
