#!/bin/bash
# Round-5 GPU call 7: direct heads after the two-chain output layer; one launch per substep with 2 / 3 / 4
# sample slabs side by side (probe library, substep_parts).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rhs.py tests/test_gpu_integrate.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 300 --steps 200 --warmup 20"
python bench.py $common --hparams '{"model_target": "time_derivative"}' > $O/rt_time_head.json 2>$O/err.log
python bench.py $common --hparams '{"model_target": "space_derivatives"}' > $O/rt_space_head.json 2>>$O/err.log
python bench.py $common --hparams '{"model_target": "flux"}' > $O/rt_flux_head.json 2>>$O/err.log
for b in 4096 8192; do
  python bench.py $common --launch-mode per_substep --batch $b > $O/persub_product_b$b.json 2>>$O/err.log
  for parts in 1 2 3 4; do
    python bench.py $common --launch-mode per_substep --batch $b --debug-option substep_parts=$parts > $O/persub_parts${parts}_b$b.json 2>>$O/err.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5g/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:30s} {:16s} {:9.3e} pts/s {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], 100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -3 $O/err.log
