#!/bin/bash
# Round-4 GPU call 2: the adaptive kernels after the controller moved to LDS
# (tests first), then A/B of the lean 64-row variant.  -> gpurun_out/r4b/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_evaluation.py tests/test_gpu_exact_solvers.py tests/test_gpu_checkpoint.py tests/test_gpu_reference_integrate_suite.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
common="--secondary-batch 0 --cpu-seconds 0 --min-timed-ms 200 --steps 200 --warmup 20"
for lib in product adnolean; do
  L=""; [ $lib != product ] && L="--library $lib"
  python bench.py $common $L --configs adaptive_rk23,adaptive_kdv_n64_b4096,adaptive_ks_n256_b1024 > $O/ab_${lib}.json 2>>$O/ab.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4b/ab_*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  for k, v in r['configs'].items():
    print('{:22s} {:26s} {:9.3e} evals/s {:6.1f} % nfev {}..{} ms/launch {:.2f} finished {}'.format(
        f.split('/')[-1], k, v['value'], 100 * v['frac'], v['nfev_min'], v['nfev_max'],
        v['kernel_ms_per_launch'], v['samples_finished']))
PY
