#!/bin/bash
# Round-5 GPU call 5: the tests that failed in call 4 (+ the files they live in), the adaptive A/B
# (each of round 5's controller changes switched off in turn), the in-LDS FFT against rocFFT.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lean.py tests/test_gpu_exact_solvers.py tests/test_gpu_adaptive.py tests/test_gpu_rhs.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
A="--secondary-batch 0 --cpu-seconds 0 --steps 20 --warmup 5 --min-timed-ms 50 --preheat-ms 50 --configs adaptive_rk23,adaptive_kdv_n64_b4096,adaptive_ks_n256_b1024"
for lib in product adnoshort adnopref adnobfly adr4 product; do
  L=""; [ $lib != product ] && L="--library $lib"
  python bench.py $A $L > $O/adaptive_$lib.json 2>>$O/err.log
  python - $lib <<'PY'
import json, sys
tag = sys.argv[1]
r = json.load(open('gpurun_out/r5e/adaptive_%s.json' % tag))
print(tag, ' '.join('{}={:.3e}/{:.1f}%'.format(k.replace('adaptive_', ''), v['value'], 100 * v['frac']) for k, v in r['configs'].items() if isinstance(v, dict)))
PY
done
timeout 600 python profiles/tools/spectral_exact_bench.py > $O/spectral_exact.txt 2>>$O/err.log
cat $O/spectral_exact.txt
tail -3 $O/err.log
