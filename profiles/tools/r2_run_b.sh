# GPU parity suite + bench (headline + secondary) for a kernel change
out=gpurun_out/${1:-r2b}
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -5 $out/pytest_gpu.log
python bench.py --cpu-seconds 0 > $out/bench_s1000.json 2> $out/bench_s1000.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $out/bench_s20.json 2> $out/bench_s20.err
python - <<PY
import json
for f in ('bench_s1000','bench_s20'):
    try:
        d=json.load(open('$out/%s.json'%f))
        print(f, 'B4096 %.1f TF (%.3f)'%(d['roofline']['fp32_tflops'], d['roofline']['frac']), 'B1024 %.1f TF (%.3f)'%(d['secondary']['fp32_tflops'], d['secondary']['frac']), d['clocks'])
    except Exception as e:
        print(f, 'failed', e); print(open('$out/%s.err'%f).read()[-2000:])
PY
run() { name=$1; shift; timeout 300 python bench.py "$@" --secondary-batch 0 --cpu-seconds 0 > $out/bench_$name.json 2>/dev/null; python -c "
import json,sys
d=json.load(open('$out/bench_$name.json')); r=d['roofline']; c=d['config']; print('%-22s'%'$name', c['equation'], c['num_points'], c['batch_per_gpu'], c['scheme'], c['launch_mode'], c['state_dtype'], '%.3e'%d['value'], '%.2f %s'%(r['achieved'], r['unit']), '%.3f'%r['frac'], c['kernel'])"; }
if [ "$2" = "matrix" ]; then
run persub_B4096 --launch-mode per_substep
run f64state_B4096 --state-dtype float64
run kdv_B4096 --equation kdv
run ks256_B8192 --equation ks --num-points 256 --batch 8192 --steps 400
run N128_B2048 --num-points 128 --batch 2048
run plain_B4096 --non-conservative
run bs3_B4096 --scheme bs3
fi
