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
