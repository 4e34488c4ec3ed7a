out=gpurun_out/${1:-r2d}
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_integrate.py -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -5 $out/pytest_gpu.log
for opt in "" "--debug-option no_pair=1"; do
  for b in 2048 4096 8192; do
    timeout 300 python bench.py --batch $b --secondary-batch 0 --cpu-seconds 0 $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(c['batch_per_gpu'], c['kernel'], c['debug_options'], '%.3e'%d['value'], '%.2f TF'%r['achieved'], '%.3f'%r['frac'], 'finite', c['finite'])"
  done
done
