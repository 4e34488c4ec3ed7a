out=gpurun_out/${1:-r2q}
mkdir -p $out
python -m pytest tests/test_gpu_integrate.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -4
run() { name=$1; shift; timeout 300 python bench.py "$@" --secondary-batch 0 --cpu-seconds 0 > $out/bench_$name.json 2>/dev/null; python -c "
import json,sys
d=json.load(open('$out/bench_$name.json')); r=d['roofline']; c=d['config']; print('%-22s'%'$name', c['equation'], c['num_points'], c['batch_per_gpu'], c['launch_mode'], '%.3e'%d['value'], '%.2f %s'%(r['achieved'], r['unit']), '%.3f'%r['frac'], c['kernel'])"; }
run persub_B4096 --launch-mode per_substep
run persub_B8192 --launch-mode per_substep --batch 8192
run persub_kdv --launch-mode per_substep --equation kdv
run stream --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 100 --warmup 10
