#!/bin/bash
# Round-2 bench matrix (one gpurun call).  JSON lines under gpurun_out/r2matrix/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2matrix
rm -rf $O; mkdir -p $O
b() { name=$1; shift; timeout 600 python bench.py "$@" > $O/bench_$name.json 2>/dev/null; }
b default
b default_s20 --steps 20 --warmup 5
b B2048 --batch 2048 --secondary-batch 0 --cpu-seconds 0
b B8192 --batch 8192 --secondary-batch 0 --cpu-seconds 0
b bs3_B4096 --scheme bs3 --secondary-batch 0 --cpu-seconds 0
b bs3_dt0.01_B1024 --batch 1024 --scheme bs3 --secondary-batch 0 --cpu-seconds 0
b persub_B4096 --launch-mode per_substep --secondary-batch 0 --cpu-seconds 0
b plain_B4096 --non-conservative --secondary-batch 0 --cpu-seconds 0
b kdv_B4096 --equation kdv --secondary-batch 0 --cpu-seconds 0
b ks256_B8192 --equation ks --num-points 256 --batch 8192 --steps 400 --secondary-batch 0 --cpu-seconds 0
b ks256_B8192_10k --equation ks --num-points 256 --batch 8192 --steps 10000 --warmup 100 --preheat-ms 0 --secondary-batch 0 --cpu-seconds 0
b ks64_B4096 --equation ks --secondary-batch 0 --steps 400 --cpu-seconds 0
b N32_B8192 --num-points 32 --batch 8192 --secondary-batch 0 --cpu-seconds 0
b N128_B2048 --num-points 128 --batch 2048 --secondary-batch 0 --cpu-seconds 0
b f64state_B4096 --state-dtype float64 --secondary-batch 0 --cpu-seconds 0
b nospec_B4096 --debug-option no_spec=1 --secondary-batch 0 --cpu-seconds 0
b generic_B1024 --batch 1024 --kernel generic --steps 100 --warmup 10 --preheat-ms 50 --secondary-batch 0 --cpu-seconds 0
b fixed_kdv_persub --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 100 --warmup 10 --secondary-batch 0 --cpu-seconds 0
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; s=d.get('secondary')
print('%-28s %.3e gps/s  %-4s %8.2f %-8s frac %.3f  %s reps %d' % ('$f'.split('/')[-1][6:-5], d['value'], r['bound'], r['achieved'], r['unit'], r['frac'], d['config']['kernel'], d['reps']), ('| secondary B%d %.2f TF %.3f' % (s['batch_per_gpu'], s['fp32_tflops'], s['frac'])) if s else '')"; done
