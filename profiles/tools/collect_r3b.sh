#!/bin/bash
# Round-3 follow-up measurements (one gpurun call) after the walk kernels' setup
# work: per-wavefront lifetimes of the per-substep mode, the rocprofv3 timeline
# of its two chains, the per-substep / per-step / persistent lines at 8 192
# samples, and the matrix of the non-headline kernels.  Outputs: gpurun_out/r3b/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b
rm -rf $O; mkdir -p $O
timeout 200 python profiles/tools/substep_wave_trace.py 4096 12 2>/dev/null | grep -v "amdgpu.ids" > $O/wave_trace.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/overlap -o run -- python bench.py --configs none \
  --secondary-batch 0 --cpu-seconds 0 --launch-mode per_substep --steps 200 --warmup 200 --preheat-ms 50 \
  --min-timed-ms 0 > $O/overlap.log 2>&1
python profiles/tools/substep_overlap_trace.py $O/overlap > $O/substep_overlap.txt 2>&1
for mode in per_substep per_step persistent; do
  timeout 200 python bench.py --configs none --secondary-batch 0 --cpu-seconds 0 --batch 8192 --launch-mode $mode \
    > $O/bench_${mode}_B8192.json 2> $O/bench_${mode}_B8192.err
done
bash profiles/tools/bench_matrix_r3.sh > $O/matrix.txt 2>&1
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*.db" -delete
tail -15 $O/matrix.txt; cat $O/substep_overlap.txt | tail -3
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3b/bench_*.json')):
  r = json.load(open(f)); print(f.split('/')[-1], '%.3e' % r['value'], '%.1f %%' % (100 * r['roofline']['frac']))
PY
