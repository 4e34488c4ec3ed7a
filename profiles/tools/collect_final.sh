#!/bin/bash
# Round-end measurement run (one gpurun call): parity suite, smoke, the bench
# matrix, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -2 > $O/pytest_gpu.txt
b() { name=$1; shift; timeout 600 python bench.py "$@" > $O/bench_$name.json 2>/dev/null; }
b default
b B2048 --batch 2048 --cpu-seconds 0
b B4096 --batch 4096 --cpu-seconds 0
b B8192 --batch 8192 --cpu-seconds 0
b bs3_B4096 --batch 4096 --scheme bs3 --cpu-seconds 0
b bs3_dt0.01_B1024 --scheme bs3 --cpu-seconds 0
b persub_B4096 --batch 4096 --launch-mode per_substep --cpu-seconds 0
b plain_B4096 --batch 4096 --non-conservative --cpu-seconds 0
b kdv_B4096 --equation kdv --batch 4096 --cpu-seconds 0
b ks256_B8192 --equation ks --num-points 256 --batch 8192 --steps 400 --cpu-seconds 0
b N32_B8192 --num-points 32 --batch 8192 --cpu-seconds 0
b N128_B2048 --num-points 128 --batch 2048 --cpu-seconds 0
b generic_B1024 --kernel generic --steps 100 --warmup 10 --cpu-seconds 0
b fixed_kdv_persub --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 100 --warmup 10 --cpu-seconds 0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o bench -- python bench.py --cpu-seconds 0 > $O/prof_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o bench -- python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 100 --warmup 10 --cpu-seconds 0 > $O/prof_stream.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_B4096 -o pmc -- python bench.py --batch 4096 --steps 200 --warmup 0 --cpu-seconds 0 > $O/pmc_sq_B4096.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_B1024 -o pmc -- python bench.py --batch 1024 --steps 200 --warmup 0 --cpu-seconds 0 > $O/pmc_sq_B1024.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_default -o pmc -- python bench.py --cpu-seconds 0 > $O/pmc_${c}_default.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_stream -o pmc -- python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 20 --warmup 2 --cpu-seconds 0 > $O/pmc_${c}_stream.log 2>&1
done
timeout 300 python profiles/tools/trace_phases.py 1024 2>&1 | tail -7 > $O/phases_B1024.txt
timeout 300 python profiles/tools/trace_phases.py 4096 2>&1 | tail -7 > $O/phases_B4096.txt
# keep the merge small: drop per-dispatch traces except the PMC collections
find $O -name "*_kernel_trace.csv" -size +4M -delete
du -sh $O | tail -1
cat $O/smoke.txt $O/pytest_gpu.txt
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], '%.3e' % d['value'], r['bound'], '%.2f %s' % (r['achieved'], r['unit']), '%.3f' % r['frac'], d['config']['kernel'])"; done
