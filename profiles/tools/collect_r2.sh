#!/bin/bash
# Round-2 measurement run (one gpurun call): rocprofv3 kernel trace of the
# bench command, PMC passes (SQ counters; FETCH_SIZE and WRITE_SIZE in separate
# passes), phase trace.  Outputs under gpurun_out/r2prof/ (scratch);
# profiles/tools/summarize_r2.py turns them into the committed profiles/r2_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2prof
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
# the headline command, un-profiled and under the kernel trace (warm-up launch
# as long as the timed one, so every launch of the kernel has the same length)
python bench.py --secondary-batch 0 --warmup 1000 --cpu-seconds 0 > $O/bench_B4096.json 2>/dev/null
python bench.py --batch 1024 --secondary-batch 0 --warmup 1000 --cpu-seconds 0 > $O/bench_B1024.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B4096 -o bench -- python bench.py --secondary-batch 0 --warmup 1000 --cpu-seconds 0 > $O/prof_B4096.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B1024 -o bench -- python bench.py --batch 1024 --secondary-batch 0 --warmup 1000 --cpu-seconds 0 > $O/prof_B1024.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_persub -o bench -- python bench.py --launch-mode per_substep --secondary-batch 0 --steps 200 --warmup 200 --preheat-ms 50 --cpu-seconds 0 > $O/prof_persub.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o bench -- python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --secondary-batch 0 --steps 100 --warmup 100 --preheat-ms 50 --cpu-seconds 0 > $O/prof_stream.log 2>&1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
for b in 4096 1024; do
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq_B$b -o pmc -- python bench.py --batch $b --secondary-batch 0 --steps 1000 --warmup 0 --preheat-ms 0 --min-timed-ms 0 --cpu-seconds 0 > $O/pmc_sq_B$b.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_B$b -o pmc -- python bench.py --batch $b --secondary-batch 0 --steps 1000 --warmup 0 --preheat-ms 0 --min-timed-ms 0 --cpu-seconds 0 > $O/pmc_${c}_B$b.log 2>&1
  done
done
for b in 4096 1024; do
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk_B$b -o pmc -- python bench.py --batch $b --secondary-batch 0 --steps 1000 --warmup 1000 --preheat-ms 200 --min-timed-ms 0 --cpu-seconds 0 > $O/pmc_clk_B$b.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_persub -o pmc -- python bench.py --launch-mode per_substep --secondary-batch 0 --steps 20 --warmup 0 --preheat-ms 0 --min-timed-ms 0 --cpu-seconds 0 > $O/pmc_${c}_persub.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_stream -o pmc -- python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --secondary-batch 0 --steps 20 --warmup 0 --preheat-ms 0 --min-timed-ms 0 --cpu-seconds 0 > $O/pmc_${c}_stream.log 2>&1
done
python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --secondary-batch 0 --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_stream.json 2>/dev/null
python bench.py --launch-mode per_substep --secondary-batch 0 --cpu-seconds 0 > $O/bench_persub.json 2>/dev/null
timeout 300 python profiles/tools/trace_phases.py 1024 2>&1 | tail -9 > $O/phases_B1024.txt
timeout 300 python profiles/tools/trace_phases.py 4096 2>&1 | tail -9 > $O/phases_B4096.txt
find $O -name "*_kernel_trace.csv" -size +4M -delete
find $O -name "*.db" -delete
du -sh $O | tail -1
cat $O/smoke.txt
ls $O
