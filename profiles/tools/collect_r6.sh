#!/bin/bash
# Round-6 measurement run (one gpurun call): the full bench line un-profiled,
# rocprofv3 kernel trace + stats of the same command and of the main configurations
# alone, PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs; SQ counters in their
# own).  Outputs under gpurun_out/r6prof/ (scratch); profiles/tools/summarize_r6.py
# turns them into the committed profiles/r6_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6prof
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_s20.json 2> $O/bench_driver_s20.err
B="python bench.py --configs none --secondary-batch 0 --cpu-seconds 0"
prof() {  # tag, extra rocprofv3 flags..., -- , bench flags
  local tag=$1; shift
  local flags=(); while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace "${flags[@]}" --output-format csv -d $O/$tag -o run -- $B "$@" > $O/$tag.log 2>&1
}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o run -- python bench.py --cpu-seconds 0 > $O/stats_default.log 2>&1
prof stats_B4096 --stats -- --warmup 1000
prof stats_B1024 --stats -- --batch 1024 --warmup 1000
prof stats_B256 --stats -- --batch 256 --warmup 1000
prof stats_persub --stats -- --launch-mode per_substep --steps 200 --warmup 200 --preheat-ms 50
prof stats_k7 --stats -- --hparams '{"kernel_size": 7}' --steps 200 --warmup 200
prof stats_f64 --stats -- --hparams '{"filter_size": 64}' --steps 200 --warmup 200
prof stats_stream_step --stats -- --equation kdv --baseline-stencils --launch-mode per_step --batch 262144 --steps 200 --warmup 20 --preheat-ms 50
prof stats_lean --stats -- --hparams '{"num_layers": 1}' --warmup 1000
prof stats_B512 --stats -- --batch 512 --warmup 1000
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_weno -o run -- python profiles/tools/weno_exact_bench.py > $O/weno_exact.txt 2> $O/stats_weno.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_adaptive -o run -- python bench.py --cpu-seconds 0 --secondary-batch 0 --steps 20 --warmup 5 --min-timed-ms 50 --configs adaptive_rk23,adaptive_ks_n256_b1024,adaptive_ks_n256_b8192,adaptive_rk23_b256,rk_substep_external > $O/stats_adaptive.log 2>&1
short="--warmup 0 --preheat-ms 0 --min-timed-ms 0"
for c in FETCH_SIZE WRITE_SIZE; do
  prof pmc_${c}_B4096 --pmc $c -- --steps 1000 $short
  prof pmc_${c}_B1024 --pmc $c -- --batch 1024 --steps 1000 $short
  prof pmc_${c}_B256 --pmc $c -- --batch 256 --steps 1000 $short
  prof pmc_${c}_lean --pmc $c -- --hparams '{"num_layers": 1}' --steps 1000 $short
  prof pmc_${c}_stream_step --pmc $c -- --equation kdv --baseline-stencils --launch-mode per_step --batch 262144 --steps 20 $short
  prof pmc_${c}_kdv --pmc $c -- --equation kdv --steps 1000 $short
  prof pmc_${c}_ks256 --pmc $c -- --equation ks --num-points 256 --batch 8192 --steps 400 $short
done
prof pmc_sq_B4096 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- --steps 1000 $short
prof pmc_sq_lean --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -- --hparams '{"num_layers": 1}' --steps 1000 $short
prof pmc_clk_B4096 --pmc GRBM_GUI_ACTIVE -- --steps 1000 --warmup 1000 --preheat-ms 200 --min-timed-ms 0
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*.db" -delete
du -sh $O | tail -1; cat $O/smoke.txt; tail -c 600 $O/bench_default.err
