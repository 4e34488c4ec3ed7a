#!/bin/bash
# Round 5, call 17: rocprofv3 kernel stats of the kernels that are new in the default bench line (wide flavour, 7 x 64 tower).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5u; rm -rf $O; mkdir -p $O
B="python bench.py --configs none --secondary-batch 0 --cpu-seconds 0 --steps 200 --warmup 200"
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wide -o run -- $B --equation ks --hparams '{"coefficient_grid_min_size": 9}' > $O/wide.log 2>&1
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k7f64 -o run -- $B --hparams '{"kernel_size": 7, "filter_size": 64}' > $O/k7f64.log 2>&1
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3 -o run -- $B --hparams '{"kernel_size": 3}' > $O/k3.log 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
for t in wide k7f64 k3; do echo "== $t"; head -4 $(find $O/$t -name "*kernel_stats.csv" | head -1) | cut -c1-300; done
