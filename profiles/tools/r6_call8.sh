#!/bin/bash
# Round 6, call 8: the command ring as shipped -- tests, bench leg, stamps of its commands
# (variant build with -DDDD_RING_TRACE=1).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ring.py -x -q > $O/ring_tests.log 2>&1; tail -5 $O/ring_tests.log
timeout 600 python -m pytest tests/test_gpu_integrate.py -x -q -k "fork_join or rk_substep" > $O/fork_tests.log 2>&1; tail -3 $O/fork_tests.log
timeout 600 python bench.py --cpu-seconds 0 --secondary-batch 0 --steps 20 --warmup 5 --min-timed-ms 50 --configs rk_substep_external > $O/bench_ring.json 2> $O/bench_ring.err; tail -3 $O/bench_ring.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6h/bench_ring.json'))['configs']['rk_substep_external']
for b in ('b4096', 'b8192'):
  e = d[b]
  print(b, {k: (round(v['frac'], 4), round(v['us_per_substep_call'], 2)) for k, v in e.items() if isinstance(v, dict) and 'frac' in v}, e.get('ring'), e['drivers_bit_identical'], e['equals_ddd_integrate_fixed_per_substep'])
PY
for b in 4096 2048; do timeout 900 python profiles/tools/ring_trace.py $b 2>&1 | grep -v amdgpu.ids > $O/ring_trace_b$b.txt; head -14 $O/ring_trace_b$b.txt; done
