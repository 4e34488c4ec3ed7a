#!/bin/bash
# Round 6, call 9: issue priority by phase (rhs_mfma.h DDD_PRIO_PHASES): product = 2 (VALU phases
# raised), variants prio0 (no priorities: rounds 1-5) and prio3 (only the steady middle of the
# matrix layers at 0) -- the same legs, twice each, interleaved.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
L="--cpu-seconds 0 --secondary-batch 1024 --configs adaptive_rk23,adaptive_kdv_n64_b4096,kdv_n64_b4096,ks_n256_b8192"
for rep in 1 2; do
  timeout 600 python bench.py $L --library prio0 > $O/bench_prio0_$rep.json 2> $O/bench_prio0.err
  timeout 600 python bench.py $L > $O/bench_prio2_$rep.json 2> $O/bench_prio2.err
  timeout 600 python bench.py $L --library prio3 > $O/bench_prio3_$rep.json 2> $O/bench_prio3.err
done
python - <<'PY'
import json
for tag in ('prio0_1', 'prio0_2', 'prio2_1', 'prio2_2', 'prio3_1', 'prio3_2'):
  d = json.load(open('gpurun_out/r6i/bench_%s.json' % tag))
  row = [tag, 'headline %.4f' % d['roofline']['frac'], 'b1024 %.4f' % d['secondary']['frac']]
  for k, v in d['configs'].items():
    row.append('%s %.4f' % (k, v['frac']))
  print(' | '.join(row))
PY
