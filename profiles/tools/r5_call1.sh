#!/bin/bash
# Round-5 GPU call 1: the VALU diet of eval_rhs (packed-clamp relu on scaled activations, exact_div as a
# branch, per-stage constants in SGPRs, packed projection): smoke + the whole GPU test tier + the default
# bench line, then the instruction mix of the adaptive kernel next to the fixed-step one (PMC).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
B="python bench.py --secondary-batch 0 --cpu-seconds 0 --steps 20 --warmup 5 --min-timed-ms 50 --preheat-ms 0 --configs adaptive_rk23,adaptive_ks_n256_b1024"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mix1 -o run -- $B > $O/pmc_mix1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mix2 -o run -- $B > $O/pmc_mix2.log 2>&1
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*.db" -delete
python - <<'PY'
import csv, glob, collections
for d in ('pmc_mix1', 'pmc_mix2'):
  for f in glob.glob('gpurun_out/r5a/%s/**/*counter_collection.csv' % d, recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f)):
      k = row['Kernel_Name'][:90]
      acc[k][row['Counter_Name']] += float(row['Counter_Value'])
      n[(k, row['Counter_Name'])] += 1
    for k in acc:
      print(d, k)
      for c, v in sorted(acc[k].items()):
        print('    %-28s total %.4e  dispatches %d  per dispatch %.4e' % (c, v, n[(k, c)], v / n[(k, c)]))
PY
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5a/bench_default.json'))
print('headline', r['value'], r['roofline']['frac'], 'secondary', r.get('secondary', {}).get('roofline', {}).get('frac'))
for k, v in r.get('configs', {}).items():
  if isinstance(v, dict):
    print(' ', k, v.get('value'), (v.get('roofline') or {}).get('frac'))
PY
