#!/bin/bash
# Round-5 GPU call 3: are the two wavefronts of a SIMD in lockstep at the layer boundaries?  The probe
# library's start stagger (odd hardware wave slots sleep `stagger` x 8128 cycles before their first
# evaluation) and static priority, on the round-5 headline kernel.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c
rm -rf $O; mkdir -p $O
common="--configs none --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 400 --steps 1000 --warmup 100"
run() {  # tag, flags...
  local tag=$1; shift
  python bench.py $common "$@" > $O/$tag.json 2>>$O/err.log
}
run product
run probe_plain --debug-option prio_split=0
for st in 1 2 3 5; do
  run probe_stagger${st} --debug-option prio_split=1 --debug-option stagger=$st
done
run probe_prio --debug-option prio_split=5
run probe_prio_stagger2 --debug-option prio_split=5 --debug-option stagger=2
run probe_blockprio_stagger2 --debug-option prio_split=3 --debug-option stagger=2
run product_b8192 --batch 8192
run probe_stagger2_b8192 --batch 8192 --debug-option prio_split=1 --debug-option stagger=2
run product_kdv --equation kdv
run probe_stagger2_kdv --equation kdv --debug-option prio_split=1 --debug-option stagger=2
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c/*.json')):
  try:
    r = json.load(open(f))
  except Exception as e:
    print(f, 'FAILED', e); continue
  print('{:32s} {:16s} {:9.3e} pts/s {:5.1f} % finite={}'.format(
      f.split('/')[-1], r['config']['kernel'], r['value'], 100 * r['roofline']['fp32_frac'], r['config']['finite']))
PY
tail -5 $O/err.log
