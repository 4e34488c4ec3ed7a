#!/bin/bash
# Round-4 GPU call 5: fused per-step streaming kernel (tests + bench).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_integrate.py -m gpu -q -x -k "streaming or launch_modes or fork_join" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py --secondary-batch 0 --cpu-seconds 0 --min-timed-ms 200 --steps 200 --warmup 20 --configs stream_fixed,stream_fixed_per_step > $O/stream.json 2>$O/err.log
python - <<'PY'
import json
r = json.load(open('gpurun_out/r4e/stream.json'))
for k, v in r['configs'].items():
  print(k, '{:.3e} pts/s'.format(v['value']), 'GB/s', round(v['achieved'], 1), 'frac', round(v['frac'], 4),
        {kk: round(v[kk], 4) for kk in ('frac_algorithmic', 'frac_of_copy_rate') if kk in v}, v['kernel'], v['finite'])
PY
