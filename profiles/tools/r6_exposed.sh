cd $GRAFT_REPO_ROOT
for lib in "" "--library halfbd"; do
  for b in 1024 4096; do
    timeout 300 python bench.py --cpu-seconds 0 --secondary-batch 0 --configs none --batch $b --hparams '{"filter_size": 16}' --steps 1000 $lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', $b, d['config']['kernel'], 'ms/step %.5f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])
"
  done
done
