run() { timeout 300 python bench.py --secondary-batch 0 --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(c['batch_per_gpu'], c['debug_options'], '%.2f TF'%r['achieved'], '%.3f'%r['frac'])"; }
run
run --debug-option prio_split=1 --debug-option stagger=1
run --debug-option prio_split=1 --debug-option stagger=2
run --debug-option prio_split=1 --debug-option stagger=3
run --debug-option prio_split=5 --debug-option stagger=0
run --debug-option prio_split=5 --debug-option stagger=1
run --batch 8192
run --batch 8192 --debug-option prio_split=1 --debug-option stagger=1
