cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print(c['equation'], c['num_points'], c['batch_per_gpu'], c['scheme'], '%.3e'%d['value'], '%.2f TF'%r['achieved'], '%.3f'%r['frac'], c['kernel'])"; }
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
for spec in 0 1; do export DDD_NO_SPEC=$spec; echo "DDD_NO_SPEC=$spec";
run --batch 1024; run --batch 2048; run --batch 4096
run --equation kdv --batch 4096; run --equation ks --num-points 256 --batch 8192 --steps 400; run --equation ks --batch 4096 --steps 400; run --non-conservative --batch 4096
done
unset DDD_NO_SPEC
timeout 300 python profiles/tools/trace_phases.py 1024 2>&1 | tail -7
