cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-extra --no-cpu-baseline --cg-iters 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('plain', d['value'], d['ms_per_step'], d['clock_ramp']['ms_per_step_by_10'][3:], d['clock_ramp'].get('ms_per_step_by_10_after_placement_ab'), [p['ms'] for p in r['placement_ab']['places']], r['memory_classes']['arena']['acquired_gib'])
"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kk -o kk -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --cg-iters 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('rocprof', d['value'], d['ms_per_step'], d['clock_ramp']['ms_per_step_by_10'][3:], d['clock_ramp'].get('ms_per_step_by_10_after_placement_ab'), [p['ms'] for p in r['placement_ab']['places']], r['memory_classes']['arena']['acquired_gib'])
"
