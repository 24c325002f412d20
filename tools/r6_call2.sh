#!/bin/bash
# round 6, GPU call 2: suite (tie-free fixtures, fused BN finals, async step words), bench A/B, small-shape profile (graph replay)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -n 12 $O/pytest.log | cut -c1-300
timeout 600 python bench.py > $O/bench.log 2>&1; tail -n 1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6b/bench_line.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'whole', d['roofline']['whole_step']['frac'], 'tail', d.get('h2_tail_sites'))
print({k:(v.get('ms') or v.get('ms_per_step') or v.get('graph') or v.get('error')) for k,v in d['secondary'].items()})
print('hbm', {k:(v['frac'], v['ms_per_step']) for k,v in d['hbm_bound'].items()})
PY
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --no-secondary --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('repeat', d['ms_per_step'])"; done
# small shape, graph replay: launches per step
timeout 600 rocprofv3 --kernel-trace --stats -d $O/small -o t -- python bench.py --batch 32 --len 512 --steps 9 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/small.log 2>&1
python tools/rocprof_summary.py $(find $O/small -name "*results.db" | head -1) $O/r06_small32_graph_kernel_stats.md 11 2 > /dev/null
rm -rf $O/small
python - <<'PY'
f='gpurun_out/r6b/r06_small32_graph_kernel_stats.md'
rows=[l.split('|') for l in open(f) if l.startswith('| `')]
calls=sum(int(r[2]) for r in rows); tot=sum(float(r[3]) for r in rows)
print('small shape: launches/step', calls/9, 'kernel ms/step', tot/9/1e3)
for r in sorted(rows, key=lambda r:-float(r[3]))[:25]: print(r[1].strip()[:70], r[2], r[3], r[4])
PY
tail -n 1 $O/small.log | cut -c1-300
