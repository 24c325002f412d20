#!/bin/bash
# round 6 final measurements: profiles (tools/profile_round.sh r06), graph-mode launch counts, bench lines (default, 8-lead shard,
# dry-collective), small-shape profile
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06 2>&1 | tail -n 12
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
python tools/sq_to_json.py $O/sq/sq.md $O/sq_k7.json "profiles/r06_sq_counters_k7.md (rocprofv3 --pmc pass of tools/bench_conv.py 'enc k7'; profiled clocks run ~5 % under un-profiled ones)" || true
timeout 600 rocprofv3 --kernel-trace --stats -d $O/graph -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary > $O/graph.log 2>&1
python tools/rocprof_summary.py $(find $O/graph -name "*results.db" | head -1) $O/r06_kernel_stats_graph_replay.md 7 3 > /dev/null
rm -rf $O/graph
timeout 600 rocprofv3 --kernel-trace --stats -d $O/small -o t -- python bench.py --batch 32 --len 512 --steps 9 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/small.log 2>&1
python tools/rocprof_summary.py $(find $O/small -name "*results.db" | head -1) $O/r06_small32_graph_kernel_stats.md 11 2 > /dev/null
rm -rf $O/small
timeout 900 python bench.py > $O/bench.log 2>&1; tail -n 1 $O/bench.log > $O/r06_bench_line.json
timeout 600 python bench.py --leads 8 --no-cpu-baseline --no-secondary > $O/bench8.log 2>&1; tail -n 1 $O/bench8.log > $O/r06_bench_line_8lead_shard.json
timeout 600 python bench.py --leads 8 --dry-collective 8 --steps 10 --warmup 3 > $O/dry8.log 2>&1; tail -n 1 $O/dry8.log > $O/r06_dry_collective_8lead.json
timeout 600 python bench.py --dry-collective 8 --steps 20 --warmup 5 > $O/dry3.log 2>&1; tail -n 1 $O/dry3.log > $O/r06_dry_collective_3lead.json
python - <<'PY'
import json
for f in ('r06_bench_line.json','r06_bench_line_8lead_shard.json'):
    d=json.load(open('gpurun_out/r06/'+f)); print(f, d['ms_per_step'], d['value'], d['roofline'] and d['roofline']['frac'], d['roofline'] and d['roofline']['whole_step']['frac'])
d=json.load(open('gpurun_out/r06/r06_bench_line.json'))
print({k:(v.get('ms') or v.get('ms_per_step') or v.get('graph') or v.get('error')) for k,v in d['secondary'].items()})
print('hbm', {k:(v['frac'], v['ms_per_step']) for k,v in d['hbm_bound'].items()}); print('sum hbm', sum(v['ms_per_step'] for v in d['hbm_bound'].values()))
for f in ('r06_dry_collective_8lead.json','r06_dry_collective_3lead.json'):
    d=json.load(open('gpurun_out/r06/'+f)); print(f, {k:(v['ms_per_step'], v.get('allreduce_ms_exposed')) for k,v in d['schedules'].items()})
for f in ('r06_kernel_stats_graph_replay.md','r06_small32_graph_kernel_stats.md','r06_kernel_stats_serialized.md'):
    rows=[l.split('|') for l in open('gpurun_out/r06/'+f) if l.startswith('| `')]
    hdr=open('gpurun_out/r06/'+f).readline()
    import re
    k=int(re.search(r'(\d+) steady-state', hdr).group(1))
    calls=sum(int(r[2]) for r in rows); small=sum(float(r[3]) for r in rows if float(r[4])<100); nsmall=sum(int(r[2]) for r in rows if float(r[4])<100)
    aten=[(r[1].strip()[:50], int(r[2])) for r in rows if 'at::' in r[1] or 'rocclr' in r[1]]
    print(f, 'launches/step', round(calls/k,1), 'small', round(nsmall/k,1), 'small ms/step', round(small/k/1e3,3), 'aten', aten)
PY
ls $O
