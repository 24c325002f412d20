#!/bin/bash
# round 6, GPU call 1: suite, bench line, tie-free screening, dry-collective, launch counts
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log | cut -c1-300
timeout 600 python bench.py > $O/bench.log 2>&1; tail -n 1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6a/bench_line.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'whole', d['roofline']['whole_step']['frac'])
print({k:(v.get('ms') or v.get('ms_per_step') or v.get('graph') or v.get('error')) for k,v in d['secondary'].items()})
print('hbm', {k:(v['frac'], v['ms_per_step']) for k,v in d['hbm_bound'].items()})
PY
timeout 900 python tests/screen_tie_free.py train:3:2:512:l1_loss:1:1170 train:3:2:512:l1_loss:1:1375 train:3:2:512:l1_loss:1:1027 train:3:2:512:l1_loss:1:1391 train:3:2:512:l1_loss:1:1231 train:3:2:512:l1_loss:1:1006 \
  train:3:2:1000:l2_loss:1:2238 train:3:2:1000:l2_loss:1:2262 train:3:2:1000:l2_loss:1:2169 train:3:2:1000:l2_loss:1:2020 train:3:2:1000:l2_loss:1:2191 train:3:2:1000:l2_loss:1:2102 \
  train:8:2:512:l1_loss:1:3017 train:8:2:512:l1_loss:1:3208 train:8:2:512:l1_loss:1:3049 train:8:2:512:l1_loss:1:3144 train:8:2:512:l1_loss:1:3213 train:8:2:512:l1_loss:1:3181 \
  nefnet2:3:2:512:l1_loss:0:4283 nefnet2:3:2:512:l1_loss:0:4082 nefnet2:3:2:512:l1_loss:0:4116 nefnet2:3:2:512:l1_loss:0:4176 nefnet2:3:2:512:l1_loss:0:4160 nefnet2:3:2:512:l1_loss:0:4108 > $O/screen.log 2>&1; tail -n 30 $O/screen.log | cut -c1-250
timeout 600 python bench.py --leads 8 --dry-collective 8 --steps 10 --warmup 3 > $O/dry8.log 2>&1; tail -n 1 $O/dry8.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats -d $O/graph -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/graph.log 2>&1
python tools/rocprof_summary.py $(find $O/graph -name "*results.db" | head -1) $O/r06_kernel_stats_graph.md 4 1 > /dev/null
NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/one -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/one.log 2>&1
python tools/rocprof_summary.py $(find $O/one -name "*results.db" | head -1) $O/r06_kernel_stats_serialized.md 4 1 > /dev/null
rm -rf $O/graph $O/one
tail -n 3 $O/r06_kernel_stats_graph.md; python - <<'PY'
import re
for f in ('gpurun_out/r6a/r06_kernel_stats_graph.md','gpurun_out/r6a/r06_kernel_stats_serialized.md'):
    rows=[l.split('|') for l in open(f) if l.startswith('| `')]
    calls=sum(int(r[2]) for r in rows); small=sum(float(r[3]) for r in rows if float(r[4])<100); nsmall=sum(int(r[2]) for r in rows if float(r[4])<100)
    aten=[(r[1].strip()[:60], int(r[2])) for r in rows if 'at::' in r[1] or 'rocclr' in r[1]]
    print(f, 'launches/step', calls/3, 'small (<100us avg) launches/step', nsmall/3, 'ms/step', small/3/1e3, 'aten', aten)
PY
