#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6v; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log | cut -c1-300; grep -n "tie-free\|full-size train" $O/pytest.log | cut -c1-250
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 2
timeout 900 python bench.py > $O/bench.log 2>&1; tail -n 1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6v/bench_line.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'traffic_ratio', d['roofline'].get('traffic_ratio'), d['roofline'].get('traffic_ratio_k3',{}).get('ratio'), 'whole', d['roofline']['whole_step']['frac'], 'tail', d.get('h2_tail_sites'), 'clamped', d['h2_clamped_waves'])
print({k:(v.get('ms') or v.get('ms_per_step') or v.get('graph') or v.get('error')) for k,v in d['secondary'].items()})
PY
