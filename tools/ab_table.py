"""Reduce the output of an A/B loop over tools/bench_conv.py (sections '==== F4=<f>' / '== <variant>') to best-of table."""
import collections, re, sys
cur = f4 = None
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open(sys.argv[1]):
    m = re.match(r'==== (.*)', line)
    if m: f4 = m.group(1).strip(); continue
    m = re.match(r'== (\w+)', line)
    if m: cur = m.group(1); continue
    m = re.match(r'(.{26})\s*(\w+)\s+([0-9.]+) ms', line)
    if m: tab[(f4, m.group(1).strip(), m.group(2))][cur].append(float(m.group(3)))
    if 'passed' in line or 'failed' in line or 'gpurun]' in line: print(line.strip())
for k, v in tab.items():
    print(k, {n: min(x) for n, x in v.items()})
