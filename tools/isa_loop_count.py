"""Instruction mix of a kernel's main loop (the widest backward branch) from hipcc -S output: tools/isa_loop_count.py file.s kernel-prefix"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(L) if l.startswith(sys.argv[2])][0]
en = [i for i, l in enumerate(L) if i > st and 's_endpgm' in l][0]
body = L[st:en]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
cnt = collections.Counter()
for l in body[best[1]:best[2] + 1]:
    l = l.strip()
    if not l or l[0] in ';.':
        continue
    op = l.split()[0]
    k = ('mfma' if op.startswith('v_mfma') else op if op.startswith(('ds_', 'buffer_', 'global_')) else op if op in ('s_waitcnt', 's_nop', 's_barrier')
         else 'salu' if op.startswith('s_') else 'valu' if op.startswith('v_') else op)
    cnt[k] += 1
print(sum(cnt.values()), dict(cnt.most_common()))
for l in body:
    if 'NumVgprs' in l or 'ScratchSize' in l: print(l)
