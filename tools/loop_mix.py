"""Instruction mix of the MFMA-densest basic block of a kernel in an assembly listing.
usage: loop_mix.py file.s <substring of the mangled kernel name>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and sys.argv[2] in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
blocks, cur = [], ('entry', [])
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    else:
        cur[1].append(l)
blocks.append(cur)
best = max(blocks, key=lambda b: sum('v_mfma' in x for x in b[1]))
cnt = collections.Counter()
for line in best[1]:
    line = line.strip()
    if not line or line[0] in ';.':
        continue
    cnt[line.split()[0]] += 1
print(best[0], 'instructions', sum(cnt.values()))
for k, v in cnt.most_common(60):
    print('%5d %s' % (v, k))
