"""Dev script: print the innermost MFMA loop of one kernel from a hipcc -save-temps .s file (MFMAs abbreviated)."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if re.search(pat, m.group(1))]
name = names[0]
start = s.index(name + ':')
end = s.index('s_endpgm', start)
lines = [l.strip() for l in s[start:end].split('\n')]
lines = [l for l in lines if l and not l.startswith(';')]
# the steady-state loop body = from the label that the last backward branch targets to that branch
labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(lines):
    m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.match(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = lines[labels[m.group(1)]:i + 1]
        n = sum(1 for b in body if b.startswith('v_mfma'))
        if n and (best is None or n >= best[0]):
            best = (n, labels[m.group(1)], i)
n, a, b = best
print(f'{name}: loop lines {a}..{b}, {n} MFMAs')
run = 0
for l in lines[a:b + 1]:
    if l.startswith('v_mfma'):
        run += 1
        continue
    if run:
        print(f'    [{run} x MFMA]')
        run = 0
    print(l[:120])
if run:
    print(f'    [{run} x MFMA]')
