"""Side-by-side table of the conv_breakdown sections in gpurun_out/ab_r2.txt (one column per configuration)."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ab_r2.txt').read()
tabs = {}
for b in txt.split('=== ')[1:]:
    name = b.split('\n')[0]
    rows = {}
    for line in b.split('\n'):
        m = re.match(r'\s*([\d.]+) ms\s+x\s*([\d.]+)\s+([\d.]+) TF/s\s+(\(.*\))', line)
        if m and m.group(4) not in rows:
            rows[m.group(4)] = float(m.group(1))
    tabs[name] = rows
names = list(tabs)
keys = sorted(tabs[names[0]], key=lambda k: -tabs[names[0]][k])
print(' | '.join(names))
print(' '.join(f'{sum(tabs[n].values()):6.2f}' for n in names), 'TOTAL (listed rows)')
for k in keys[:int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    print(' '.join(f'{tabs[n].get(k, 0):6.2f}' for n in names), k)
