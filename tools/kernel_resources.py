"""Summarise `-Rpass-analysis=kernel-resource-usage` output (dev tool).  usage: python tools/kernel_resources.py build.log"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
names = []
rows = []
for b in txt.split('Function Name: ')[1:]:
    name = b.split('\n')[0].strip()
    def g(k):
        m = re.search(k + r': (\d+)', b); return int(m.group(1)) if m else -1
    rows.append((name, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
dem = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, d in zip(rows, dem):
    d = d.replace('pk::', '').replace('(SolveArgs)', '').replace('void ', '')
    print('%-48s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %d' % (d[:48], *r[1:]))
