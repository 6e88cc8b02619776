#!/usr/bin/env python3
"""Static instruction counts and register use per kernel of a gfx950 assembly file (hipcc -S --offload-device-only)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name = m.group(1)
    if 'k_' not in name: continue
    c = Counter()
    for l in m.group(2).split('\n'):
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'): continue
        op = l.split()[0]
        c['valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else
          'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'] += 1
    i = s.find('.amdhsa_kernel ' + name)
    meta = s[i:s.find('.end_amdhsa_kernel', i)]
    vg = re.search(r'next_free_vgpr (\d+)', meta).group(1)
    print('%-50s vgpr %3s  %s' % (name[-40:], vg, dict(c)))
