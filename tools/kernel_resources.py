#!/usr/bin/env python3
"""Register / LDS / scratch usage of kernels in a hipcc -S listing (development aid).  usage: kernel_resources.py file.s pattern..."""
import re
import sys
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)', s, re.S):
    name = m.group(1)
    if pats and not any(p in name for p in pats):
        continue
    blk = m.group(2)
    g = lambda k: (re.search(k + r':\s+(\d+)', blk) or [None, '?'])[1]
    print(name, 'vgpr', g(r'\.vgpr_count'), 'sgpr', g(r'\.sgpr_count'), 'lds', g(r'\.group_segment_fixed_size'), 'scratch',
          g(r'\.private_segment_fixed_size'), 'vspill', m.group(3), 'sspill', g(r'\.sgpr_spill_count'))
