#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch per kernel from a hipcc -save-temps gfx950 .s file."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
names = re.findall(r'^\s*\.amdhsa_kernel (\S+)', s, re.M)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    blk = s[s.index('.amdhsa_kernel ' + n):]
    blk = blk[:blk.index('.end_amdhsa_kernel')]
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\d+)', blk).group(1)
    d = re.sub(r'^void ', '', d)
    d = re.sub(r'\(.*$', '', d)
    print(f"{d[:64]:64s} vgpr={g('next_free_vgpr'):>4} sgpr={g('next_free_sgpr'):>4} lds={g('group_segment_fixed_size'):>6} scratch={g('private_segment_fixed_size')}")
