"""Print per-kernel LDS / scratch / register figures from a hipcc -save-temps .s file (development aid)."""
import re
import sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    print(g('name')[:90], 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), 'vgpr', g('vgpr_count'),
          'agpr', g('agpr_count'), 'spill', g('vgpr_spill_count'))
