"""Registers / LDS / scratch of the kernels whose mangled name contains one of the given substrings, from a -save-temps .s"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count: (\d+)", s, re.S):
    if any(k in m.group(2) for k in sys.argv[2:]):
        print(m.group(2)[:80], "lds", m.group(1), "scratch", m.group(3), "sgpr", m.group(4), "vgpr", m.group(5), "spill", m.group(6))
