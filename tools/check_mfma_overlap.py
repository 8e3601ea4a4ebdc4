#!/usr/bin/env python3
"""Build-time guard: hipcc (ROCm 7.2) can allocate an MFMA's destination so that it PARTIALLY overlaps its
SrcC accumulator (seen after SLP vectorisation of an epilogue); the multi-pass MFMA then reads rows it has
already overwritten.  Scans device assembly for such instructions.  usage: check_mfma_overlap.py file.s ..."""
import re
import sys

PAT = re.compile(r'v_mfma_\w+ ([av])\[(\d+):(\d+)\], [^,]+, [^,]+, ([av])\[(\d+):(\d+)\]')
bad = 0
for path in sys.argv[1:]:
    for n, line in enumerate(open(path), 1):
        m = PAT.search(line)
        if not m:
            continue
        if m.group(1) != m.group(4):
            continue
        d0, d1, c0, c1 = int(m.group(2)), int(m.group(3)), int(m.group(5)), int(m.group(6))
        if (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
            bad += 1
            print('%s:%d: %s' % (path, n, line.strip()))
if bad:
    sys.exit('%d MFMA instruction(s) with partially overlapping vDst / SrcC' % bad)
