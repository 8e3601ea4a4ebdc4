#!/usr/bin/env python3
"""Build-time guard on the MFMA register tuples hipcc (ROCm 7.2) allocates.

A multi-pass MFMA whose destination PARTIALLY overlaps its SrcC accumulator reads rows it has already overwritten
when the destination is wider than 4 VGPRs (LLVM marks exactly those forms early-clobber; the 32x32 forms).  For the
4-VGPR destination of v_mfma_f32_16x16x4_f32 -- the only MFMA these kernels use -- a partial overlap is legal, and
tools/mfma_overlap_probe.hip confirms it on gfx950 (chains sliding up / down by two registers, with and without
back-to-back forwarding, 8 waves per SIMD on every CU: bit-identical to the tied form).  Round 1 forbade every partial
overlap after a wrong sum that went away with -fno-slp-vectorize; the probe shows the overlap was not its cause.

Scans device assembly for a partially overlapping destination wider than 4 registers.
usage: check_mfma_overlap.py file.s ..."""
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
        if d1 - d0 + 1 <= 4:
            continue
        if (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
            bad += 1
            print('%s:%d: %s' % (path, n, line.strip()))
if bad:
    sys.exit('%d MFMA instruction(s) with a wide destination partially overlapping SrcC' % bad)
