#!/usr/bin/env python3
"""reads a TPP_HIP_CHAIN_STAMPS file (workgroup, layer, 8 s_memrealtime stamps at 100 MHz) and prints, per layer, the phases of the
chain kernel in microseconds relative to the earliest stamp of the launch: median / max over the workgroups"""
import sys
import numpy as np

rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip()]
a = np.array(rows, dtype=np.int64)
t0 = a[:, 2:][a[:, 2:] > 0].min()
names = ["layer start", "chunk0 ready", "K loop done", "stores issued", "drained+S1", "A: wait", "A: arrived", "A: requested"]
for l in sorted(set(a[:, 1])):
    sel = a[a[:, 1] == l]
    out = []
    for i in range(8):
        v = sel[:, 2 + i]
        v = v[v > 0]
        if len(v):
            us = (v - t0) / 100.0
            out.append("%s %.2f/%.2f/%.2f" % (names[i], us.min(), np.median(us), us.max()))
    print("layer %d (min/med/max us): " % l + " | ".join(out))
