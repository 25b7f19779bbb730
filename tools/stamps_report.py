#!/usr/bin/env python3
"""reads a TPP_HIP_CHAIN_STAMPS file (workgroup, layer, 8 s_memrealtime stamps at 100 MHz) and prints, per layer, the phases of the
chain kernel in microseconds relative to the earliest stamp of the launch: median / max over the workgroups"""
import sys
import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--chunks":
    # <file>.chunks: workgroup, loader (0 A / 1 B), steady-state iteration, shader-clock stamps [landed, barrier released, issued]
    rows = np.array([list(map(int, l.split())) for l in open(sys.argv[2]) if l.strip()], dtype=np.int64)
    for which, name in ((0, "A loader"), (1, "B loader")):
        r = rows[rows[:, 1] == which]
        per, wait_dma, wait_bar, issue = [], [], [], []
        for w in sorted(set(r[:, 0])):
            x = r[r[:, 0] == w]
            x = x[np.argsort(x[:, 2])]
            for i in range(1, len(x)):
                if x[i, 2] != x[i - 1, 2] + 1:
                    continue
                d = x[i, 3] - x[i - 1, 3]
                if 0 < d < 20000:  # (iterations of one layer: the seam between layers is not a chunk)
                    per.append(d)
                    wait_dma.append(x[i, 3] - x[i - 1, 5])
                    wait_bar.append(x[i, 4] - x[i, 3])
                    issue.append(x[i, 5] - x[i, 4])
        q = lambda v: "%5.0f / %5.0f / %5.0f" % tuple(np.percentile(v, [10, 50, 90])) if len(v) else "-"  # noqa: E731
        print("%s (%d iterations of %d workgroups; shader cycles, 10th / 50th / 90th percentile):" % (name, len(per), len(set(r[:, 0]))))
        print("  iteration (landed -> landed)          %s" % q(per))
        print("  issued(t-1) -> landed(t): own DMA     %s" % q(wait_dma))
        print("  landed -> barrier released: the others %s" % q(wait_bar))
        print("  released -> 16 (8) DMA instr. issued   %s" % q(issue))
    sys.exit(0)

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
