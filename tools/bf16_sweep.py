#!/usr/bin/env python3
"""bf16_sweep - whole-layer bf16 calls of the reference's benchmark shape set (benchmarks/config/matmul/*.json, fc/*.json: the
bf16 dp2 / dp4 rows) over every bf16 tile family, VNNI-2 and VNNI-4: what pick_bf16_lw_tile / plan_gemm (brgemm_f32.hip) are
fitted to ("give bf16 the sweep f32 got"). One tpp_replay process, one case per line; the tile-invoke form of
every shape (runtime's choice) is measured next to the whole-layer rows.
usage: python tools/bf16_sweep.py [-n 300] > profiles/r06_bf16_sweep.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from refbench import SHAPES  # noqa: E402  (M, N, K, tiles) - the 17 shapes of the reference's configs

# forced variants (brgemm_f32.hip GemmVariant): name, (bm, bn) the shape must divide, VNNI-4 twin (or None)
VARIANTS = [(16, "fast64x64", (64, 64), None), (17, "dma128x128", (128, 128), None), (19, "small32x32k4", (32, 32), None),
            (20, "lw32x64k2", (32, 64), 28), (21, "lw64x64", (64, 64), 29), (22, "lw64x128", (64, 128), 30), (23, "lw128x128", (128, 128), 31)]
n_iter = int(sys.argv[sys.argv.index("-n") + 1]) if "-n" in sys.argv else 300
cases = [("warm", 0, "", ["--batch", "1024", "--layers", "1024,1024", "--whole-layer", "--bf16", "-n", "2000"])]
for (M, N, K, tl) in SHAPES:
    for vf in (2, 4):
        base = ["--batch", str(M), "--layers", "%d,%d" % (K, N), "--kernel", "args", "--bf16", "--vnni", str(vf), "-n", str(n_iter)]
        for v, vn, (bm, bn), v4 in VARIANTS:
            if M % bm or N % bn:
                continue
            fv = v if vf == 2 else v4
            if fv is None:
                continue
            cases.append(((M, N, K), vf, vn, base + ["--whole-layer", "--variant", str(fv)]))
        cases.append(((M, N, K), vf, "auto", base + ["--whole-layer"]))
        cases.append(((M, N, K), vf, "tiles %d,%d,%d" % tl, base + ["--tiles", "%d,%d,%d" % tl, "--queue", "1"]))
with tempfile.NamedTemporaryFile("w", suffix=".cases", delete=False) as f:
    for c in cases:
        f.write(" ".join(c[3]) + "\n")
r = subprocess.run([os.path.join(ROOT, "tools", "tpp_replay"), "--cases", f.name], capture_output=True, text=True, timeout=3000)
os.unlink(f.name)
res = re.findall(r"mean ([0-9.]+) us .*?([0-9.]+) GFLOP/s .*kernel (.*)", r.stderr)
if len(res) != len(cases):
    sys.stderr.write(r.stderr[-3000:])
    raise SystemExit("%d cases, %d results" % (len(cases), len(res)))
print("# whole-layer bf16 C += A W (tpp_replay, -n %d): us per call by forced tile family, VNNI-2 / VNNI-4; the runtime's own choice (auto) and "
      "the tile-invoke form (tile queue) beside them; 2500 TF peak" % n_iter)
last = None
for c, (us, gf, kn) in zip(cases[1:], res[1:]):
    if c[0] != last:
        last = c[0]
        print("== M%d N%d K%d  (%.3f GFLOP)" % (c[0] + (2e-9 * c[0][0] * c[0][1] * c[0][2],)))
    print("  vnni%d %-16s %8.2f us  %6.3f of peak  %s" % (c[1], c[2], float(us), float(gf) / 2.5e6, kn.strip()), flush=True)
