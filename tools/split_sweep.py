#!/usr/bin/env python3
"""split_sweep - whole-layer f32 calls of the reference's skinny benchmark shapes over (tile variant, workgroups per tile):
what the cost model of choose_f32_split (brgemm_f32.hip) is fitted to. One tpp_replay process, one case per line.
usage: python tools/split_sweep.py [-n 300] > profiles/r05_split_sweep.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(128, 1024, 1024), (128, 1024, 4096), (128, 3072, 768), (128, 4096, 1024), (128, 768, 2304), (128, 768, 3072), (128, 768, 768),
          (256, 1024, 1024), (256, 768, 3072), (256, 768, 768), (256, 3072, 768), (1024, 352, 512), (1024, 512, 256), (512, 1024, 1024)]
VARIANTS = {6: "64x64k2", 7: "64x32k4", 9: "32x32k4", 11: "32x16k4"}
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]
n_iter = int(sys.argv[sys.argv.index("-n") + 1]) if "-n" in sys.argv else 300
cases = [("warm", 0, 0, ["--batch", "1024", "--layers", "1024,1024", "--whole-layer", "-n", "2000"])]
for (M, N, K) in SHAPES:
    for v, vn in VARIANTS.items():
        bm, bn = (64, 64) if v == 6 else (64, 32) if v == 7 else (32, 16) if v == 11 else (32, 32)
        if M % bm or N % bn:
            continue
        for S in ([1] if v == 11 else SPLITS):  # (the 16x16-block tiles have no split launch)
            if S > 1 and K // 64 // S < 2:
                continue
            cases.append(((M, N, K), vn, S, ["--batch", str(M), "--layers", "%d,%d" % (K, N), "--whole-layer", "--kernel", "args", "-n", str(n_iter),
                                             "--variant", str(v), "--split", str(S)]))
    cases.append(((M, N, K), "auto", -1, ["--batch", str(M), "--layers", "%d,%d" % (K, N), "--whole-layer", "--kernel", "args", "-n", str(n_iter)]))
with tempfile.NamedTemporaryFile("w", suffix=".cases", delete=False) as f:
    for c in cases:
        f.write(" ".join(c[3]) + "\n")
r = subprocess.run([os.path.join(ROOT, "tools", "tpp_replay"), "--cases", f.name], capture_output=True, text=True, timeout=3000)
os.unlink(f.name)
res = re.findall(r"mean ([0-9.]+) us .*?([0-9.]+) GFLOP/s .*kernel (.*)", r.stderr)
if len(res) != len(cases):
    sys.stderr.write(r.stderr[-3000:])
    raise SystemExit("%d cases, %d results" % (len(cases), len(res)))
print("# whole-layer f32 C += A W (tpp_replay, -n %d): us per call by tile variant and workgroups per tile (S); 157.3 TF peak" % n_iter)
last = None
for c, (us, gf, kn) in zip(cases[1:], res[1:]):
    if c[0] != last:
        last = c[0]
        print("== M%d N%d K%d  (%.3f GFLOP)" % (c[0] + (2e-9 * c[0][0] * c[0][1] * c[0][2],)))
    print("  %-8s S=%-3d %8.2f us  %6.3f of peak  %s" % (c[1], c[2], float(us), float(gf) / 157300.0, kn.strip() if c[1] == "auto" else ""), flush=True)
