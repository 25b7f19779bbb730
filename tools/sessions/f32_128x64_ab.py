"""A/B of the f32 128x64 tiles: round-1 kernel (variant 3, DMA issued by the MFMA waves) against the loader-wave one (variant 10)"""
import os
import sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
sys.argv = ["sweep.py"]
import sweep
for (m, n, k, br) in ((4096, 4096, 64, 64), (2048, 2048, 64, 32), (2048, 1024, 64, 16), (8192, 8192, 64, 16)):
    for v in (3, 10, 3, 10):
        sweep.f32_case(m, n, k, br, force=v, tag="forced v%d" % v)
