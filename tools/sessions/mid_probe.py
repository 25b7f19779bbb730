import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import sweep
for m in (1024, 1280, 1536, 1792, 2048):
    for v in (16, 17):
        sweep.bf16_case(m, 1024, 64, 16, force=v, tag="forced v%d" % v)
for (m, n) in ((1536, 1536), (1024, 2048)):
    for v in (16, 17):
        sweep.bf16_case(m, n, 64, 16, force=v, tag="forced v%d" % v)
for m in (1024, 1280, 1536, 2048, 3072):
    for v in (0, 3, 4):
        sweep.f32_case(m, 1024, 64, 16, force=v, tag="forced v%d" % v)
