import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
sys.argv = ["sweep.py"]
import sweep
sweep.bf16_case(4096, 4096, 64, 64, "dbg=" + os.environ.get("TPP_HIP_CHAIN_DBG", "0"), force=23)
