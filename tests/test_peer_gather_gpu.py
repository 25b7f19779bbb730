"""The peer-store all-gather (csrc/peer_gather.hip, tpp-mlir_amd/peer.py) with TWO ranks on ONE device: two processes on cuda:0
exchange hipIpcMemHandles over a gloo group and store their row blocks into each other's output buffers, six steps with fresh
inputs (double-buffered outputs, ready / landed flags), bit-identical to the unsharded MLP and to a gloo all_gather of the
same blocks (tests/peer_worker.py). The 8-GPU run is the driver's; this is the same code path with the peer mappings
pointing into the same HBM."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_peer_store_gather_ranks_on_one_device(world):
    port = 29611 + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "OK" in o, "rank %d failed:\n%s" % (r, o[-3000:])
    print("\n" + "".join(l + "\n" for o in outs for l in o.splitlines() if "us per step" in l))
