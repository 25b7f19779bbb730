"""A starved chain launch degrades instead of ending the process: the persistent chain kernel
(xsmm_hip_fused_brgemm_chain_invoke as ONE launch) needs all of its workgroups resident; when another stream / process holds compute
units the hand-offs inside the launch time out (bounded, 50 ms). The library then runs the calls one by one - same results, return
value 0 - and stops launching chains for the rest of the process. Each scenario in its own process (tests/chain_starved_worker.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(scenario):
    if not os.path.exists(os.path.join(ROOT, "tools", "cu_hog.so")):
        pytest.skip("tools/cu_hog.so not built (tpp-mlir_amd/build.py build_tools)")
    env = {k: v for k, v in os.environ.items() if k not in ("TPP_HIP_CHAIN", "TPP_HIP_ASYNC", "TPP_HIP_VARIANT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_starved_worker.py"), scenario], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, "worker exited %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_first_chain_launch_on_a_shared_device_runs_call_by_call():
    d, err = run("A")
    assert d["hog_workgroups_started"] >= 150, d
    assert d["one_launch"] is False and d["identical"] is True and d["later_one_launch"] is False, d
    assert "starved" in err and "call by call" in err, err[-1500:]


def test_chain_launches_starved_later_are_rerun_at_the_synchronisation_point():
    d, err = run("B")
    assert d["free_one_launch"] == [True, True] and d["free_identical"] is True, d
    assert d["hog_workgroups_started"] >= 150, d
    assert d["hogged_one_launch"] == [True, True, True], d  # launched asynchronously, found out at the synchronisation
    assert d["identical"] is True and d["later_one_launch"] is False, d
    assert "starved" in err and "re-run call by call" in err, err[-1500:]
    # round 6: the sticky status counts the journaled launches that were repaired (the starved one and those behind it);
    # the healthy launches before the hog were checked at their own synchronisation and left alone
    assert d["free_chain_status"] == 0 and 1 <= d["chain_status"] <= 3, d
