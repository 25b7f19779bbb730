#!/bin/bash
# host_path_rig.sh - the per-invoke HOST path of the runtime without a GPU: tools/tpp_replay.cpp + csrc/runtime.cpp + host_cache.cpp
# against tests/tsan/fake_hip.cpp (the HIP host API and the kernel launchers over host memory), g++ -O3, launches switched to
# "return at once" (FAKE_HIP_NO_COMPUTE=1) and result checks off (TPP_REPLAY_NO_CHECK=1): the "host side of the invokes" figure is then
# the enqueue path alone. For single-thread elimination runs (cut a piece, look at the difference); NOT for anything that hands work
# between threads - a virtualised box can have microseconds of cross-core latency (measured in the build container: 3.9 us round trip,
# which made the launch thread look 2x slower there while it is 25-30 % faster on the MI355X host).
#   tools/host_path_rig.sh [out-binary]            then e.g.
#   FAKE_HIP_NO_COMPUTE=1 TPP_REPLAY_NO_CHECK=1 TPP_HIP_LAUNCH_THREAD=0 ./replay_host --script mha_qk --queue 1 -n 4000
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/replay_host}
g++ -std=c++17 -O3 -fPIC -fopenmp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" -I"$ROOT/tpp-mlir_amd/csrc" \
  "$ROOT/tpp-mlir_amd/csrc/runtime.cpp" "$ROOT/tpp-mlir_amd/csrc/host_cache.cpp" "$ROOT/tests/tsan/fake_hip.cpp" "$ROOT/tools/tpp_replay.cpp" \
  -o "$OUT" -pthread -ldl
echo "built $OUT"
