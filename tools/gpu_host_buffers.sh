#!/bin/bash
# host buffers through the reference's 15 symbols only (tools/tpp_replay --host-buffers: plain malloc, 64-byte aligned, modes from the
# environment) against device pointers: BASELINE config 2's shape as a whole-layer call, and the reference's headline MLP as emitted
R=tools/tpp_replay; F="mean\|host buffers\|repeats"
echo "# C2-shaped whole-layer call (1024x1024x1024 f32, k=64 br=16, BETA_0)"
echo "## device pointers (hipMalloc + async): the headline configuration"; $R --batch 1024 --layers 1024,1024 --whole-layer -n 1000 --repeats 5 2>&1 | grep "$F"
echo "## host buffers, no environment: the plain per-invoke mirror (synchronous)"; $R --host-buffers --batch 1024 --layers 1024,1024 --whole-layer -n 200 2>&1 | grep "$F"
echo "## host buffers, TPP_HIP_HOST_CACHE=1 (synchronous: results on the host at every return)"; TPP_HIP_HOST_CACHE=1 $R --host-buffers --batch 1024 --layers 1024,1024 --whole-layer -n 200 2>&1 | grep "$F"
echo "## host buffers, TPP_HIP_ASYNC=1 TPP_HIP_HOST_CACHE=1"; TPP_HIP_ASYNC=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers --batch 1024 --layers 1024,1024 --whole-layer -n 1000 --repeats 5 2>&1 | grep "$F"
echo "## host buffers, TPP_HIP_ASYNC=1 only (no cache): every invoke mirrors + synchronises"; TPP_HIP_ASYNC=1 $R --host-buffers --batch 1024 --layers 1024,1024 --whole-layer -n 200 2>&1 | grep "$F"
echo
echo "# the reference's headline MLP as emitted: 3 x 256 invokes of 32x32x32 tiles, bs 256, bias + relu"
M="--batch 256 --layers 1024,1024,1024,1024 --tiles 32 --bias --relu"
echo "## device pointers, tile queue"; $R $M --queue 1 -n 200 --repeats 5 2>&1 | grep "$F"
echo "## device pointers, 8 OpenMP callers"; $R $M --queue 1 -n 200 --repeats 5 --threads 8 2>&1 | grep "$F"
echo "## host buffers, no environment (768 synchronous mirrored invokes per iteration)"; $R --host-buffers $M -n 5 2>&1 | grep "$F"
echo "## host buffers, TPP_HIP_HOST_CACHE=1 (synchronous)"; TPP_HIP_HOST_CACHE=1 $R --host-buffers $M -n 5 2>&1 | grep "$F"
echo "## host buffers, TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1"; TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers $M -n 200 --repeats 5 2>&1 | grep "$F"
echo "## the same, 8 OpenMP callers"; TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers $M -n 200 --repeats 5 --threads 8 2>&1 | grep "$F"
echo "## the same, N = 1000 calls per loop (the one-off stall of the first write-back amortised)"; TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers $M -n 1000 --repeats 3 2>&1 | grep "$F"
echo "## bf16 + VNNI-2, host buffers, async + queue + cache"; TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers --bf16 $M -n 200 --repeats 5 2>&1 | grep "$F"
echo "## whole-layer calls on host buffers, async + cache"; TPP_HIP_ASYNC=1 TPP_HIP_HOST_CACHE=1 $R --host-buffers --batch 256 --layers 1024,1024,1024,1024 --whole-layer --bias --relu -n 500 --repeats 5 2>&1 | grep "$F"
echo "## whole-layer calls, device pointers"; $R --batch 256 --layers 1024,1024,1024,1024 --whole-layer --bias --relu -n 500 --repeats 5 2>&1 | grep "$F"
