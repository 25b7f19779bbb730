"""Peer-store all-gather of the row-sharded MLP's output: one process per GPU; every rank stores its row
block straight into every peer's output buffer through IPC-mapped device pointers (xGMI on a node), completion by flags -
csrc/peer_gather.hip. torch.distributed is used ONCE, to exchange the 64-byte IPC handles (any backend: the handles are host
objects); the steady state is two small kernels per step on the rank's own stream, no collective library call.

The reference has no distributed path; this replaces the dist.all_gather_into_tensor of mlp.all_gather_rows
where the buffers can be mapped (same node). RCCL remains the fallback: PeerGather.create returns None when a mapping fails.
"""
import ctypes

import numpy as np

VP = ctypes.c_void_p


class _RawDevice:
    """a raw device allocation exposed to torch without a copy (__cuda_array_interface__)"""

    def __init__(self, ptr, nbytes, typestr, itemsize):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PeerGather:
    """full[parity] buffers of `total_bytes` on every rank + flag / ready words; gather(src) runs one step and returns the
    tensor holding the gathered output of THIS step (double-buffered: valid until the step after next)."""

    def __init__(self, rt, rank, world, total_bytes, group=None):
        import torch
        import torch.distributed as dist
        self.rt, self.rank, self.world, self.total = rt, rank, world, int(total_bytes)
        lib = rt.lib
        ctl_bytes = 4096  # flags[world] at 0, ready[world] at 1024, ticket at 2048, err at 2112
        self._own = [lib.xsmm_hip_peer_alloc(self.total), lib.xsmm_hip_peer_alloc(self.total), lib.xsmm_hip_peer_alloc(ctl_bytes)]
        handles = []
        for p in self._own:
            h = (ctypes.c_ubyte * 64)()
            if lib.xsmm_hip_ipc_export(p, h) != 0:
                raise RuntimeError("hipIpcGetMemHandle failed")
            handles.append(bytes(h))
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, handles, group=group)
        else:
            everyone[0] = handles
        self._mapped = []
        self.full_ptr = [[None] * world, [None] * world]  # [parity][peer]
        self.ctl_ptr = [None] * world
        for w in range(world):
            if w == rank:
                ptrs = self._own
            else:
                ptrs = []
                for hb in everyone[w]:
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(hb)
                    q = lib.xsmm_hip_ipc_open(buf)
                    if not q:
                        raise RuntimeError("hipIpcOpenMemHandle failed for peer %d" % w)
                    ptrs.append(q)
                    self._mapped.append(q)
            self.full_ptr[0][w], self.full_ptr[1][w], self.ctl_ptr[w] = ptrs
        arr = lambda v: (VP * world)(*[VP(int(x)) for x in v])  # noqa: E731
        self._dst = [arr(self.full_ptr[0]), arr(self.full_ptr[1])]
        self._flags = arr(self.ctl_ptr)
        self._ready = arr([int(c) + 1024 for c in self.ctl_ptr])
        me = int(self.ctl_ptr[rank])
        self._my_flags, self._my_ready, self._ticket, self._err = VP(me), VP(me + 1024), VP(me + 2048), VP(me + 2112)
        self.epoch = 0
        self.full = [torch.as_tensor(_RawDevice(self._own[i], self.total, "<i2", 2), device="cuda") for i in (0, 1)]
        self._err_view = torch.as_tensor(_RawDevice(me + 2112, 4, "<i4", 4), device="cuda")
        if world > 1:
            dist.barrier(group=group)  # every rank has mapped every buffer before anybody writes

    @classmethod
    def create(cls, rt, rank, world, total_bytes, group=None):
        """the gather object, or None when the buffers cannot be shared (then use RCCL: mlp.all_gather_rows)"""
        import sys
        try:
            obj = cls(rt, rank, world, total_bytes, group)
        except Exception as ex:  # a mapping problem must not cost the run: RCCL is the fallback
            sys.stderr.write("[tpp-mlir_amd.peer] peer-store gather unavailable (%s): falling back to RCCL\n" % ex)
            obj = None
        # every rank or none - and only if three known patterns come out right on EVERY rank (both buffer parities and a re-use):
        # a mapping that "works" but is not coherent between these devices must cost a fallback, not a wrong output or a hung step
        ok = obj is not None
        if world > 1:
            import torch.distributed as dist
            votes = [None] * world
            dist.all_gather_object(votes, ok, group=group)
            if not all(votes):
                if obj is not None:
                    obj.close()
                return None
            ok = obj.selftest()
            dist.all_gather_object(votes, ok, group=group)
            if not all(votes):
                if rank == 0:
                    sys.stderr.write("[tpp-mlir_amd.peer] peer-store gather failed its self-test on ranks %s: falling back to RCCL\n"
                                     % [w for w, v in enumerate(votes) if not v])
                obj.close()
                return None
        return obj

    def selftest(self):
        """three gathers of a pattern every rank can compute for every other rank; True if this rank's copies are exact and no
        wait timed out"""
        import torch
        per = (self.total // self.world) // 16 * 16
        n16 = per // 2
        base = torch.arange(n16, device="cuda", dtype=torch.int32)

        def pattern(w, step):
            return ((base * 31 + 7919 * (w + 1) + 104729 * step) % 32749).to(torch.int16)
        ok = True
        try:
            for step in range(3):
                src = pattern(self.rank, step)
                torch.cuda.synchronize()  # (the runtime's stream need not be torch's)
                out = self.gather(src, per, self.rank * per)
                self.rt.synchronize()
                torch.cuda.synchronize()
                if int(self._err_view.cpu()[0]) != 0:
                    self._err_view.zero_()
                    ok = False
                    break
                for w in range(self.world):
                    if not torch.equal(out[w * n16:(w + 1) * n16], pattern(w, step)):
                        ok = False
        except Exception:
            ok = False
        return ok

    def gather(self, src, nbytes, dst_offset):
        """enqueue one step on the runtime's stream: this rank's `nbytes` at `src` go to offset `dst_offset` of every rank's
        output buffer; returns the (int16 view of the) buffer that holds the gathered output once the stream reaches this point"""
        self.epoch += 1
        par = self.epoch & 1
        from .runtime import _addr
        self.rt.lib.xsmm_hip_peer_gather(_addr(src), int(nbytes), int(dst_offset), self.world, self.rank, self._dst[par], self._flags,
                                         self._ready, self._my_flags, self._my_ready, self._ticket, self._err, self.epoch)
        return self.full[par]

    def overlap(self, enable=True):
        """the wait kernels on a side stream (csrc/peer_gather.hip): the next step's compute starts behind the scatter instead of
        behind the arrival of every peer's block. Process-wide switch; call drain() (or synchronize the device) before reading a
        gathered output."""
        return bool(self.rt.lib.xsmm_hip_peer_overlap(1 if enable else 0))

    def drain(self):
        self.rt.lib.xsmm_hip_peer_drain()

    def check(self):
        """after a synchronize: did a wait time out (a peer died / never arrived)?"""
        self.drain()
        e = int(self._err_view.cpu()[0])
        if e:
            raise RuntimeError("peer-store gather timed out (code 0x%x): a peer never arrived" % e)

    def close(self):
        for q in self._mapped:
            self.rt.lib.xsmm_hip_ipc_close(q)
        self._mapped = []
