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
        """Local part only (allocate + export the IPC handles). No collective runs here and nothing raises: a failure is recorded
        in self.ok and voted on in create(), so that every rank issues the same collectives in the same order whatever happens on
        one of them (a rank that raised before the barrier left the others hanging in it)."""
        self.rt, self.rank, self.world, self.total = rt, rank, world, int(total_bytes)
        self.group = group
        self._own, self._mapped, self.handles = [], [], None
        self.ok, self.why = True, ""
        self.epoch = 0
        lib = rt.lib
        ctl_bytes = 4096  # flags[world] at 0, ready[world] at 1024, ticket at 2048, err at 2112
        try:
            for nbytes in (self.total, self.total, ctl_bytes):
                q = lib.xsmm_hip_peer_alloc(nbytes)
                if not q:
                    raise RuntimeError("device allocation of %d bytes failed" % nbytes)
                self._own.append(q)
            handles = []
            for q in self._own:
                h = (ctypes.c_ubyte * 64)()
                if lib.xsmm_hip_ipc_export(q, h) != 0:
                    raise RuntimeError("hipIpcGetMemHandle failed")
                handles.append(bytes(h))
            self.handles = handles
        except Exception as ex:
            self.ok, self.why = False, str(ex)

    def _map(self, everyone):
        """open every peer's handles (local; failures are recorded, not raised) and build the pointer tables"""
        import torch
        lib, world, rank = self.rt.lib, self.world, self.rank
        try:
            if not self.ok:
                raise RuntimeError(self.why)
            if any(h is None for h in everyone):
                raise RuntimeError("rank(s) %s could not export their buffers" % [w for w, h in enumerate(everyone) if h is None])
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
            self.full = [torch.as_tensor(_RawDevice(self._own[i], self.total, "<i2", 2), device="cuda") for i in (0, 1)]
            self._err_view = torch.as_tensor(_RawDevice(me + 2112, 4, "<i4", 4), device="cuda")
        except Exception as ex:
            self.ok, self.why = False, str(ex)

    @classmethod
    def create(cls, rt, rank, world, total_bytes, group=None):
        """The gather object, or None - on EVERY rank or on none - when the buffers cannot be shared or the self-test fails
        somewhere (then use RCCL: mlp.all_gather_rows). Collectives, identical on every rank regardless of local failures:
        all_gather_object(handles) -> [local mapping] -> all_gather_object(ok) -> [self-test: three gathers] ->
        all_gather_object(ok). The second vote also orders "every rank has mapped every buffer" before anybody's first store."""
        import sys
        obj = cls(rt, rank, world, total_bytes, group)
        if world > 1:
            import torch.distributed as dist
            everyone = [None] * world
            dist.all_gather_object(everyone, obj.handles, group=group)
        else:
            everyone = [obj.handles]
        obj._map(everyone)
        votes = [obj.ok]
        if world > 1:
            votes = [None] * world
            dist.all_gather_object(votes, obj.ok, group=group)
        if not all(votes):
            sys.stderr.write("[tpp-mlir_amd.peer] rank %d: peer-store gather unavailable (%s; ranks that failed: %s): falling back to RCCL\n"
                             % (rank, obj.why or "a peer failed", [w for w, v in enumerate(votes) if not v]))
            obj.close(collective=False)  # (nobody has stored anything: no barrier needed, and every rank is on this path)
            return None
        # only if three known patterns come out right on EVERY rank (both buffer parities and a re-use): a mapping that "works" but
        # is not coherent between these devices must cost a fallback, not a wrong output or a hung step. (world == 1 too: the
        # kernels and the flag protocol run the same way against the rank's own buffers.)
        ok = obj.selftest()
        votes = [ok]
        if world > 1:
            votes = [None] * world
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

    def close(self, collective=True):
        """unmap the peers' buffers and free this rank's own (2 x total_bytes + the control block). Collective by default: the
        owners free only after every rank has drained its stream and unmapped (a peer may still be storing into this rank's
        buffers, or polling its flags). collective=False: only where no gather has run and every rank takes the same path."""
        lib = self.rt.lib
        if collective and self.world > 1:
            import torch.distributed as dist
            try:
                self.rt.synchronize()
                self.drain()
            except Exception:
                pass
            dist.barrier(group=self.group)  # nobody stores into anybody's buffers any more
        for q in self._mapped:
            lib.xsmm_hip_ipc_close(q)
        self._mapped = []
        if collective and self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.group)  # every peer has unmapped: the owners may free
        self.full, self._err_view = [], None
        for q in self._own:
            lib.xsmm_hip_peer_free(q)
        self._own = []
