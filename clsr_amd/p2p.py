"""Small-message all-reduce communicator over peer-mapped exchange buffers (csrc/p2p.hip).

The synchronised batch-norm statistics of the data-parallel step (reference batch-norm:
models/base_model.py:673-679) are 16 all-reduces of <= 2 KB per step, each needed by the very next launch of a
dependent chain.  ``SmallComm`` sums them with ONE single-workgroup kernel on the issuing stream instead of a
``torch.distributed`` call (host call + two stream hops into RCCL's stream, ~20 us each).  The communicator is created
once: every rank allocates a fine-grained exchange buffer, the hipIpc handles travel through the process group
(``all_gather_object``), every rank maps its peers' buffers.  Works between the GPUs of one node (xGMI peer access) and
between processes that share one GPU (the test box)."""
import ctypes

import torch

from clsr_amd import _lib, ops


class _PeerMapped(object):
    """An exchange buffer per rank, every rank's buffer mapped into every other rank (hipIpc handles through
    ``gather_objects``), and a communicator object of the C side built from the mapped addresses."""

    _alloc, _create, _destroy, _max_world = None, None, None, None       # names of the C entry points (subclasses)

    def __init__(self, rank, world, gather_objects):
        """``gather_objects(obj) -> list of every rank's obj`` (torch.distributed.all_gather_object or equivalent).

        Every rank runs the SAME sequence of gathers whatever fails locally (a rank that skipped one would pair its next
        collective with a different one of its peers): stage 1 gathers (handle or None), stage 2 gathers (mapped-all or
        the error text); the verdict of a stage is computed from the gathered list, so all ranks raise at the same stage --
        after releasing whatever they had allocated or opened."""
        lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self._own, self._peers, self.handle = None, [], None
        nb = lib.clsr_comm_ipc_handle_bytes()
        mine, err = None, None
        try:
            if self.world > getattr(lib, self._max_world)():
                raise ValueError("%s: at most %d ranks (one node)" % (type(self).__name__, getattr(lib, self._max_world)()))
            buf = ctypes.c_void_p()
            _lib.check(getattr(lib, self._alloc)(ctypes.byref(buf)), self._alloc)
            self._own = buf
            h = ctypes.create_string_buffer(nb)
            _lib.check(lib.clsr_comm_ipc_handle(buf, h), "clsr_comm_ipc_handle")
            mine = bytes(h.raw)
        except Exception as e:      # noqa: BLE001 -- reported to every rank below
            err = "rank %d: %s" % (self.rank, str(e)[:120])
        handles = gather_objects(mine)                                   # ---- stage 1: every rank, always
        if any(h is None for h in handles):
            self.close()
            raise RuntimeError("%s: exchange buffer / IPC handle failed on ranks %s%s" % (
                type(self).__name__, [r for r, h in enumerate(handles) if h is None], (" (%s)" % err) if err else ""))
        try:
            ptrs = (ctypes.c_void_p * self.world)()
            for r in range(self.world):
                if r == self.rank:
                    ptrs[r] = self._own
                    continue
                p = ctypes.c_void_p()
                hb = ctypes.create_string_buffer(handles[r], nb)
                _lib.check(lib.clsr_comm_ipc_open(hb, ctypes.byref(p)), "clsr_comm_ipc_open (rank %d)" % r)
                self._peers.append(p)
                ptrs[r] = p
            comm = ctypes.c_void_p()
            _lib.check(getattr(lib, self._create)(self.rank, self.world, ptrs, ctypes.byref(comm)), self._create)
            self.handle = comm.value      # (an integer address: what ops.call passes for a void*)
        except Exception as e:      # noqa: BLE001
            err = "rank %d: %s" % (self.rank, str(e)[:120])
        # ---- stage 2: nobody pushes before every rank has mapped every buffer -- and everybody learns if one could not
        mapped = gather_objects(err or "ready")
        bad = [m for m in mapped if m != "ready"]
        if bad:
            self.close()
            raise RuntimeError("%s: mapping the peers' exchange buffers failed: %s" % (type(self).__name__, "; ".join(bad)))

    def close(self):
        lib = _lib.load()
        if getattr(self, "handle", None):
            getattr(lib, self._destroy)(ctypes.c_void_p(self.handle))
            self.handle = None
        for p in getattr(self, "_peers", []):
            lib.clsr_comm_ipc_close(p)
        self._peers = []
        if getattr(self, "_own", None):
            lib.clsr_comm_free(self._own)
            self._own = None


class SmallComm(_PeerMapped):
    """Communicator of clsr_allreduce_small (csrc/p2p.hip)."""

    _alloc, _create, _destroy, _max_world = "clsr_comm_alloc", "clsr_comm_create", "clsr_comm_destroy", "clsr_comm_max_world"

    def __init__(self, rank, world, gather_objects):
        super().__init__(rank, world, gather_objects)
        self.max_doubles = _lib.load().clsr_comm_max_doubles()

    def all_reduce(self, t, n=None):
        """in-place sum of the first ``n`` doubles of ``t`` over the ranks, asynchronous on the current stream"""
        n = t.numel() if n is None else int(n)
        assert t.dtype == torch.float64 and n <= self.max_doubles
        ops.call("clsr_allreduce_small", self.handle, t, n)

    def set_abort(self, flag):
        """``flag``: a device double (view) that an all-reduce which gives up waiting raises; None detaches it"""
        _lib.check(_lib.load().clsr_comm_set_abort(ctypes.c_void_p(self.handle), None if flag is None else ctypes.c_void_p(flag.data_ptr())),
                   "clsr_comm_set_abort")

    def reset_channels(self):
        """forget the stream -> channel assignment (every rank at the same point of its call sequence)"""
        _lib.check(_lib.load().clsr_comm_reset_channels(ctypes.c_void_p(self.handle)), "clsr_comm_reset_channels")

    def error(self):
        """sequence number of the last all-reduce that gave up waiting for a peer (0: none); synchronises the device"""
        return int(_lib.load().clsr_comm_error(ctypes.c_void_p(self.handle)))


def from_process_group(rank, world, group=None):
    """SmallComm whose handles travel through torch.distributed (any backend with all_gather_object)."""
    import torch.distributed as dist

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj, group=group)
        return out

    return SmallComm(rank, world, gather)


class HeadsComm(_PeerMapped):
    """Communicator of the fused heads launches (csrc/headsfused.hip) for synchronised batch-norm statistics across ranks:
    every rank's exchange buffer mapped into every other rank; the workgroups of the launches push their group sums into
    all of them."""

    _alloc, _create, _destroy, _max_world = ("clsr_heads_comm_alloc", "clsr_heads_comm_create", "clsr_heads_comm_destroy",
                                             "clsr_heads_comm_max_world")

    def self_test(self, device, nblocks=48, timeout_s=5.0):
        """Every rank at the same point: the arrive / wait protocol of the heads launches through all stages with known
        contributions; True iff this rank saw the expected sums."""
        nbytes = int(ops.query("clsr_heads_fused_workspace_bytes"))
        ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)
        ok = torch.ones(1, dtype=torch.int32, device=device)
        ops.call("clsr_heads_comm_self_test", self.handle, int(nblocks), ws, ws.numel() * 4, ok, float(timeout_s))
        torch.cuda.synchronize(device)
        return bool(int(ok.item()) == 1)


def heads_comm_from_process_group(rank, world, group=None):
    import torch.distributed as dist

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj, group=group)
        return out

    return HeadsComm(rank, world, gather)
