"""RCCL bound directly (ctypes): the collectives of the sharded step as plain stream-ordered launches.

Why not torch.distributed's "nccl" backend for the data path: c10d wraps every collective in work objects, runs them on its own
stream and keeps a watchdog thread that polls HIP events -- a poll while a HIP-graph capture is open aborts the capture (round 3:
the sharded step captured in round 2 and aborted in round 3 on the same settings, a matter of timing).  Here a collective is ONE
call of ``ncclAllReduce / ncclReduceScatter / ncclAllGather`` on the HIP stream the caller is on: nothing else touches the stream,
so a step that contains them captures and replays like any other sequence of kernels, and the transfers are ordered with the
kernels around them by the stream itself (no events, no host synchronisation).

The library is the ``librccl.so`` PyTorch ships and has already loaded (one RCCL per process).  The communicator is bootstrapped
over the existing ``torch.distributed`` group of any backend: rank 0 draws the ``ncclUniqueId`` and broadcasts its 128 bytes.
``Comm`` duck-types what dist.py needs from a process group (dist.all_reduce / reduce_scatter_tensor / all_gather_into_tensor and
functional._all_reduce_sum dispatch on ``is_direct_rccl``).

Reference counterpart: none -- the reference is single-device (collectives NONE); the exchange steps are
SURVEY section 8(e)'s.
"""
import ctypes
import os

import torch

_NCCL_UNIQUE_ID_BYTES = 128
_DTYPE = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7, torch.float64: 8,
          torch.bfloat16: 9}
_OP = {"sum": 0, "max": 2, "min": 3}


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * _NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_lib = None


def load():
    """The RCCL PyTorch itself uses (torch/lib/librccl.so): dlopen of a loaded library returns the loaded instance."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        raise RcclError(f"librccl.so not found next to torch ({path})")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = ci, [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype, lib.ncclCommInitRank.argtypes = ci, [ctypes.POINTER(vp), ci, _UniqueId, ci]
    lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = ci, [vp]
    lib.ncclCommAbort.restype, lib.ncclCommAbort.argtypes = ci, [vp]
    lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [ci]
    lib.ncclAllReduce.restype, lib.ncclAllReduce.argtypes = ci, [vp, vp, sz, ci, ci, vp, vp]
    lib.ncclReduceScatter.restype, lib.ncclReduceScatter.argtypes = ci, [vp, vp, sz, ci, ci, vp, vp]
    lib.ncclAllGather.restype, lib.ncclAllGather.argtypes = ci, [vp, vp, sz, ci, vp, vp]
    _lib = lib
    return lib


def _check(code, what):
    if code != 0:
        raise RcclError(f"{what}: {load().ncclGetErrorString(code).decode()} ({code})")


class Comm:
    """One RCCL communicator over the ranks of `bootstrap` (a torch.distributed group; None = the default group; with
    ``world == 1`` no group is needed at all).  Every method launches on ``torch.cuda.current_stream()`` and returns at once."""

    is_direct_rccl = True

    def __init__(self, rank, world, device, bootstrap=None, unique_id=None):
        """`unique_id`: the 128 bytes rank 0 drew (bring_up() below distributes them in a step every rank takes part in whatever
        happened before); None = draw / broadcast here (one-rank communicators, tests)."""
        lib = load()
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self._comm = ctypes.c_void_p()
        self.launches = 0
        uid = _UniqueId()
        if unique_id is not None:
            ctypes.memmove(ctypes.byref(uid), unique_id, _NCCL_UNIQUE_ID_BYTES)
        else:
            if self.rank == 0:
                _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
            if self.world > 1:
                import torch.distributed as dist
                box = [bytes(uid.internal) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=dist.get_global_rank(bootstrap, 0) if bootstrap is not None else 0, group=bootstrap)
                ctypes.memmove(ctypes.byref(uid), box[0], _NCCL_UNIQUE_ID_BYTES)
        # RCCL prints a version banner on the process's stdout when it initialises: keep stdout clean (bench.py prints ONE JSON line there)
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            with torch.cuda.device(self.device):
                _check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        # an exception between ncclCommInitRank and destroy() (a failed step, a failed capture) must not leak the communicator, and
        # must not wait on a stream that a hung collective may never drain: abort instead of synchronise + destroy
        if exc_type is None:
            self.destroy()
        else:
            self.abort()
        return False

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass

    def size(self):
        return self.world

    @staticmethod
    def _index(device):
        """cuda and cuda:<current> are the same device."""
        return device.index if device.index is not None else torch.cuda.current_device()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _prep(self, t):
        if not t.is_cuda or not t.is_contiguous():
            raise RcclError("direct RCCL collectives take contiguous device tensors")
        if self._index(t.device) != self._index(self.device):
            raise RcclError(f"tensor on {t.device}, communicator on {self.device}")
        if t.dtype not in _DTYPE:
            raise RcclError(f"dtype {t.dtype} has no RCCL counterpart")
        return ctypes.c_void_p(t.data_ptr()), _DTYPE[t.dtype]

    def all_reduce(self, t, op="sum"):
        """In place."""
        p, dt = self._prep(t)
        _check(load().ncclAllReduce(p, p, t.numel(), dt, _OP[op], self._comm, self._stream()), "ncclAllReduce")
        self.launches += 1

    def reduce_scatter_tensor(self, out, inp, op="sum"):
        """out [n] = this rank's block of the reduction over ranks of inp [world * n]."""
        if inp.numel() != out.numel() * self.world or inp.dtype != out.dtype:
            raise RcclError(f"reduce-scatter: {inp.numel()} elements in, {out.numel()} out, world {self.world}")
        pi, dt = self._prep(inp)
        po, _ = self._prep(out)
        _check(load().ncclReduceScatter(pi, po, out.numel(), dt, _OP[op], self._comm, self._stream()), "ncclReduceScatter")
        self.launches += 1

    def all_gather_into_tensor(self, full, mine):
        if full.numel() != mine.numel() * self.world or full.dtype != mine.dtype:
            raise RcclError(f"all-gather: {mine.numel()} elements in, {full.numel()} out, world {self.world}")
        pm, dt = self._prep(mine)
        pf, _ = self._prep(full)
        _check(load().ncclAllGather(pm, pf, mine.numel(), dt, self._comm, self._stream()), "ncclAllGather")
        self.launches += 1

    def destroy(self):
        """Orderly end: waits for the device, then frees the communicator.  After an error use abort()."""
        if self._comm:
            torch.cuda.synchronize(self.device)
            load().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    def abort(self):
        """Error path: frees the communicator WITHOUT synchronising (a hung collective never drains its stream)."""
        if self._comm:
            comm, self._comm = self._comm, ctypes.c_void_p()
            load().ncclCommAbort(comm)


def _run_bounded(fn, timeout_s):
    """fn() on a helper thread: (finished, result-or-exception).  A call that has not returned after timeout_s stays stuck on its
    thread (a daemon: it cannot keep the process alive) and is reported as not finished."""
    import threading
    box = {}

    def work():
        try:
            box["value"] = fn()
        except BaseException as e:      # noqa: BLE001  (reported to the caller, never swallowed)
            box["error"] = e

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        return False, None
    return True, box.get("error", box.get("value"))


def bring_up(rank, world, device, control=None, timeout_s=120.0, log=None):
    """A direct communicator over `world` ranks, brought up in STAGES that every rank leaves together (advisor r4: the one-try-block
    bring-up of round 4 could leave peers in different collectives when one rank failed early).  `control`: the torch.distributed
    group of the control plane (gloo); None = the default group.  Returns the Comm on every rank, or None on every rank (then the
    caller falls back to torch.distributed's nccl backend).  A rank whose ncclCommInitRank or probe all-reduce has not RETURNED
    after `timeout_s` cannot be recovered in-process (the call is stuck inside RCCL): after the ranks have agreed on it, every rank
    logs and exits with status 3 -- the launcher then starts fresh processes; nothing is re-executed from a process that has
    touched the GPU.

      stage 1  load librccl.so                      -> agree (MIN over ranks)
      stage 2  rank 0 draws the unique id           -> broadcast (every rank takes part; a failure travels as None)
      stage 3  ncclCommInitRank, bounded            -> agree
      stage 4  one probe all-reduce, bounded        -> agree
    """
    import torch.distributed as dist
    say = log if log is not None else (lambda m: None)

    def agree(ok):
        flag = torch.tensor([int(bool(ok))], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=control)
        return bool(int(flag.item()))

    try:
        load()
        ok = True
    except Exception as e:          # noqa: BLE001
        say(f"rank {rank}: librccl.so did not load ({type(e).__name__}: {str(e)[:120]})")
        ok = False
    if not agree(ok):
        return None
    box = [None]
    if rank == 0:
        try:
            uid = _UniqueId()
            _check(load().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
            box = [bytes(uid.internal)]
        except Exception as e:      # noqa: BLE001
            say(f"rank 0: ncclGetUniqueId failed ({type(e).__name__}: {str(e)[:120]})")
    dist.broadcast_object_list(box, src=(dist.get_global_rank(control, 0) if control is not None else 0), group=control)
    if box[0] is None:
        return None
    done, comm = _run_bounded(lambda: Comm(rank, world, device, unique_id=box[0]), timeout_s)
    stuck = not done
    failed = stuck or isinstance(comm, BaseException)
    if failed and not stuck:
        say(f"rank {rank}: ncclCommInitRank failed ({type(comm).__name__}: {str(comm)[:120]})")
    all_ok = agree(not failed)
    any_stuck = not agree(not stuck)
    if any_stuck:
        say(f"rank {rank}: a rank's ncclCommInitRank did not return within {timeout_s:.0f} s; exiting (status 3) for a fresh launch")
        os._exit(3)
    if not all_ok:
        if isinstance(comm, Comm):
            comm.abort()
        return None

    def probe():
        t = torch.ones(8, device=device)
        comm.all_reduce(t, "sum")
        torch.cuda.synchronize(device)
        return bool((t == world).all())

    done, good = _run_bounded(probe, timeout_s)
    stuck = not done
    failed = stuck or good is not True
    all_ok = agree(not failed)
    any_stuck = not agree(not stuck)
    if any_stuck:
        say(f"rank {rank}: the probe all-reduce did not complete within {timeout_s:.0f} s on some rank; exiting (status 3) for a fresh launch")
        os._exit(3)
    if not all_ok:
        say(f"rank {rank}: the probe all-reduce gave a wrong sum or raised on some rank")
        comm.abort()
        return None
    return comm


class VirtualWorld:
    """TIMING-ONLY rehearsal of rank `rank` of `world` on ONE GPU: the shard, the chunk arithmetic and every
    collective LAUNCH of the real run, on a communicator of one rank -- a reduce-scatter moves this rank's own block, an all-gather
    fills this rank's block of a zeroed output.  The numbers are those of a run in which the other ranks contribute zeros: finite,
    wrong, and never compared with anything."""

    is_direct_rccl = True

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self._one = Comm(0, 1, device)

    launches = property(lambda self: self._one.launches)

    def size(self):
        return self.world

    def all_reduce(self, t, op="sum"):
        self._one.all_reduce(t, op)

    def reduce_scatter_tensor(self, out, inp, op="sum"):
        n = out.numel()
        self._one.reduce_scatter_tensor(out, inp.view(-1)[self.rank * n:(self.rank + 1) * n], op)

    def all_gather_into_tensor(self, full, mine):
        n = mine.numel()
        full.zero_()
        self._one.all_gather_into_tensor(full.view(-1)[self.rank * n:(self.rank + 1) * n], mine)

    def destroy(self):
        self._one.destroy()
