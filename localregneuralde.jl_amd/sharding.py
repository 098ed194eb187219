"""Batch-dimension sharding: rank r owns columns [r*B/R, (r+1)*B/R) of the
global batch; weights replicated; the only exchange is the fp64 partial-sum all-reduce that
liblrnde issues per attempted step (RCCL over xGMI)."""
import ctypes as C

import torch

from . import _lib as L


def shard_columns(x, rank, nranks):
    """x: (B_global, D) tensor or array -> this rank's contiguous block of samples."""
    B = x.shape[0]
    if B % nranks:
        raise ValueError(f"global batch {B} must divide evenly over {nranks} ranks")
    per = B // nranks
    return x[rank * per:(rank + 1) * per]


def init_comm(handle, rank, nranks, group=None):
    """Creates the library's RCCL communicator: rank 0 makes the unique id, torch.distributed
    (any backend) broadcasts its 128 bytes."""
    import torch.distributed as dist
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        rc = L.lib.lrnde_comm_unique_id(buf)
        if rc != 0:
            raise L.LrndeError(rc, "lrnde_comm_unique_id failed")
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
    if nranks > 1:
        backend = dist.get_backend(group)
        if backend == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=group)
        t = t.cpu()
    raw = bytes(t.tolist())
    buf2 = (C.c_uint8 * 128).from_buffer_copy(raw)
    handle._chk(L.lib.lrnde_comm_init(handle._ctx, buf2, int(rank), int(nranks)))


class LocalComm:
    """In-process communicator (include/lrnde_hooks.h): several handles of ONE process, one host thread each, run the
    library's nranks > 1 code — on one GPU (single-device by construction) — where RCCL cannot (it refuses two ranks on one device).
    `join(handle, rank)` replaces `init_comm`; the sharded calls must then be issued by all ranks concurrently
    (`run_ranks`)."""

    def __init__(self, nranks):
        self.nranks = int(nranks)
        self._lc = C.c_void_p()
        rc = L.lib.lrnde_local_comm_create(C.byref(self._lc), self.nranks)
        if rc != 0:
            raise L.LrndeError(rc, "lrnde_local_comm_create failed")
        self._handles = []

    def join(self, handle, rank):
        handle._chk(L.lib.lrnde_comm_init_local(handle._ctx, self._lc, int(rank)))
        self._handles.append(handle)

    def close(self):
        if getattr(self, "_lc", None):
            for h in self._handles:
                if getattr(h, "_ctx", None):
                    L.lib.lrnde_comm_destroy(h._ctx)
            L.lib.lrnde_local_comm_destroy(self._lc)
            self._lc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_ranks(fns):
    """Runs one callable per rank, each on its own host thread (ctypes releases the GIL inside the library), and
    returns their results in rank order; the first exception of any rank is re-raised."""
    import threading
    out, err = [None] * len(fns), [None] * len(fns)

    def work(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # noqa: BLE001
            err[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
