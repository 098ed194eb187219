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
