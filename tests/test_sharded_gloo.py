"""world_size-2 gloo test of the batch-sharded path (BASELINE config 3): two
ranks each own half of the columns, exchange only the fp64 partial sums, and must take exactly
the same steps — and produce exactly the same states — as the unsharded oracle solve."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, D, H, B, tol, scale, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import oracle as O
    import lrnde_amd as P
    from sharded_emulation import sharded_solve
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    p = O.glorot_mlp_params(D, H, seed=0) * np.float32(scale)
    x = np.random.default_rng(0).random((B, D), dtype=np.float32)
    fld = O.MlpField(D, H, p, nthreads=2)
    xl = np.ascontiguousarray(P.shard_columns(x, rank, world))
    r = sharded_solve(O, fld, xl, rank, world, dist, 0.0, 1.0, tol, tol)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **r)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("D,H,B,tol,scale", [(32, 64, 24, 1e-5, 4.0), (784, 100, 8, 1.4e-8, 1.0)])
def test_two_rank_sharded_solve_matches_unsharded_oracle(oracle, tmp_path, D, H, B, tol, scale):
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, D, H, B, tol, scale, str(tmp_path)), nprocs=2, join=True)
    p = oracle.glorot_mlp_params(D, H, seed=0) * np.float32(scale)
    x = np.random.default_rng(0).random((B, D), dtype=np.float32)
    ref = oracle.solve(oracle.MlpField(D, H, p, nthreads=2), x, 0.0, 1.0, tol, tol, saveat=[1.0], maxiters=10000)
    parts = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(2)]
    for q in parts:
        assert int(q["naccept"]) == ref["stats"]["naccept"] and int(q["nreject"]) == ref["stats"]["nreject"]
        assert np.array_equal(q["dts"], ref["trace"]["dt"])       # identical dt sequence on every rank
    u = np.concatenate([parts[0]["u_end"], parts[1]["u_end"]], axis=0)
    assert np.array_equal(u, ref["u"][-1])
