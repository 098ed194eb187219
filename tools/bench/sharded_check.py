# Check of the sharded path on a multi-GPU node (this session's one-GPU box cannot run it: RCCL refuses two ranks on
# one device, "invalid usage"):
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/bench/sharded_check.py
# Every rank solves its block of columns through the library's RCCL communicator (torch.distributed only broadcasts
# the unique id, over gloo); the sharded forward must equal the unsharded oracle bit for bit, the sharded adjoint to
# tolerance.
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
import lrnde_amd as P
from localregneuralde_jl_amd.layers import Handle, _mlp_desc
D, H, Bl = 784, 100, 24
model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
p = P.glorot_params(model, seed=0)
xg = np.random.default_rng(0).random((Bl * world, D), dtype=np.float32)
h = Handle(_mlp_desc(model)); h.set_params(torch.from_numpy(p))
try:
    P.init_comm(h, rank, world)
except Exception as e:
    print(f"rank {rank}: init_comm failed: {e}", flush=True)
    dist.destroy_process_group(); sys.exit(2)  # a communicator that cannot be built is a FAILURE of this check
nr, kind = h.comm_count()
comm_ok = (nr == world and kind == 1)   # the library's communicator is RCCL and has WORLD_SIZE ranks
print(f"rank {rank}: lrnde_comm_count = {nr} (kind {kind}) for WORLD_SIZE {world} -> {'OK' if comm_ok else 'MISMATCH'}", flush=True)
x = torch.from_numpy(np.ascontiguousarray(P.shard_columns(xg, rank, world))).cuda()
r = h.node_forward(x, 0.0, 1.0, 1e-5, 1e-5, mode="unbiased", t1_or_rand=0.37, maxiters=2000)
import oracle as O
fld = O.MlpField(D, H, p, nthreads=4)
ref = O.node_forward(fld, xg, 0.0, 1.0, 1e-5, 1e-5, mode="unbiased", t1_or_rand=0.37, maxiters=2000)
mine = ref["u_end"][rank * Bl:(rank + 1) * Bl]
ok = comm_ok and r["nfe"] == ref["nfe"] and r["reg_val"] == ref["reg_val"] and np.array_equal(r["u_end"].cpu().numpy(), mine)
print(f"rank {rank}: nfe {r['nfe']} vs {ref['nfe']}, reg {r['reg_val']} vs {ref['reg_val']}, u_end equal {np.array_equal(r['u_end'].cpu().numpy(), mine)} -> {'OK' if ok else 'MISMATCH'}", flush=True)
# sharded adjoint
g = np.random.default_rng(4).standard_normal(xg.shape).astype(np.float32)
bo = O.node_backward(fld, xg, 0.0, 1.0, 1e-5, 1e-5, g, mode="unbiased", t1_or_rand=0.37, w_reg=2.5, maxiters=5000)
bg = h.node_backward(x, 0.0, 1.0, 1e-5, 1e-5, torch.from_numpy(g[rank * Bl:(rank + 1) * Bl]).cuda(), mode="unbiased", t1_or_rand=0.37, w_reg=2.5, maxiters=5000)
rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))
rdx, rdp = rel(bg['dx'].cpu().numpy(), bo['dx'][rank*Bl:(rank+1)*Bl]), rel(bg['dp'].cpu().numpy(), bo['dp'])
print(f"rank {rank}: adjoint naccept {bg['stats_bwd']['naccept']} vs {bo['stats_bwd']['naccept']}  dx rel {rdx:.2e}  dp rel {rdp:.2e}", flush=True)
ok = ok and rdx < 2e-3 and rdp < 2e-3
flag = torch.tensor([0 if ok else 1])
dist.all_reduce(flag)
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 0 else 1)  # non-zero if ANY rank saw a mismatch
