"""The nranks > 1 code of liblrnde, executed on ONE GPU: R handles of this process (one host thread and one HIP stream
each) joined by the in-process local communicator (include/lrnde_hooks.h) shard a batch exactly as R processes on R GPUs
would — offsets into the partial-sum vectors, the separate receive buffers, the per-step exchange, the sharded
adjoint's parameter-cotangent and norm sums all run; only the transport differs (stream-ordered kernels and events
instead of ncclAllReduce, which RCCL refuses for two ranks on one device).

Bars: forward — dt trace, accept/reject decisions, nfe, reg_val and every element of sol.u[end] EQUAL to the unsharded
run of the same library and to the oracle; backward — tolerance (the sharded sums over the batch are grouped by rank).
BASELINE.json config 3 (B=4096 over 8 ranks of 512) runs here as 8 ranks on one device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, H = 784, 100


def _setup(pkg, B, seed=0):
    model = pkg.TDChain(pkg.Chain(pkg.Dense(D + 1, H, "tanh"), pkg.Dense(H + 1, D)))
    p = pkg.glorot_params(model, seed=seed)
    p = p + np.random.default_rng(seed + 1).standard_normal(p.size).astype(np.float32) * np.float32(0.01)
    x = np.random.default_rng(seed + 2).random((B, D), dtype=np.float32)
    return model, p, x


def _handles(pkg, model, p, nranks, gather):
    """nranks handles on their own streams, joined to one local communicator"""
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    old = os.environ.pop("LRNDE_GATHER_TILES", None)
    if gather:
        os.environ["LRNDE_GATHER_TILES"] = "1"
    try:
        lc = pkg.LocalComm(nranks)
        hs = []
        for r in range(nranks):
            h = Handle(_mlp_desc(model), stream=torch.cuda.Stream())
            h.set_params(torch.from_numpy(p))
            lc.join(h, r)
            hs.append(h)
    finally:
        os.environ.pop("LRNDE_GATHER_TILES", None)
        if old is not None:
            os.environ["LRNDE_GATHER_TILES"] = old
    torch.cuda.synchronize()
    return lc, hs


def _unsharded(pkg, model, p):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(p))
    return h


@pytest.mark.parametrize("nranks,Bl,tol,gather", [(2, 512, 1.4e-8, False), (2, 512, 1.4e-8, True), (4, 64, 1e-5, False),
                                                   (3, 20, 1e-4, True), (2, 2050 // 2, 1e-4, False)])
def test_sharded_forward_equals_unsharded_bitwise(oracle, gpu_pkg, nranks, Bl, tol, gather):
    """solve (with the dt / EEst / accept trace) and the layer forward (local step at t1) on R shards vs one handle"""
    import torch
    P = gpu_pkg
    B = nranks * Bl
    model, p, x = _setup(P, B)
    hu = _unsharded(P, model, p)
    xd = torch.from_numpy(x).cuda()
    sv = [0.37, 1.0]
    ref = hu.solve(xd, 0.0, 1.0, tol, tol, saveat=sv, maxiters=10000, trace=True)
    reff = hu.node_forward(xd, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="stiffness_estimate", t1_or_rand=0.37, maxiters=10000)
    lc, hs = _handles(P, model, p, nranks, gather)
    xs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(x, r, nranks))).cuda() for r in range(nranks)]
    torch.cuda.synchronize()
    got = P.run_ranks([(lambda r=r: hs[r].solve(xs[r], 0.0, 1.0, tol, tol, saveat=sv, maxiters=10000, trace=True)) for r in range(nranks)])
    gotf = P.run_ranks([(lambda r=r: hs[r].node_forward(xs[r], 0.0, 1.0, tol, tol, mode="unbiased", reg_type="stiffness_estimate",
                                                      t1_or_rand=0.37, maxiters=10000)) for r in range(nranks)])
    torch.cuda.synchronize()
    for r in range(nranks):
        assert got[r]["stats"] == ref["stats"], (r, got[r]["stats"], ref["stats"])
        for f in ("t", "dt", "eest", "accepted"):
            assert np.array_equal(got[r]["trace"][f], ref["trace"][f]), (r, f)
        assert np.array_equal(got[r]["t"], ref["t"])
        assert gotf[r]["nfe"] == reff["nfe"] and gotf[r]["reg_val"] == reff["reg_val"] and gotf[r]["t1"] == reff["t1"]
    u = np.concatenate([g["u"].cpu().numpy() for g in got], axis=1)
    assert np.array_equal(u, ref["u"].cpu().numpy()), "saved states differ from the unsharded run"
    ue = np.concatenate([g["u_end"].cpu().numpy() for g in gotf], axis=0)
    assert np.array_equal(ue, reff["u_end"].cpu().numpy())
    assert ref["stats"]["naccept"] > 3
    if B <= 1024:  # and the unsharded ORACLE (seconds of CPU at these sizes)
        fld = oracle.MlpField(D, H, p, nthreads=8)
        ro = oracle.node_forward(fld, x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="stiffness_estimate", t1_or_rand=0.37,
                                 maxiters=10000)
        assert ro["nfe"] == gotf[0]["nfe"] and ro["reg_val"] == gotf[0]["reg_val"]
        assert np.array_equal(ro["u_end"], ue)
    lc.close()


def test_config3_b4096_over_8_ranks(gpu_pkg):
    """BASELINE.json config 3: B=4096 as 8 shards of 512 columns — the layer forward of every rank takes the steps of the
    single-handle B=4096 run (which uses the 16-column kernels: another tile shape, same results) and returns its block"""
    import torch
    P = gpu_pkg
    nranks, Bl, tol = 8, 512, 1e-6
    model, p, x = _setup(P, nranks * Bl)
    hu = _unsharded(P, model, p)
    ref = hu.node_forward(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.61, maxiters=10000)
    lc, hs = _handles(P, model, p, nranks, False)
    xs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(x, r, nranks))).cuda() for r in range(nranks)]
    torch.cuda.synchronize()
    got = P.run_ranks([(lambda r=r: hs[r].node_forward(xs[r], 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=0.61, maxiters=10000))
                       for r in range(nranks)])
    torch.cuda.synchronize()
    for g in got:
        assert g["stats"] == ref["stats"] and g["nfe"] == ref["nfe"] and g["reg_val"] == ref["reg_val"]
    assert np.array_equal(np.concatenate([g["u_end"].cpu().numpy() for g in got], axis=0), ref["u_end"].cpu().numpy())
    lc.close()


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


@pytest.mark.parametrize("nranks,Bl,gather", [(2, 48, False), (2, 48, True), (2, 512, False)])
def test_sharded_backward_and_training_step(gpu_pkg, nranks, Bl, gather):
    """pullback of <g, sol.u[end]> + w_reg*reg_val on R shards: dx is the unsharded dx's block, dp (replicated) its dp —
    2e-4 of the norm: mu is summed over the batch per rank and then over ranks, the error norm adds one fp64 sum per rank;
    the adjoint takes the same number of steps.  Then the classifier head: loss and dpc of the GLOBAL mean loss."""
    import torch
    P = gpu_pkg
    tol = 1e-5
    B = nranks * Bl
    model, p, x = _setup(P, B, seed=3)
    g = (np.random.default_rng(5).standard_normal((B, D)) * 1e-2).astype(np.float32)
    hu = _unsharded(P, model, p)
    ref = hu.node_backward(torch.from_numpy(x).cuda(), 0.0, 1.0, tol, tol, torch.from_numpy(g).cuda(), mode="unbiased",
                           t1_or_rand=0.37, w_reg=2.5, maxiters=5000)
    lc, hs = _handles(P, model, p, nranks, gather)
    xs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(x, r, nranks))).cuda() for r in range(nranks)]
    gs = [torch.from_numpy(np.ascontiguousarray(P.shard_columns(g, r, nranks))).cuda() for r in range(nranks)]
    torch.cuda.synchronize()
    got = P.run_ranks([(lambda r=r: hs[r].node_backward(xs[r], 0.0, 1.0, tol, tol, gs[r], mode="unbiased", t1_or_rand=0.37, w_reg=2.5,
                                                        maxiters=5000)) for r in range(nranks)])
    torch.cuda.synchronize()
    dx = np.concatenate([q["dx"].cpu().numpy() for q in got], axis=0)
    assert _rel(dx, ref["dx"].cpu().numpy()) < 2e-4
    for q in got:
        assert q["stats_fwd"] == ref["stats_fwd"]
        assert abs(q["stats_bwd"]["naccept"] - ref["stats_bwd"]["naccept"]) <= 1
        assert _rel(q["dp"].cpu().numpy(), ref["dp"].cpu().numpy()) < 2e-4
        assert np.array_equal(q["dp"].cpu().numpy(), got[0]["dp"].cpu().numpy()), "dp must come out replicated"
    # classifier + logitcrossentropy on the shards: the mean over the global batch
    K = 10
    rng = np.random.default_rng(7)
    pc = torch.from_numpy((rng.random(K * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).cuda()
    labels = rng.integers(0, K, B).astype(np.int32)
    u = ref["dx"]  # any (B, D) state will do
    cu = hu.classifier_ce(u, pc, K, torch.from_numpy(labels).cuda())
    us = [u[r * Bl:(r + 1) * Bl].contiguous() for r in range(nranks)]
    ls = [torch.from_numpy(labels[r * Bl:(r + 1) * Bl].copy()).cuda() for r in range(nranks)]
    torch.cuda.synchronize()
    cs = P.run_ranks([(lambda r=r: hs[r].classifier_ce(us[r], pc, K, ls[r])) for r in range(nranks)])
    torch.cuda.synchronize()
    for q in cs:
        assert abs(float(q["loss"]) - float(cu["loss"])) <= 2e-6 * abs(float(cu["loss"]))
        assert _rel(q["dpc"].cpu().numpy(), cu["dpc"].cpu().numpy()) < 1e-5
    assert _rel(np.concatenate([q["du"].cpu().numpy() for q in cs], axis=0), cu["du"].cpu().numpy()) < 1e-6
    with pytest.raises(P.LrndeError, match="label"):
        bad = labels.copy(); bad[3] = K
        hu.classifier_ce(u, pc, K, torch.from_numpy(bad).cuda())
    lc.close()
