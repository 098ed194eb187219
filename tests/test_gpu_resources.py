"""Resource hygiene of the C ABI: handles (with their companion context, workspaces, pinned report words, streams and events)
are released by `lrnde_destroy`, and a handle's footprint stops growing once its workspaces are sized — the library sits
inside a training loop that runs for days.  Device memory is read with hipMemGetInfo (`torch.cuda.mem_get_info`), so
allocations made by the library (outside torch's caching allocator) are what is seen."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()     # blocks torch caches for freed tensors go back to the driver: what is left is live memory
    return torch.cuda.mem_get_info()[0]


def _one_round(P, model, ps, pc, x, lab, seed):
    import torch
    node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=1e-4, reltol=1e-4,
                       save_start=False, maxiters=10000)
    st = node.initialstates(np.random.default_rng(seed))
    loss, st, stats, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
    assert np.isfinite(float(loss))
    sol, _ = node(x, ps, st)
    assert torch.isfinite(sol.u[-1]).all()
    node._handle.close()


def test_create_use_destroy_does_not_leak_device_memory(gpu_pkg):
    import torch
    P = gpu_pkg
    D, H, B = 784, 100, 64
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    ps = torch.from_numpy(P.glorot_params(model, seed=0)).cuda()
    pc = torch.zeros(10 * (D + 1), device="cuda")
    x = torch.rand((B, D), device="cuda")
    lab = torch.randint(0, 10, (B,), device="cuda", dtype=torch.int32)
    for i in range(3):                       # warm every lazily created thing (torch allocator blocks, module load)
        _one_round(P, model, ps, pc, x, lab, i)
    gc.collect()
    before = _free_bytes()
    for i in range(40):
        _one_round(P, model, ps, pc, x, lab, 10 + i)
    gc.collect()
    after = _free_bytes()
    # one leaked handle of this shape would hold > 10 MB (dense record, stage buffers, packed weights)
    print(f"40 rounds: free memory changed by {(after - before) / 2**20:+.1f} MiB")
    assert before - after < (4 << 20), f"device memory fell by {(before - after) / 2**20:.1f} MiB over 40 create/use/destroy rounds"
    # the measurement sees what a leak would look like: five live handles hold memory, closing them returns it (the first
    # such peak may leave the runtime with a larger reserve of its own; a second identical peak must not add to it)
    def peak():
        held = []
        for i in range(5):
            node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=1e-4, reltol=1e-4,
                               save_start=False, maxiters=10000)
            P.run_training_step(node, ps, pc, node.initialstates(np.random.default_rng(i)), x, lab, 2.5)
            held.append(node)
        live = _free_bytes()
        for node in held:
            node._handle.close()
        held.clear()
        return live, _free_bytes()
    live1, rest1 = peak()
    live2, rest2 = peak()
    live3, rest3 = peak()
    print(f"five live handles hold {(after - live1) / 2**20:.1f} MiB; after closing them free memory is "
          f"{(rest1 - after) / 2**20:+.1f}, {(rest2 - after) / 2**20:+.1f}, {(rest3 - after) / 2**20:+.1f} MiB from the start")
    assert after - live1 > (10 << 20)
    assert rest2 - rest3 < (4 << 20) and rest1 - rest3 < (4 << 20)


def test_a_long_lived_handle_stops_allocating(gpu_pkg):
    import torch
    P = gpu_pkg
    D, H, B = 784, 100, 128
    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    ps = torch.from_numpy(P.glorot_params(model, seed=0)).cuda()
    pc = torch.zeros(10 * (D + 1), device="cuda")
    x = torch.rand((B, D), device="cuda")
    lab = torch.randint(0, 10, (B,), device="cuda", dtype=torch.int32)
    node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=1e-5, reltol=1e-5,
                       save_start=False, maxiters=10000)
    st = node.initialstates(np.random.default_rng(0))
    for _ in range(5):
        loss, st, *_ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
    before = _free_bytes()
    for _ in range(60):
        loss, st, stats, grads, _ = P.run_training_step(node, ps, pc, st, x, lab, 2.5)
        sol, _ = node(x, ps, st)
    after = _free_bytes()
    # (the dense record may double once if a pass needs more steps than any before it: 128 columns x 784 x 7 slots x 4 B
    #  per step is 2.8 MB per step recorded; anything steady would show as tens of MB over 60 passes)
    assert before - after < (48 << 20), f"device memory fell by {(before - after) / 2**20:.1f} MiB over 60 steps on one handle"


def test_conv_and_sde_handles_release_their_memory(gpu_pkg):
    import torch
    P = gpu_pkg
    W = H = 8; B = 4; K = 10
    rng = np.random.default_rng(12)
    core = P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")),
                             P.Chain(P.Conv((3, 3), 65, 64), P.BatchNorm(64, "gelu")), P.Conv((3, 3), 65, 8)))
    pstem = (rng.standard_normal(156) * 0.3).astype(np.float32); pstem[140:148] = 1.0; pstem[148:156] = 0.0
    params = dict(stem=torch.from_numpy(pstem).cuda(), neural_ode=torch.from_numpy(P.glorot_conv_params(8, 64, seed=0)).cuda(),
                  head=torch.from_numpy((rng.standard_normal(73 + K * H * W + K) * 0.1).astype(np.float32)).cuda())
    x = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
    Ds, Hs, Bs = 32, 64, 64
    pd = torch.from_numpy((rng.standard_normal(Hs * Ds + Hs + Ds * Hs + Ds) * 0.1).astype(np.float32)).cuda()
    pg = torch.from_numpy((rng.standard_normal(Ds * Ds + Ds) * 0.1).astype(np.float32)).cuda()
    xs = torch.from_numpy(rng.standard_normal((Bs, Ds)).astype(np.float32)).cuda()

    def one_round(i):
        node = P.NeuralODE(core, regularize="unbiased", abstol=1e-3, reltol=1e-3, save_start=False, maxiters=2000)
        st = node.initialstates(np.random.default_rng(i))
        loss, *_ = P.run_cifar_training_step(node, params, st, x, lab, 2.5)
        assert np.isfinite(loss)
        node._handle.close()
        sde = P.NeuralDSDE(P.Chain(P.Dense(Ds, Hs, "tanh"), P.Dense(Hs, Ds)), P.Dense(Ds, Ds), regularize="unbiased", nsteps=8,
                           abstol=0.14, reltol=0.14, nfine=32)
        sts = sde.initialstates(np.random.default_rng(i))
        dx, dps, info = sde.pullback(xs, dict(drift=pd, diffusion=pg), sts, torch.ones_like(xs), w_reg=1.0)
        assert torch.isfinite(dx).all()
        sde._handle.close()

    for i in range(3):
        one_round(i)
    gc.collect()
    before = _free_bytes()
    for i in range(25):
        one_round(10 + i)
    gc.collect()
    after = _free_bytes()
    assert before - after < (4 << 20), f"device memory fell by {(before - after) / 2**20:.1f} MiB over 25 conv + SDE rounds"
