"""A batch-sharded adaptive Tsit5 loop driven from python over the ORACLE's per-shard step,
with the library's exchange protocol (all-reduce of a zero-padded fp64 partial-sum vector,
fixed-order local sum) — used by the world_size-2 gloo test to cover the N>1 logic on CPU."""
import numpy as np

f32 = np.float32


def _eps(x):
    return np.spacing(f32(abs(x)))


def allreduce_slots(local_sums, rank, nranks, dist):
    """one all-reduce(sum); every rank holds zeros outside its own slot -> exact gather."""
    import torch
    v = torch.zeros(nranks, len(local_sums), dtype=torch.float64)
    v[rank] = torch.tensor(local_sums, dtype=torch.float64)
    if nranks > 1:
        dist.all_reduce(v)
    return v.sum(dim=0).numpy()  # rank-order sum, identical on all ranks


def sharded_solve(O, fld, x_local, rank, nranks, dist, t0, t1, abstol, reltol, maxiters=10000):
    L = O.lib()
    t0, t1, abstol, reltol = f32(t0), f32(t1), f32(abstol), f32(reltol)
    n_global = float(x_local.size * nranks)
    # --- initdt, sums exchanged ---
    f0 = fld.rhs(x_local, t0)
    sk = abstol + np.abs(x_local) * reltol
    r0, r1 = x_local / sk, f0 / sk
    s = allreduce_slots([np.sum((r0 * r0).astype(np.float64)), np.sum((r1 * r1).astype(np.float64))], rank, nranks, dist)
    d0, d1 = f32(np.sqrt(s[0] / n_global)), f32(np.sqrt(s[1] / n_global))
    dtmax = f32(t1 - t0)
    dt0 = f32(1e-6) if (float(d0) < 1e-5 or float(d1) < 1e-5) else f32(f32(d0 / d1) / f32(100))
    dt0 = min(dt0, dtmax)
    f1 = fld.rhs((x_local + dt0 * f0).astype(f32), f32(t0 + dt0))
    r2 = (f1 - f0) / sk
    s = allreduce_slots([np.sum((r2 * r2).astype(np.float64))], rank, nranks, dist)
    d2 = f32(f32(np.sqrt(s[0] / n_global)) / dt0)
    maxd = max(d1, d2)
    if float(maxd) <= 1e-15:
        dt1 = max(f32(1e-6), f32(dt0 * f32(1e-3)))
    else:
        e = f32(f32(-(f32(2) + f32(np.log10(float(maxd))))) / f32(5))
        dt1 = f32(10.0 ** float(e))
    dt = min(f32(f32(100) * dt0), dt1, dtmax)
    # --- loop ---
    gamma, qmin, qmax, qoldinit = f32(0.9), f32(0.2), f32(10), f32(1e-4)
    beta1, beta2 = f32(7.0 / 50.0), f32(2.0 / 25.0)
    dtmin = max(_eps(t1), _eps(t0))
    t, uprev, k1 = t0, x_local, f0
    qold, q11, dtpropose = qoldinit, f32(1), dt
    accept, it, naccept, nreject = False, 0, 0, 0
    u = k7 = None
    dts = []
    while t < t1:
        if it > 0:
            if accept:
                uprev, k1, dt = u, k7, dtpropose
            else:
                dt = f32(dt / min(f32(f32(1) / qmin), f32(q11 / gamma)))
        it += 1
        dt = max(min(dtmax, dt), dtmin)
        dt = min(f32(abs(dt)), f32(abs(t1 - t)))
        assert it <= maxiters and dt > dtmin
        r = O.tsit5_step_sums(fld, uprev, k1, t, dt, abstol, reltol)
        u, k7 = r["u"], r["k7"]
        tot = allreduce_slots(list(r["sums"]), rank, nranks, dist)
        eest = f32(np.sqrt(tot[0] / n_global))
        dts.append(dt)
        if eest == 0:
            q = f32(f32(1) / qmax)
        else:
            q11 = f32(L.lro_fastpow(float(eest), float(beta1)))
            q = f32(q11 / f32(L.lro_fastpow(float(qold), float(beta2))))
            q = max(f32(f32(1) / qmax), min(f32(f32(1) / qmin), f32(q / gamma)))
        accept = bool(eest <= 1)
        if accept:
            naccept += 1
            dtnew = f32(dt / q)
            qold = max(eest, qoldinit)
            ttmp = f32(t + dt)
            t = t1 if abs(f32(ttmp - t1)) < f32(f32(100) * _eps(max(t, t1))) else ttmp
            dtpropose = max(min(dtmax, dtnew), max(_eps(t), dtmin))
        else:
            nreject += 1
    return dict(u_end=u, naccept=naccept, nreject=nreject, dts=np.array(dts, dtype=f32), dt_init=dts[0])
