"""`include/lrnde.h`: one handle is not thread-safe, distinct handles are independent.  Four host threads, each with its own
handle on its own stream (and so its own companion stream, report ring and pinned words), run forward / record / backward
loops at the same time — ctypes drops the GIL for the duration of every call, so the library's host loops really do run
concurrently — and every result must equal, bit for bit, what the same work gives when run alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _work(P, h, x, du, t1s, tol):
    out = []
    for t1 in t1s:
        f = h.node_forward(x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="error_estimate", t1_or_rand=t1, maxiters=10000)
        r = h.node_forward_record(x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="stiffness_estimate", t1_or_rand=t1,
                                  maxiters=10000)
        b = h.node_backward_recorded(du, w_reg=1.5)
        out.append((f["u_end"].clone(), f["reg_val"], f["nfe"], r["u_end"].clone(), r["reg_val"], b["dx"].clone(), b["dp"].clone(),
                    b["stats_bwd"]["naccept"]))
    return out


def test_four_handles_on_four_threads_equal_the_serial_runs(gpu_pkg):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    P = gpu_pkg
    shapes = [(784, 100, 64, 1e-5), (32, 64, 33, 1e-6), (784, 100, 16, 1e-4), (20, 40, 9, 1e-6)]
    jobs = []
    for i, (D, H, B, tol) in enumerate(shapes):
        model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
        p = torch.from_numpy(P.glorot_params(model, seed=i) * np.float32(1.5))
        rng = np.random.default_rng(10 + i)
        x = torch.from_numpy(rng.random((B, D), dtype=np.float32)).cuda()
        du = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
        t1s = [float(v) for v in rng.random(6, dtype=np.float32)]
        jobs.append((model, p, x, du, t1s, tol))

    def make(job, stream):
        h = Handle(_mlp_desc(job[0]), stream=stream); h.set_params(job[1])
        return h

    torch.cuda.synchronize()
    serial = []
    for job in jobs:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            h = make(job, s)
            serial.append(_work(P, h, *job[2:]))
            s.synchronize(); h.close()

    results, errors = [None] * len(jobs), []
    start = threading.Barrier(len(jobs))

    def run(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                h = make(jobs[i], s)
                start.wait()
                results[i] = _work(P, h, *jobs[i][2:])
                s.synchronize(); h.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in th)
    for i, (a, b) in enumerate(zip(serial, results)):
        for k, (ra, rb) in enumerate(zip(a, b)):
            for j, (va, vb) in enumerate(zip(ra, rb)):
                same = torch.equal(va, vb) if isinstance(va, torch.Tensor) else va == vb
                assert same, f"job {i} pass {k} item {j} differs between the threaded and the serial run"
