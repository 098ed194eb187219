"""Parity with the REFERENCE ITSELF, where available: bench/julia_ref/dump_reference.jl (run on a machine that has Julia and
the reference's environment) writes tests/golden/reference_mnist_mlp_b16.npz from the same committed inputs.  Absent in this
repository's build image (no julia) — these tests are skipped then, and the oracle stays "parity unpinned" (DESIGN.md §2)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "reference_mnist_mlp_b16.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="no reference dump (bench/julia_ref/README.md)")

RTOL = 1e-5  # BASELINE.json north_star: rtol 1e-5 fp32, bit-exact accepted-step counts


def _close(a, b):
    sc = max(float(np.abs(b).max()), 1e-30)
    assert float(np.abs(np.asarray(a) - np.asarray(b)).max()) <= RTOL * sc


def _inputs():
    g = np.load(os.path.join(ROOT, "tests", "golden", "mnist_mlp_b16.npz"))
    return g, np.load(REF)


def test_oracle_matches_the_reference(oracle):
    g, r = _inputs()
    tol = float(g["tol"])
    fld = oracle.MlpField(784, 100, g["params"], nthreads=4)
    _close(fld.rhs(g["x"], float(g["t"])), r["k1"])
    nd = oracle.node_forward(fld, g["x"], 0.0, 1.0, tol, tol, mode="none", maxiters=10000)
    assert nd["stats"]["nf"] == int(r["nf_none"]) and nd["stats"]["naccept"] == int(r["naccept_none"])
    _close(nd["u_end"], r["u_end_none"])
    nu = oracle.node_forward(fld, g["x"], 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=float(r["t1"]), maxiters=10000)
    assert nu["nfe"] == int(r["nfe_unbiased"])
    _close(nu["u_end"], r["u_end_unbiased"])
    st = oracle.tsit5_step(fld, g["x"], r["k1"], float(g["t"]), float(g["dt"]), tol, tol)
    _close(st["u"], r["step_u"])


@pytest.mark.gpu
def test_hip_path_matches_the_reference(gpu_pkg):
    import torch
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc
    g, r = _inputs()
    tol = float(g["tol"])
    model = gpu_pkg.TDChain(gpu_pkg.Chain(gpu_pkg.Dense(785, 100, "tanh"), gpu_pkg.Dense(101, 784)))
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(g["params"]))
    x = torch.from_numpy(g["x"]).cuda()
    _close(h.rhs(x, float(g["t"])).cpu().numpy(), r["k1"])
    nd = h.node_forward(x, 0.0, 1.0, tol, tol, mode="none", maxiters=10000)
    assert nd["stats"]["nf"] == int(r["nf_none"]) and nd["stats"]["naccept"] == int(r["naccept_none"])
    _close(nd["u_end"].cpu().numpy(), r["u_end_none"])
    nu = h.node_forward(x, 0.0, 1.0, tol, tol, mode="unbiased", t1_or_rand=float(r["t1"]), maxiters=10000)
    assert nu["nfe"] == int(r["nfe_unbiased"])
    _close(nu["u_end"].cpu().numpy(), r["u_end_unbiased"])
