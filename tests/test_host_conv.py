"""Host-side logic of the conv mirror that needs no GPU: layer specs, topology matching, parameter layout."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _core(P, act="gelu", hc=64, c=8):
    return P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), c + 1, hc), P.BatchNorm(hc, act)),
                             P.Chain(P.Conv((3, 3), hc + 1, hc), P.BatchNorm(hc, act)),
                             P.Conv((3, 3), hc + 1, c)))


def test_conv_topology_matches_the_cifar_block():
    import lrnde_amd as P
    from localregneuralde_jl_amd.conv import conv_topology
    assert conv_topology(_core(P)) == (8, 64, "gelu", 1e-5)
    mlp = P.TDChain(P.Chain(P.Dense(785, 100, "tanh"), P.Dense(101, 784)))
    assert conv_topology(mlp) is None
    with pytest.raises(ValueError):
        conv_topology(P.TDChain(P.Chain(P.Chain(P.Conv((3, 3), 9, 64), P.BatchNorm(64, "gelu")),
                                        P.Chain(P.Conv((3, 3), 65, 32), P.BatchNorm(32, "gelu")), P.Conv((3, 3), 65, 8))))
    with pytest.raises(NotImplementedError):
        P.Conv((5, 5), 9, 64)
    with pytest.raises(NotImplementedError):
        P.Conv((3, 3), 9, 64, use_bias=True)


def test_conv_param_layout_and_count_agree_with_the_oracle_and_the_abi():
    import ctypes as C
    import lrnde_amd as P
    import oracle as O
    from localregneuralde_jl_amd import _lib as L
    p = P.glorot_conv_params(8, 64, seed=3)
    assert np.array_equal(p, O.glorot_conv_params(8, 64, seed=3))
    d = L.ConvDesc(32, 32, 8, 64, L.ACT["gelu"], 1, L.DTYPE["f32"], 1e-5)
    assert p.size == L.lib.lrnde_conv_param_count(C.byref(d)) == O.lib().lro_conv_param_count(8, 64) == 47560
    n1 = 9 * 9 * 64
    assert np.all(p[n1:n1 + 64] == 1.0) and np.all(p[n1 + 64:n1 + 128] == 0.0)  # bn1.scale, bn1.bias


def test_neural_ode_recognises_the_conv_field_without_a_gpu():
    import lrnde_amd as P
    node = P.NeuralODE(_core(P), regularize="unbiased", abstol=1e-4, reltol=1e-4)
    assert node._conv == (8, 64, "gelu", 1e-5) and node.desc is None
    with pytest.raises(ValueError):
        node.handle()  # needs the input to size the handle


def test_conv_create_rejects_unsupported_shapes_before_touching_a_device():
    """shape checks come first in lrnde_conv_create, so the status is visible without a GPU"""
    import ctypes as C
    from localregneuralde_jl_amd import _lib as L
    for (w, h, c, hc, dt) in [(32, 32, 3, 64, 0), (32, 32, 8, 32, 0), (30, 32, 8, 64, 0), (32, 32, 8, 64, 7), (256, 4, 8, 64, 0)]:
        d = L.ConvDesc(w, h, c, hc, L.ACT["gelu"], 1, dt, 1e-5)
        ctx = C.c_void_p()
        assert L.lib.lrnde_conv_create(C.byref(ctx), C.byref(d), 0, None) == 8  # LRNDE_UNSUPPORTED
        assert not ctx.value
    assert L.lib.lrnde_conv_create(None, None, 0, None) == 4  # LRNDE_BADARG
    assert L.lib.lrnde_conv_last_error(None) == b"null handle"
