"""The reference's training step around the NeuralODE layer:
`Chain(flatten, neural_ode, sol_to_arr, classifier=Dense(D => K))` with
`loss = logitcrossentropy(y_pred, y) + w_reg * reg_val` (experiments/src/construct.jl:20-35,180-200) and the
fwd / bwd timing split of `run_training_step` (experiments/src/utils.jl:104-123).  The optimiser update is the
caller's (out of scope)."""
import copy
import time

import numpy as np
import torch


def run_training_step(node, ps, pc, st, x, labels, w_reg, num_classes=10):
    """One forward + pullback.  ps: flat NeuralODE parameters, pc: flat classifier parameters [vec(W) K x D; b]
    (CUDA float32), labels: CUDA int32 (B).  Returns (loss, st_, stats, grads, times) like the reference:
    stats = (y_pred, nfe, ce_loss, reg_val), grads = dict(neural_ode=dps, classifier=dpc, x=dx),
    times = dict(fwd_time, bwd_time) in seconds (wall, synchronised)."""
    t0, t2 = node.tspan
    kw = node.kwargs
    abstol, reltol = kw.get("abstol", 1e-6), kw.get("reltol", 1e-3)
    mode = node.regularize if st["training"] else "none"
    rng = copy.deepcopy(st["rng"])
    r01 = np.float32(rng.random(dtype=np.float32))  # the package's one draw convention (layers.biased_index)
    t1_or_rand = np.float32(r01 * (t2 - t0) + t0) if mode == "unbiased" else r01
    torch.cuda.synchronize()
    tic = time.perf_counter()
    h = node._bind(ps)  # the repack of the (updated) parameters is part of the step
    # (layer forward and loss head in ONE library call, as they sit inside one pullback in the reference: the head's launches
    #  follow the solve's last launch in the queue, the call ends with one synchronisation)
    fw, head = h.node_forward_record_ce(x, t0, t2, abstol, reltol, pc, num_classes, labels, mode=mode, reg_type=node.regularize_type,
                                        t1_or_rand=t1_or_rand, maxiters=node.maxiters, save_start=kw.get("save_start", True))
    loss = np.float32(head["loss"] + np.float32(w_reg) * fw["reg_val"])
    torch.cuda.synchronize()
    fwd_time = time.perf_counter() - tic
    tic = time.perf_counter()
    bw = h.node_backward_recorded(head["du"], w_reg=w_reg)
    torch.cuda.synchronize()
    bwd_time = time.perf_counter() - tic
    st_ = dict(model=st["model"], nfe=fw["nfe"], reg_val=fw["reg_val"], rng=rng, training=st["training"])
    stats = dict(y_pred=head["logits"], nfe=fw["nfe"], ce_loss=head["loss"], reg_val=fw["reg_val"])
    grads = dict(neural_ode=bw["dp"], classifier=head["dpc"], x=bw["dx"])
    return loss, st_, stats, grads, dict(fwd_time=fwd_time, bwd_time=bwd_time, adjoint=bw["stats_bwd"], forward=fw["stats"])


def run_cifar_training_step(node, params, st, x, labels, w_reg, num_classes=10):
    """The CIFAR10 model's forward + pullback (experiments/src/construct.jl:213-227):
    Chain(augment, bn, neural_ode, sol_to_arr, classifier) with loss = logitcrossentropy + w_reg * reg_val.
    params: dict(stem=(156,), neural_ode=(47560,), head=(72+1+K*H*W+K,)) CUDA float32; x (B,3,H,W); labels CUDA int32.
    Returns (loss, st_, stats, grads, times).  The forward keeps its dense record (lrnde_conv_node_forward_record) and the
    pullback runs from it (lrnde_conv_node_backward_recorded), as the reference's training step runs one forward."""
    h = node._bind(params["neural_ode"], torch.empty((x.shape[0], 8, x.shape[2], x.shape[3]), device=x.device))
    node._model_state_in(h, st)
    t0, t2 = node.tspan
    kw = node.kwargs
    abstol, reltol = kw.get("abstol", 1e-6), kw.get("reltol", 1e-3)
    mode = node.regularize if st["training"] else "none"
    rng = copy.deepcopy(st["rng"])
    r01 = np.float32(rng.random(dtype=np.float32))
    t1_or_rand = np.float32(r01 * (t2 - t0) + t0) if mode == "unbiased" else r01
    torch.cuda.synchronize()
    tic = time.perf_counter()
    stem_in = st.get("stem_bn_state") if isinstance(st, dict) else None
    u0, stem_bn = h.cifar_stem_forward(x, params["stem"], stem_in, return_state=True)  # BatchNorm(8)'s state advances too
    fw = h.node_forward_record(u0, t0, t2, abstol, reltol, mode=mode, reg_type=node.regularize_type, t1_or_rand=t1_or_rand,
                               maxiters=node.maxiters, save_start=kw.get("save_start", True))
    bn_after = h.get_bn_state()
    head = h.cifar_head_ce(fw["u_end"], params["head"], num_classes, labels)
    loss = np.float32(head["loss"] + np.float32(w_reg) * fw["reg_val"])
    torch.cuda.synchronize()
    fwd_time = time.perf_counter() - tic
    tic = time.perf_counter()
    bw = h.node_backward_recorded(x.shape[0], head["du"], w_reg=w_reg)  # from the forward's record: no second forward solve
    h.set_bn_state(bn_after)  # (the local step's gradient sweep re-evaluates the field; the model state stays the forward's)
    dstem = h.cifar_stem_backward(x, params["stem"], bw["dx"], stem_in)
    torch.cuda.synchronize()
    bwd_time = time.perf_counter() - tic
    st_ = dict(model=dict(bn_state=bn_after), stem_bn_state=stem_bn, nfe=fw["nfe"], reg_val=fw["reg_val"], rng=rng, training=st["training"])
    stats = dict(y_pred=head["logits"], nfe=fw["nfe"], ce_loss=head["loss"], reg_val=fw["reg_val"])
    grads = dict(stem=dstem, neural_ode=bw["dp"], head=head["dph"])
    return loss, st_, stats, grads, dict(fwd_time=fwd_time, bwd_time=bwd_time, adjoint=bw["stats_bwd"], forward=fw["stats"])
