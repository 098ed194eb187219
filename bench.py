#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

One "step" = one NeuralODE forward pass over one batch: the adaptive Tsit5 solve of the MNIST-ODE
MLP field on (0,1) at abstol=reltol=1.4e-8 (experiments/mnist_ode/mlp.yml) plus the local
regularisation step (regularize=:unbiased, :error_estimate), B=512 columns per GPU, inputs
resident in HBM.  Metric: NFE/s = vector-field evaluations (of one 512-column shard) per second,
summed over ranks.  Launch: `python bench.py --gpus 1` or, for N>1,
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

D, H = 784, 100
FLOP_PER_FEVAL_PER_COL = 2 * (785 * 100 + 101 * 784)  # 315 368
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def cpu_baseline(params, x, tol, t1, cores):
    """The oracle (a port, not the Julia reference — no julia in the image) timed on the host
    cores on one full forward pass of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    fld = O.MlpField(D, H, params, nthreads=cores)
    t0 = time.time()
    nfe, passes = 0, 0
    while True:  # bounded sample: repeat the pass for >= ~10 s of CPU work
        r = O.node_forward(fld, x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="error_estimate", t1_or_rand=t1,
                           maxiters=10000)
        nfe += r["nfe"]
        passes += 1
        if time.time() - t0 > 10.0 or passes >= 200:
            break
    el = time.time() - t0
    return r, el, nfe, passes


def cpu_baseline_numpy(params, x, tol, t1, cores):
    """SURVEY.md §8(d)(2) / BASELINE.md §3.2: the numpy/OpenBLAS restatement of the same forward pass (float32 sgemm for
    the two Dense layers — the BLAS family Julia's Dense uses — numpy broadcasts for the stage arithmetic, as the
    reference's unfused broadcasts), OpenBLAS limited to `cores` threads.  A restatement, not the Julia reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import np_restatement as R
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover
        threadpool_limits = None
    f = R.NpMlp(D, H, params)
    ctx = threadpool_limits(limits=cores, user_api="blas") if threadpool_limits else None
    try:
        R.node_forward(f, x, 0.0, 1.0, tol, tol, t1, fast=True)  # warm (page in BLAS, allocate)
        t0 = time.time()
        nfe, passes = 0, 0
        while True:
            r = R.node_forward(f, x, 0.0, 1.0, tol, tol, t1, fast=True)
            nfe += r["nfe"]
            passes += 1
            if time.time() - t0 > 10.0 or passes >= 400:
                break
        el = time.time() - t0
    finally:
        if ctx is not None:
            ctx.unregister() if hasattr(ctx, "unregister") else ctx.__exit__(None, None, None)
    return r, el, nfe, passes


def cpu_baseline_torch(params, x, tol, t1, cores):
    """Third CPU leg: the same restatement with the field on torch's CPU kernels — MKL sgemm (addmm) and
    the vectorised tanh, `cores` intra-op threads — while numpy's OpenBLAS is held to one thread for the stage sums (two
    spinning thread pools on the same cores cost 30x).  A restatement, not the Julia reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import np_restatement as R
    from threadpoolctl import threadpool_limits
    f = R.TorchMlp(D, H, params, threads=cores)
    with threadpool_limits(limits=1, user_api="blas"):
        torch.set_num_threads(cores)
        R.node_forward(f, x, 0.0, 1.0, tol, tol, t1, fast=True)
        t0 = time.time()
        nfe, passes = 0, 0
        while True:
            r = R.node_forward(f, x, 0.0, 1.0, tol, tol, t1, fast=True)
            nfe += r["nfe"]
            passes += 1
            if time.time() - t0 > 8.0 or passes >= 400:
                break
        el = time.time() - t0
    return r, el, nfe, passes


def measured_traffic(key):
    """HBM/fabric bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this round
    (profiles/r<N>/traffic.json, newest round first: 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 corrections of MI355X_MICROARCH.md, collected in
    their own passes) — or None when no committed measurement exists for this workload.  bench.py does not guess it."""
    for rnd in ("r3", "r2", "r1"):
        f = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        if os.path.exists(f):
            try:
                v = json.load(open(f)).get(key)
            except (OSError, ValueError):
                v = None
            if v is not None:
                return v
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="columns per GPU (weak scaling)")
    ap.add_argument("--tol", type=float, default=1.4e-8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=512)
    ap.add_argument("--no-conv", action="store_true", help="skip the 28x28 conv-field side measurement of the default run")
    ap.add_argument("--adjoint-steps", type=int, default=30, help="timed forward+adjoint passes (single GPU)")
    ap.add_argument("--sustain-s", type=float, default=1.0, help="length of the sustained leg (seconds of forward passes)")
    ap.add_argument("--workload", default="mlp", choices=["mlp", "cifar_conv_bf16", "cifar_conv_f32", "cifar_conv_f32_split", "mnist_conv_f32", "mnist_conv_f32_split", "mnist_sde"],  # *_split: opt-in fast mode, never part of the default line
                    help="mlp: the headline MNIST-ODE MLP field (default).  The conv workloads time the CIFAR10 node_core "
                         "(BASELINE.json configs 4 and 2-ii); single GPU.")
    args = ap.parse_args()
    if args.workload == "mnist_sde":
        return sde_main(args)
    if args.workload != "mlp":
        return conv_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    # LRNDE_BENCH_FORCE_DIST=1 (with LRNDE_FORCE_COMM=1, under torch.distributed.run --nproc-per-node 1): the whole N > 1 code
    # path of this file - process group, library communicator, signature all-gather, exchange timing - on a one-GPU box
    force_dist = os.environ.get("LRNDE_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import lrnde_amd as P
    from localregneuralde_jl_amd.layers import Handle, _mlp_desc

    model = P.TDChain(P.Chain(P.Dense(D + 1, H, "tanh"), P.Dense(H + 1, D)))
    params = P.glorot_params(model, seed=0)  # random-init weights of the reference architecture
    Bg = args.batch * world
    xg = np.random.default_rng(0).random((Bg, D), dtype=np.float32)  # MNIST pixels lie in [0,1]
    x = torch.from_numpy(np.ascontiguousarray(P.shard_columns(xg, rank, world))).cuda()
    h = Handle(_mlp_desc(model))
    h.set_params(torch.from_numpy(params))
    if dist:
        P.init_comm(h, rank, world)
    t1s = np.random.default_rng(1).random(args.steps + args.warmup, dtype=np.float32)  # host RNG draw of t1

    def one_pass(i):
        return h.node_forward(x, 0.0, 1.0, args.tol, args.tol, mode="unbiased", reg_type="error_estimate",
                              t1_or_rand=float(t1s[i]), maxiters=10000)

    for i in range(args.warmup):
        one_pass(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    import hashlib
    import struct
    sig = hashlib.sha256()  # what every rank must agree on, pass by pass: the controller's decisions and the regulariser

    def sign(rr):
        st = rr["stats"]
        sig.update(struct.pack("<5i4f", rr["nfe"], st["naccept"], st["nreject"], st["nf"], st["iters"], st["dt_init"], st["dt_final"],
                               st["eest_last"], float(rr["reg_val"])))

    t0 = time.perf_counter()
    nfe_total, steps_total = 0, 0
    for i in range(args.steps):
        r = one_pass(args.warmup + i)
        nfe_total += r["nfe"]
        steps_total += r["stats"]["naccept"] + r["stats"]["nreject"] + 1
        sign(r)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # Self-verification of the N > 1 run (the first RCCL run with more than one rank is the driver's): the communicator the
    # library built has WORLD_SIZE ranks, and every rank took the SAME steps in every timed pass (nfe, accepted, rejected,
    # first / last dt, last EEst, reg_val: one all-reduce per attempted step makes them equal by construction — if a rank
    # drifts, its collectives no longer pair up and the numbers above mean nothing).  Any mismatch: non-zero exit.
    rccl_nranks, comm_kind = h.comm_count()
    verify = {"rccl_nranks": rccl_nranks, "comm_kind": {0: "none", 1: "rccl", 2: "local"}[comm_kind],
              "ranks_took_identical_steps": None, "allreduce_us_per_step": None}
    if dist:
        mine = torch.tensor(list(sig.digest()[:16]), dtype=torch.int64, device="cuda")
        allsig = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allsig, mine)
        same = all(bool(torch.equal(a, allsig[0])) for a in allsig)
        verify["ranks_took_identical_steps"] = same
        verify["allreduce_us_per_step"] = h.bench_exchange(args.batch, reps=100)  # collective: every rank calls it
        if rccl_nranks != world or comm_kind != 1 or not same:
            if rank == 0:
                print(json.dumps({"error": "sharded run failed its self-check", "world": world, **verify}), file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)

    # sustained leg: >= --sustain-s seconds of forward passes in one timed region (the driver's --steps 20 is a 34-ms sample)
    sustained = None
    if args.sustain_s > 0:
        npass = max(int(np.ceil(args.sustain_s / (el / args.steps))), args.steps)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        nfe_s = 0
        for i in range(npass):
            nfe_s += one_pass(i % (args.steps + args.warmup))["nfe"]
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        els = time.perf_counter() - ts0
        if dist:
            tt = torch.tensor([els], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            els = float(tt.item())
        sustained = {"passes": npass, "seconds": els, "ms_per_pass": els / npass * 1e3, "nfe_per_s": world * nfe_s / els}

    # forward + adjoint ms/batch (the second half of BASELINE.json's metric): the reference's training step
    # (experiments/src/utils.jl:104-123) = pullback of  logitcrossentropy(classifier(sol.u[end]), y) + w_reg*reg_val
    # through Chain(neural_ode, classifier Dense(784=>10)); fwd/bwd split timed as there.  Timed on one GPU only
    # (the scaling metric is the forward NFE/s; the sharded adjoint is exercised by tests/ and tools/bench/sharded_check.py).
    fwd_adj_ms, bwd_stats = None, None
    if world == 1 and args.adjoint_steps > 0:
        node = P.NeuralODE(model, regularize="unbiased", regularize_type="error_estimate", abstol=args.tol, reltol=args.tol,
                           save_start=False, maxiters=10000)
        node._handle = h  # reuse the bench handle; run_training_step repacks the parameters on every call (inside fwd_time)
        ps_d = torch.from_numpy(params).cuda()
        rngc = np.random.default_rng(2)
        pc = torch.from_numpy((rngc.random(10 * (D + 1), dtype=np.float32) - np.float32(0.5)) *
                              np.float32(np.sqrt(24.0 / (D + 10)))).cuda()
        labels = torch.from_numpy(rngc.integers(0, 10, args.batch).astype(np.int32)).cuda()
        st0 = node.initialstates(np.random.default_rng(3))
        P.run_training_step(node, ps_d, pc, st0, x, labels, 2.5)
        fw_t, bw_t, adj_acc, adj_nf = [], [], [], []
        for i in range(args.adjoint_steps):
            st_i = dict(st0, rng=np.random.default_rng(100 + i))
            loss, _, tstats, grads, times = P.run_training_step(node, ps_d, pc, st_i, x, labels, 2.5)
            fw_t.append(times["fwd_time"]); bw_t.append(times["bwd_time"])
            adj_acc.append(times["adjoint"]["naccept"]); adj_nf.append(times["adjoint"]["nf"])
        fwd_adj_ms = float(np.median(np.add(fw_t, bw_t))) * 1e3   # the MEDIAN pass (the mean rides along below)

        def dist3(v):
            v = np.asarray(v) * 1e3
            return {"min": float(v.min()), "median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "mean": float(v.mean())}
        bwd_stats = {"passes": args.adjoint_steps, "train_fwd_ms": dist3(fw_t), "train_bwd_ms": dist3(bw_t),
                     "fwd_plus_adjoint_ms": dist3(np.add(fw_t, bw_t)),
                     "train_fwd_ms_median": float(np.median(fw_t)) * 1e3, "train_bwd_ms_median": float(np.median(bw_t)) * 1e3,
                     "adjoint_naccept_per_pass": [int(a) for a in adj_acc], "adjoint_nf_per_pass": [int(a) for a in adj_nf],
                     "adjoint_naccept": times["adjoint"]["naccept"], "adjoint_nreject": times["adjoint"]["nreject"],
                     "adjoint_nf": times["adjoint"]["nf"], "loss": float(loss), "w_reg": 2.5,
                     "what": "run_training_step: node forward with dense record + Dense(784=>10) + logitcrossentropy "
                             "(the regulariser's reverse sweep depends on the forward alone and runs on the handle's companion "
                             "stream from the moment sol(t1) exists, beside the rest of the solve), then continuous adjoint + "
                             "classifier cotangents + the sweep's gradient added"}

    # roofline leg: the dominant kernel (one full Tsit5 step per launch).  Two HIP-event clocks on the handle's stream:
    #  (a) us_per_launch: 100 back-to-back launches of the step kernel on fixed inputs (lrnde_bench_step) — the number
    #      the roofline fraction is computed from;
    #  (b) in_solve: events around the kernels of one real solve (2 init launches + full steps + the terminal /
    #      speculative launches that find the solve finished), divided by its full steps — what a per-dispatch profile of
    #      the same command averages to for FULL steps, minus the profiler's own per-dispatch overhead
    #      (profiles/r2/README.md compares the three).
    k1 = h.rhs(x, 0.0)
    dt_typ = float(r["stats"]["dt_final"]) if r["stats"]["dt_final"] > 0 else 0.02
    us_batches = sorted(h.bench_step(x, k1, 0.0, dt_typ, args.tol, args.tol, reps=100) for _ in range(5))
    us = us_batches[2]  # median of five batches of 100 launches (each batch: one HIP-event pair, mean over its launches)
    flop_per_launch = 6 * FLOP_PER_FEVAL_PER_COL * args.batch
    achieved = flop_per_launch / (us * 1e-6) / 1e12
    h.last_solve_kernel_ms()  # arms the solve's event pair (off in the timed passes above: two marker packets per solve)
    rs = one_pass(args.warmup + args.steps - 1)
    solve_ms, solve_launches = h.last_solve_kernel_ms()
    full_steps = rs["stats"]["naccept"] + rs["stats"]["nreject"]

    out = {
        "metric": "NFE/s (vector-field evals/s inside the adaptive Tsit5 NeuralODE forward, MNIST-ODE B=512/GPU)",
        "value": world * nfe_total / el,
        "unit": "NFE/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "MNIST-ODE MLP field TDChain(Dense(785=>100,tanh),Dense(101=>784)), "
                               f"B={args.batch}/GPU, Tsit5 adaptive abstol=reltol={args.tol:g}, tspan=(0,1), "
                               "regularize=unbiased/error_estimate, forward pass (solve + local reg step)",
                   "global_batch": Bg, "parallelism": f"batch-shard x{world}",
                   "nfe_per_pass": nfe_total / args.steps, "rk_steps_per_sec": world * steps_total / el,
                   "fwd_ms_per_batch": el / args.steps * 1e3,
                   "fwd_plus_adjoint_ms_per_batch": fwd_adj_ms, "adjoint": bwd_stats,
                   "sustained": sustained, "sharded_self_check": verify,
                   "nfe_note": "nfe_per_pass is the count THIS arithmetic takes: at abstol=reltol=1.4e-8 (below fp32 eps) the "
                               "embedded error estimate is rounding noise of the field, so the accepted-step count depends on "
                               "the summation order of the dense layers - this kernel order 35 steps, OpenBLAS sgemm 40, a float64 "
                               "field 20, all with sol.u[end] equal to 5e-7 (DESIGN.md 2); the Julia reference would count its "
                               "own BLAS's number"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                     # HBM/fabric bytes per launch: read from the committed PMC summary of this round (measured_traffic),
                     # null for any workload that has no committed measurement
                     "traffic": measured_traffic(f"k_step_q_b{args.batch}") if world == 1 else None,
                     "kernel": "k_step_q<false, 1> (one attempted Tsit5 step: 6 f-evals, fused stage combination, error norm)"
                               if args.batch <= 2048 else "k_step<4,false>",
                     "us_per_launch": us, "us_per_launch_batches": us_batches, "flop_per_launch": flop_per_launch,
                     "in_solve": {"solve_kernel_ms": solve_ms, "step_launches": solve_launches, "full_steps": full_steps,
                                  "us_per_full_step": solve_ms * 1e3 / max(full_steps, 1),
                                  "what": "HIP events around all kernels of one adaptive solve (2 init + steps + terminal/"
                                          "speculative launches) / full steps: an upper bound of the step kernel's in-solve duration"}},
    }
    if rank == 0 and world == 1 and not args.no_conv:
        # BASELINE.json configs[1] read literally ("28x28 conv vector field, batch=512 fp32"): the CIFAR block topology on a
        # 28x28x8 state (not a model of the reference) — measured alongside, `--workload
        # mnist_conv_f32` gives its full line
        out["config"]["conv_field_28x28_b512"] = conv_measure(args, "mnist_conv_f32", brief=True)
        # BASELINE.json configs[3] (CIFAR10 block, B=256, bf16): reported next to the fp32 handle on the same inputs — at the
        # reference's tolerance 1e-4 the bf16 field's rounding noise drives the controller (NFE counts below)
        out["config"]["cifar_conv_bf16_b256"] = conv_measure(args, "cifar_conv_bf16", brief=True)
        out["config"]["cifar_conv_f32_b256"] = conv_measure(args, "cifar_conv_f32", brief=True)
        # BASELINE.json configs[4] (MNIST-SDE, B=512): the NeuralDSDE layer's forward and pullback; `--workload mnist_sde` gives its line
        out["config"]["mnist_sde_layer_b512"] = sde_layer_side_measurement(512)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the GPU box gives one GPU's share of the host: 16 cores (os.cpu_count() reports the whole host)
        cores = int(os.environ.get("LRNDE_CPU_CORES", min(len(os.sched_getaffinity(0)), 16)))
        cb = min(args.cpu_batch, args.batch)
        ref, cel, cnfe, cpasses = cpu_baseline(params, xg[:cb], args.tol, float(t1s[args.warmup + args.steps - 1]), cores)
        port = {"value": cnfe / cel * (cb / args.batch), "unit": "NFE/s", "cores": cores, "kind": "port",
                "sample": f"{cpasses} forward passes ({cnfe} f-evals) of the same workload at B={cb} with "
                          f"the C oracle (OpenMP, {cores} threads, canonical fma chains) in {cel:.1f} s"}
        if cb == args.batch:
            same = (ref["nfe"] == r["nfe"]) and bool(np.array_equal(ref["u_end"], r["u_end"].cpu().numpy()))
            port["gpu_matches_oracle_bitwise"] = same
        nref, nel, nnfe, npasses = cpu_baseline_numpy(params, xg[:cb], args.tol, float(t1s[args.warmup + args.steps - 1]), cores)
        rest = {"value": nnfe / nel * (cb / args.batch), "unit": "NFE/s", "cores": cores, "kind": "port",
                "sample": f"{npasses} forward passes ({nnfe} f-evals, {nref['naccept']} accepted steps per pass) of the same "
                          f"workload at B={cb} with the numpy/OpenBLAS restatement (float32 sgemm, {cores} BLAS threads) in {nel:.1f} s",
                "label": "restatement - not the Julia reference",
                "u_end_vs_gpu_max_err_of_scale": float(np.abs(nref["u_end"] - r["u_end"].cpu().numpy()).max() /
                                                       np.abs(nref["u_end"]).max()) if cb == args.batch else None}
        if cb == args.batch:
            # ... and the ADJOINT against the oracle on this workload: the pullback of <g, sol.u[end]> +
            # 2.5 reg_val with g = the GPU classifier head's cotangent, GPU vs C oracle: equal step counts, equal bits
            import oracle as O
            fldb = O.MlpField(D, H, params, nthreads=cores)
            t1b = float(t1s[args.warmup + args.steps - 1])
            rngc = np.random.default_rng(2)
            pcb = torch.from_numpy((rngc.random(10 * (D + 1), dtype=np.float32) - np.float32(0.5)) * np.float32(np.sqrt(24.0 / (D + 10)))).cuda()
            labb = torch.from_numpy(rngc.integers(0, 10, args.batch).astype(np.int32)).cuda()
            gdev = h.classifier_ce(r["u_end"], pcb, 10, labb)["du"]
            tb0 = time.time()
            bo = O.node_backward(fldb, xg[:cb], 0.0, 1.0, args.tol, args.tol, gdev.cpu().numpy(), mode="unbiased", t1_or_rand=t1b, w_reg=2.5)
            tb1 = time.time()
            bg = h.node_backward(x, 0.0, 1.0, args.tol, args.tol, gdev, mode="unbiased", t1_or_rand=t1b, w_reg=2.5, maxiters=10000)
            keys = ("naccept", "nreject", "nf")
            port["adjoint_counts_match_oracle"] = all(bg["stats_bwd"][k] == bo["stats_bwd"][k] for k in keys)
            port["adjoint_counts"] = {"gpu": {k: bg["stats_bwd"][k] for k in keys}, "oracle": {k: bo["stats_bwd"][k] for k in keys}}
            port["adjoint_bits_match_oracle"] = bool(np.array_equal(bg["dx"].cpu().numpy(), bo["dx"]) and np.array_equal(bg["dp"].cpu().numpy(), bo["dp"]))
            port["oracle_fwd_plus_adjoint_s"] = tb1 - tb0
        tref, tel, tnfe, tpasses = cpu_baseline_torch(params, xg[:cb], args.tol, float(t1s[args.warmup + args.steps - 1]), cores)
        tleg = {"value": tnfe / tel * (cb / args.batch), "unit": "NFE/s", "cores": cores, "kind": "port",
                "sample": f"{tpasses} forward passes ({tnfe} f-evals, {tref['naccept']} accepted steps per pass) of the same workload at "
                          f"B={cb} with the field on torch CPU kernels (MKL sgemm, {cores} threads; stage sums numpy) in {tel:.1f} s",
                "label": "restatement - not the Julia reference"}
        # the reported baseline is the FASTEST of the three CPU restatements; the others ride along
        legs = sorted([rest, port, tleg], key=lambda d: -d["value"])
        best, other = legs[0], legs[1:]
        out["cpu_baseline"] = dict(best, other_baselines=other)
        out["config"]["gpu_over_best_cpu_restatement"] = out["value"] / best["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


# ---- conv vector field workloads (BASELINE.json config 4: CIFAR10 block B=256 bf16; config 2-ii: 28x28 conv field) ----
CONV_FLOP_PER_PIXEL = 2 * (81 * 64 + 585 * 64 + 585 * 8)  # 94 608 (96 878 592 per 32x32 sample)
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md dense bf16


def conv_main(args):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("the conv workloads are single-GPU (train-mode BatchNorm couples the batch: replicas only)")
    torch.cuda.set_device(0)
    print(json.dumps(conv_measure(args, args.workload)))


def conv_measure(args, workload, brief=False):
    import lrnde_amd as P
    W, H, B, dt, tol, train = {"cifar_conv_bf16": (32, 32, 256, "bf16", 1e-4, True),
                               "cifar_conv_f32": (32, 32, 256, "f32", 1e-4, True),
                               "cifar_conv_f32_split": (32, 32, 256, "f32_split", 1e-4, True),
                               "mnist_conv_f32": (28, 28, 512, "f32", 1e-4, False),
                               "mnist_conv_f32_split": (28, 28, 512, "f32_split", 1e-4, False)}[workload]
    if args.batch != 512 and not brief:
        B = args.batch
    steps, warmup = (3, 1) if brief else (min(args.steps, 10), min(args.warmup, 2))
    params = P.glorot_conv_params(8, 64, seed=0)
    xh = np.random.default_rng(0).standard_normal((B, 8, H, W)).astype(np.float32)  # u0 ~ N(0,1)
    x = torch.from_numpy(xh).cuda()
    h = P.ConvHandle(W, H, 8, 64, act="gelu", bn_train=train, compute_dtype=dt)
    h.set_params(params)
    t1s = np.random.default_rng(1).random(steps + warmup, dtype=np.float32)

    def one_pass(i):
        return h.node_forward(x, 0.0, 1.0, tol, tol, mode="unbiased", reg_type="error_estimate",
                              t1_or_rand=float(t1s[i]), maxiters=10000)

    for i in range(warmup):
        one_pass(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nfe_total, steps_total = 0, 0
    for i in range(steps):
        r = one_pass(warmup + i)
        nfe_total += r["nfe"]
        steps_total += r["stats"]["naccept"] + r["stats"]["nreject"] + 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fwd_adj_ms, bwd = None, None
    if args.adjoint_steps > 0 and not brief:  # (bf16: bf16 forward re-solve, fp32 adjoint) pullback of <g, sol.u[end]> + 2.5*reg_val, g ~ 1e-3*N(0,1) (a mean-loss cotangent's size)
        g = torch.from_numpy((np.random.default_rng(2).standard_normal(xh.shape) * 1e-3).astype(np.float32)).cuda()
        nb = min(args.adjoint_steps, 3)
        h.node_backward(x, 0.0, 1.0, tol, tol, g, mode="unbiased", t1_or_rand=float(t1s[0]), w_reg=2.5, maxiters=10000)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(nb):
            rb = h.node_backward(x, 0.0, 1.0, tol, tol, g, mode="unbiased", t1_or_rand=float(t1s[i % len(t1s)]), w_reg=2.5,
                                 maxiters=10000)
        torch.cuda.synchronize()
        fwd_adj_ms = (time.perf_counter() - tb) / nb * 1e3
        bwd = {"adjoint_naccept": rb["stats_bwd"]["naccept"], "adjoint_nreject": rb["stats_bwd"]["nreject"],
               "adjoint_nf": rb["stats_bwd"]["nf"]}
    us = h.bench_rhs(x, 0.3, reps=20)  # HIP events on the handle's stream around 20 f-evals
    flop = CONV_FLOP_PER_PIXEL * W * H * B
    # f32_split: three fp16 MFMAs per product -> a third of the fp16 (= bf16) dense rate bounds conv2/conv3
    peak = {"bf16": PEAK_BF16_MFMA_TFLOPS, "f32_split": PEAK_BF16_MFMA_TFLOPS / 3.0}.get(dt, PEAK_F32_MFMA_TFLOPS)
    achieved = flop / (us * 1e-6) / 1e12
    out = {
        "metric": f"NFE/s (vector-field evals/s inside the adaptive Tsit5 NeuralODE forward, conv field {W}x{W}x8, B={B})",
        "value": nfe_total / el, "unit": "NFE/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dt, "data": "synthetic",
        "config": {"workload": f"{workload}: TDChain(Conv3x3(9=>64)+BN+gelu, Conv3x3(65=>64)+BN+gelu, Conv3x3(65=>8)) on "
                               f"{W}x{H}x8xB={B}, BatchNorm {'batch' if train else 'running'} statistics, Tsit5 adaptive "
                               f"abstol=reltol={tol:g}, tspan=(0,1), regularize=unbiased/error_estimate, forward pass",
                   "global_batch": B, "parallelism": "single GPU", "nfe_per_pass": nfe_total / steps,
                   "rk_steps_per_sec": steps_total / el, "fwd_ms_per_batch": el / steps * 1e3,
                   "fwd_plus_adjoint_ms_per_batch": fwd_adj_ms, "adjoint": bwd},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     # HBM/fabric bytes per f-eval from the committed PMC summary (profiles/r<N>/traffic.json, separate rocprofv3
                     # --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE over the five launches) or null: never a literal
                     "traffic": measured_traffic(f"conv_feval_{workload}_b{B}"), "kernel": "one f-eval = k_conv_wide(conv1) + k_bn_finalize + k_conv_wide(conv2) + "
                                                "k_bn_finalize + k_conv_out(conv3)",
                     "us_per_launch": us, "flop_per_launch": flop},
    }
    if brief:
        return {"workload": out["config"]["workload"], "value": out["value"], "unit": "NFE/s", "nfe_per_pass": nfe_total / steps,
                "fwd_ms_per_batch": el / steps * 1e3, "us_per_feval": us, "achieved_tflops": achieved, "roofline_frac": achieved / peak,
                "dtype": dt}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        cores = int(os.environ.get("LRNDE_CPU_CORES", min(len(os.sched_getaffinity(0)), 16)))
        cb = min(B, 16)  # bounded sample: f-evals of a 16-sample slice (batch statistics of the slice)
        fld = O.ConvField(W, H, 8, 64, params, act="gelu", bn_train=train, nthreads=cores, bf16=(dt == "bf16"))  # f32_split: the fp32 oracle
        tc = time.time()
        n = 0
        while time.time() - tc < 10.0:
            fld.rhs(xh[:cb].reshape(cb, -1), 0.3)
            n += 1
        cel = time.time() - tc
        out["cpu_baseline"] = {"value": n / cel * (cb / B), "unit": "NFE/s", "cores": cores, "kind": "port",
                               "sample": f"{n} f-evals of the C oracle (OpenMP, {cores} threads) on a {cb}-sample slice in "
                                         f"{cel:.1f} s, scaled by {cb}/{B} to whole-batch f-evals per second"}
    return out


def sde_layer_leg(h, ud, rng, B, D):
    """The NeuralDSDE LAYER as the reference runs it (src/layers/neural_sde.jl:50-123): adaptive solve on a Brownian path drawn up
    front, local step at (sol(t1), t1), and the pullback through the recorded accepted steps - BASELINE config 5's fwd + adjoint."""
    nfine = 256
    hh = np.float32(1.0 / nfine)
    Wp = np.concatenate([np.zeros((1, B, D), np.float32),
                         np.cumsum((rng.standard_normal((nfine, B, D)) * np.sqrt(hh)).astype(np.float32), axis=0, dtype=np.float32)], axis=0)
    Wd, zd = torch.from_numpy(Wp).cuda(), torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    du1 = torch.from_numpy(rng.standard_normal((1, B, D)).astype(np.float32)).cuda()
    tf, tb, att = [], [], 0
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fw = h.node_forward_record(ud, Wd, 0.0, 1.0, 0.14, 0.14, z_local=zd, mode="unbiased", t1_or_rand=0.3 + 0.03 * i, saveat=(), save_start=-1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        h.node_backward_recorded(du1.expand(fw["u"].shape[0], B, D).contiguous(), w_reg=2.0)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if i >= 2:
            tf.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
        att = fw["stats"]["naccept"] + fw["stats"]["nreject"]
    return {"what": "NeuralDSDE layer (state 32, hidden 64, abstol = reltol = 0.14), :unbiased, adaptive Euler-Heun on a 256-interval Brownian "
                    "path (lrnde_sde_node_forward_record: one launch for the whole solve, initial dts on the device) and the pullback through the "
                    "recorded steps incl. the regulariser's local step (lrnde_sde_node_backward_recorded: one launch for the whole sweep, the "
                    "parameter cotangent from its history records by an MFMA GEMM); medians of 12",
            "attempted_steps": att, "fwd_ms": float(np.median(tf)), "pullback_ms": float(np.median(tb)),
            "fwd_plus_adjoint_ms_per_batch": float(np.median(np.array(tf) + np.array(tb)))}


def sde_layer_side_measurement(B):
    """the same leg for the default bench line (BASELINE config 5 next to the headline, like the conv configs)"""
    import lrnde_amd as P
    from localregneuralde_jl_amd.layers import _mlp_desc
    D, H = 32, 64
    rng = np.random.default_rng(0)
    lim1, lim2 = np.sqrt(6.0 / (D + H)), np.sqrt(6.0 / (H + D))
    pd = np.concatenate([(rng.random(H * D, dtype=np.float32) * 2 - 1) * np.float32(lim1), np.zeros(H, np.float32),
                         (rng.random(D * H, dtype=np.float32) * 2 - 1) * np.float32(lim2), np.zeros(D, np.float32)]).astype(np.float32)
    pg = np.concatenate([(rng.random(D * D, dtype=np.float32) * 2 - 1) * np.float32(np.sqrt(6.0 / (2 * D))), np.zeros(D, np.float32)]).astype(np.float32)
    h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
    h.set_params(pd, pg)
    ud = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).cuda()
    return sde_layer_leg(h, ud, rng, B, D)


# ---- MNIST-SDE (BASELINE.json config 5): Euler-Heun steps with the local-regularisation residual, B=512 ----
def sde_main(args):
    """state 32, drift Dense(32=>64,tanh)->Dense(64=>32), diagonal diffusion Dense(32=>32), dW supplied.
    15.7 MFLOP and 65 KB of state per step: a latency measurement (us/step), not a roofline one."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("the SDE workload is a single-GPU latency measurement")
    torch.cuda.set_device(0)
    import lrnde_amd as P
    from localregneuralde_jl_amd.layers import _mlp_desc
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    D, H, B, nsteps = 32, 64, args.batch, 20
    rng = np.random.default_rng(0)
    lim1, lim2 = np.sqrt(6.0 / (D + H)), np.sqrt(6.0 / (H + D))
    pd = np.concatenate([(rng.random(H * D, dtype=np.float32) * 2 - 1) * np.float32(lim1), np.zeros(H, np.float32),
                         (rng.random(D * H, dtype=np.float32) * 2 - 1) * np.float32(lim2), np.zeros(D, np.float32)]).astype(np.float32)
    pg = np.concatenate([(rng.random(D * D, dtype=np.float32) * 2 - 1) * np.float32(np.sqrt(6.0 / (2 * D))), np.zeros(D, np.float32)]).astype(np.float32)
    u0 = rng.standard_normal((B, D)).astype(np.float32)
    dt = np.float32(1.0 / nsteps)
    dW = (rng.standard_normal((nsteps, B, D)) * np.sqrt(dt)).astype(np.float32)
    h = P.SdeHandle(_mlp_desc(P.Chain(P.Dense(D, H, "tanh"), P.Dense(H, D))))
    h.set_params(pd, pg)
    ud, dWd = torch.from_numpy(u0).cuda(), torch.from_numpy(dW).cuda()

    def one_pass():  # lrnde_sde_solve_fixed: the grid's steps in one launch + one for their records, one host sync per solve
        tr = h.solve_fixed(ud, dWd, 0.0, dt, 0.14, 0.14, 1.0 / 6.0)  # abstol=reltol=0.14 (mnist_sde/mlp.yml)
        return dict(eest=tr["eest"][-1], reg_val=tr["reg_val"][-1])

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_pass()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nst = args.steps * nsteps
    flop = 3 * (2 * B * 2 * D * H) + 3 * (2 * B * D * D)
    out = {
        "metric": f"Euler-Heun SDE steps/s with local regularisation (MNIST-SDE, B={B})", "value": nst / el, "unit": "steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"mnist_sde: NeuralDSDE drift Dense(32=>64,tanh)->Dense(64=>32), diffusion Dense(32=>32), Euler-Heun "
                               f"(src/perform_step.jl:172-206) on a fixed grid of {nsteps} steps with supplied dW, abstol=reltol=0.14, B={B}; "
                               "one pass = one solve (lrnde_sde_solve_fixed: ONE launch marches the grid's steps, a second forms their records); every step's "
                               "state, error estimate and residual are returned with the solve",
                   "global_batch": B, "parallelism": "single GPU", "us_per_sde_step": el / nst * 1e6, "flop_per_step": flop,
                   "last_eest": float(r["eest"]), "last_reg_val": float(r["reg_val"])},
        "roofline": {"bound": "mfma", "achieved": flop / (el / nst) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": flop / (el / nst) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "kernel": f"k_sde_eh_fast, march mode: the {nsteps} steps of the solve in ONE launch (weights resident in registers, no step waits for another "
                               "workgroup; k_sde_march_records forms the records); 15.7 MFLOP per step: latency bound by construction; wall clock per solve incl. the host's part",
                     "us_per_launch": el / args.steps * 1e6, "flop_per_launch": flop * nsteps},
    }
    out["config"]["layer"] = sde_layer_leg(h, ud, rng, B, D)
    if not args.no_cpu_baseline:
        import oracle as O
        cores = int(os.environ.get("LRNDE_CPU_CORES", min(len(os.sched_getaffinity(0)), 16)))
        p2 = np.concatenate([np.eye(D, dtype=np.float32).ravel(), np.zeros(D, np.float32), pg])
        drift = O.MlpField(D, H, pd, time_dep=False, act="tanh", nthreads=cores)
        diff = O.MlpField(D, D, p2, time_dep=False, act="identity", nthreads=cores)
        tc = time.time(); n = 0
        while time.time() - tc < 10.0:
            u = u0
            for i in range(nsteps):
                u = O.euler_heun_step(drift, diff, u, dW[i], float(i) * float(dt), dt, 0.14, 0.14, 1.0 / 6.0)["u"]
            n += nsteps
        cel = time.time() - tc
        out["cpu_baseline"] = {"value": n / cel, "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"{n} Euler-Heun steps of the C oracle (OpenMP, {cores} threads) at B={B} in {cel:.1f} s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
