# LRNDEBackend.jl — ccall binding of liblrnde (include/lrnde.h) for LocalRegNeuralDE.jl.
#
# UNTESTED: Julia is not installed in the build image of this repository.  The C ABI bound here is exercised
# by the Python mirror (localregneuralde.jl_amd/_lib.py, ctypes) and its GPU tests; struct layouts below are the
# ones of include/lrnde.h.  A maintainer would `include` this file from src/LocalRegNeuralDE.jl and route the
# Tsit5 / TDChain cases of NeuralODE (src/layers/neural_ode.jl) and NeuralDSDE (src/layers/neural_sde.jl)
# through it; see INTEGRATION.md for the call sites.  Arrays are device arrays whose `pointer` is a raw device
# pointer (AMDGPU.ROCArray); states are the reference's own column-major arrays (D×B, W×H×C×B), parameters
# the flat ComponentArray data vector (`getdata(ps)`).
module LRNDEBackend

import ChainRulesCore
import Libdl
using ChainRulesCore: NoTangent

const lib = get(ENV, "LRNDE_LIB", "liblrnde.so")
# entry points chosen at run time (a Symbol in a variable): `ccall((name, lib), ...)` needs a constant tuple, so these go
# through dlsym
sym(name::Symbol) = Libdl.dlsym(Libdl.dlopen(lib), name)

# ---- structs of include/lrnde.h ----
struct ModelDesc; state_dim::Int32; hidden_dim::Int32; time_dep::Int32; act::Int32; end
struct SolveOpts
    abstol::Float32; reltol::Float32; maxiters::Int32; save_start::Int32; save_everystep::Int32; exact_pow::Int32
end
mutable struct Stats
    retcode::Int32; nf::Int32; naccept::Int32; nreject::Int32; iters::Int32; nsaved::Int32
    t_final::Float32; dt_final::Float32; eest_last::Float32; dt_init::Float32
    Stats() = new()
end
struct ConvDesc
    width::Int32; height::Int32; channels::Int32; hidden::Int32; act::Int32; bn_train::Int32; compute_dtype::Int32
    bn_eps::Float32
end
struct SriTableau   # FourStageSRIConstantCache fields in the order src/perform_step.jl:51-55 unpacks them
    a021::Float32; a031::Float32; a032::Float32; a041::Float32; a042::Float32; a043::Float32
    a121::Float32; a131::Float32; a132::Float32; a141::Float32; a142::Float32; a143::Float32
    b021::Float32; b031::Float32; b032::Float32; b041::Float32; b042::Float32; b043::Float32
    b121::Float32; b131::Float32; b132::Float32; b141::Float32; b142::Float32; b143::Float32
    c02::Float32; c03::Float32; c04::Float32; c11::Float32; c12::Float32; c13::Float32; c14::Float32
    alpha1::Float32; alpha2::Float32; alpha3::Float32; alpha4::Float32
    beta11::Float32; beta12::Float32; beta13::Float32; beta14::Float32
    beta21::Float32; beta22::Float32; beta23::Float32; beta24::Float32
    beta31::Float32; beta32::Float32; beta33::Float32; beta34::Float32
    beta41::Float32; beta42::Float32; beta43::Float32; beta44::Float32
end
SriTableau(cache) = SriTableau((Float32(getfield(cache, f) isa Number ? getfield(cache, f) : 0) for f in
                                (:a021, :a031, :a032, :a041, :a042, :a043, :a121, :a131, :a132, :a141, :a142, :a143,
                                 :b021, :b031, :b032, :b041, :b042, :b043, :b121, :b131, :b132, :b141, :b142, :b143,
                                 :c02, :c03, :c04, :c11, :c12, :c13, :c14, :α1, :α2, :α3, :α4,
                                 :beta11, :beta12, :beta13, :beta14, :beta21, :beta22, :beta23, :beta24,
                                 :beta31, :beta32, :beta33, :beta34, :beta41, :beta42, :beta43, :beta44))...)

const ACT_IDENTITY, ACT_TANH, ACT_GELU = Int32(0), Int32(1), Int32(2)
const MODE = Dict(:none => Int32(0), :unbiased => Int32(1), :biased => Int32(2))
const REG_TYPE = Dict(:error_estimate => Int32(0), :stiffness_estimate => Int32(1))
const DTYPE = Dict(:f32 => Int32(0), :bf16 => Int32(1), :f32_split => Int32(2))

function check(ctx, rc; last_error=:lrnde_last_error)
    rc == 0 && return nothing
    msg = unsafe_string(ccall(sym(last_error), Cstring, (Ptr{Cvoid},), ctx))
    rc == 4 ? throw(ArgumentError(msg)) : error("lrnde status $rc: $msg")
end
nbatch(x) = Int32(size(x, ndims(x)))

# ---- MLP field (experiments/src/construct.jl:180-189): handle, parameters ----
function create(D, H, time_dep::Bool, act::Int32; device=0, stream=C_NULL)
    ctx = Ref{Ptr{Cvoid}}()
    rc = ccall((:lrnde_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Ref{ModelDesc}, Cint, Ptr{Cvoid}),
               ctx, ModelDesc(D, H, time_dep, act), device, stream)
    rc == 0 || error("lrnde_create: status $rc")
    return ctx[]
end
destroy(ctx) = ccall((:lrnde_destroy, lib), Cint, (Ptr{Cvoid},), ctx)
set_params!(ctx, ps) = check(ctx, ccall((:lrnde_set_params, lib), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t),
                                        ctx, pointer(ps), length(ps)))

# n.solver of the layer's global solve: 0 Tsit5, 1 VCAB3, 2 VCABM3 (experiments/src/construct.jl:154-164)
set_solver!(ctx, alg) = check(ctx, ccall((:lrnde_set_solver, lib), Cint, (Ptr{Cvoid}, Int32), ctx, alg))

# dudt(u, p, t) — src/layers/neural_ode.jl:44-48
function rhs(ctx, u, t)
    du = similar(u)
    check(ctx, ccall((:lrnde_rhs, lib), Cint, (Ptr{Cvoid}, Ptr{Float32}, Float32, Int32, Ptr{Float32}),
                     ctx, pointer(u), t, nbatch(u), pointer(du)))
    return du
end

# _perform_step(integrator, cache::Tsit5ConstantCache, p, Val(reg_type)) — src/perform_step.jl:3-47
function perform_step(ctx, uprev, k1, t, dt, abstol, reltol)
    u = similar(uprev); k7 = similar(uprev)
    ee = Ref{Float32}(); re = Ref{Float32}(); rs = Ref{Float32}()
    check(ctx, ccall((:lrnde_perform_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32,
         Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        ctx, pointer(uprev), pointer(k1), nbatch(uprev), t, dt, abstol, reltol, pointer(u), pointer(k7), ee, re, rs))
    return u, k7, ee[], re[], rs[]
end

# the reference's own return tuple, src/perform_step.jl:31: (u, reg_val, 6 + integrator.sol.destats.nf, dt) — the step
# costs six f-evals (k2..k7; k1 is fsalfirst) and leaves integrator.dt as it was
function _perform_step(ctx, uprev, k1, t, dt, abstol, reltol, ::Val{RT}, nf_sol::Integer) where {RT}
    u, _, _, re, rs = perform_step(ctx, uprev, k1, t, dt, abstol, reltol)
    return u, (RT === :stiffness_estimate ? rs : re), 6 + nf_sol, dt
end

# (n::NeuralODE{R,RT})(x, ps, st) — src/layers/neural_ode.jl:56-100; t1_or_rand: t1 for :unbiased, rand for :biased
function node_forward(ctx, x, t0, t2, opts::SolveOpts, mode::Symbol, reg_type::Symbol, t1_or_rand)
    u_end = similar(x); reg = Ref{Float32}(); nfe = Ref{Int32}(); st = Stats(); t1 = Ref{Float32}()
    check(ctx, ccall((:lrnde_node_forward, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int32, Float32, Float32, Ref{SolveOpts}, Int32, Int32, Float32,
         Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Ref{Stats}, Ptr{Float32}),
        ctx, pointer(x), nbatch(x), t0, t2, opts, MODE[mode], REG_TYPE[reg_type], t1_or_rand,
        pointer(u_end), reg, nfe, st, t1))
    return u_end, reg[], Int(nfe[]), st, t1[]
end

# loss terms of one layer call and their pullback (the reference differentiates `solve` with
# InterpolatingAdjoint and `_perform_step` w.r.t. p only: src/layers/neural_ode.jl:40,118, src/utils.jl:60-62)
node_loss_terms(ctx, x, ps, t0, t2, opts, mode, reg_type, t1) =
    (set_params!(ctx, ps); node_forward(ctx, x, t0, t2, opts, mode, reg_type, t1)[1:3])

function ChainRulesCore.rrule(::typeof(node_loss_terms), ctx, x, ps, t0, t2, opts, mode, reg_type, t1)
    set_params!(ctx, ps)
    u_end = similar(x); reg = Ref{Float32}(); nfe = Ref{Int32}(); st = Stats(); t1u = Ref{Float32}()
    check(ctx, ccall((:lrnde_node_forward_record, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int32, Float32, Float32, Ref{SolveOpts}, Int32, Int32, Float32,
         Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Ref{Stats}, Ptr{Float32}),
        ctx, pointer(x), nbatch(x), t0, t2, opts, MODE[mode], REG_TYPE[reg_type], t1, pointer(u_end), reg, nfe, st, t1u))
    function pullback((ū, r̄, _))                     # cotangents of (sol.u[end], reg_val, nfe)
        dx = similar(x); dp = similar(ps); sb = Stats()
        check(ctx, ccall((:lrnde_node_backward_recorded, lib), Cint,
            (Ptr{Cvoid}, Int32, Ptr{Float32}, Float32, Ptr{Float32}, Ptr{Float32}, Ref{Stats}),
            ctx, nbatch(x), pointer(ū), Float32(r̄), pointer(dx), pointer(dp), sb))
        return NoTangent(), NoTangent(), dx, dp, ntuple(_ -> NoTangent(), 6)...
    end
    return (u_end, reg[], Int(nfe[])), pullback
end

# ---- conv field (experiments/src/construct.jl:213-218) ----
function conv_create(W, H; channels=8, hidden=64, act=ACT_GELU, bn_train=true, compute_dtype=:f32, bn_eps=1f-5, device=0,
                     stream=C_NULL)
    ctx = Ref{Ptr{Cvoid}}()
    rc = ccall((:lrnde_conv_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Ref{ConvDesc}, Cint, Ptr{Cvoid}),
               ctx, ConvDesc(W, H, channels, hidden, act, bn_train, DTYPE[compute_dtype], bn_eps), device, stream)
    rc == 0 || error("lrnde_conv_create: status $rc")
    return ctx[]
end
conv_check(ctx, rc) = check(ctx, rc; last_error=:lrnde_conv_last_error)
conv_set_params!(ctx, ps) = conv_check(ctx, ccall((:lrnde_conv_set_params, lib), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t),
                                                  ctx, pointer(ps), length(ps)))
conv_set_bn_state!(ctx, st) = conv_check(ctx, ccall((:lrnde_conv_set_bn_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t),
                                                    ctx, pointer(st), length(st)))
function conv_get_bn_state(ctx, like)   # like: a device vector of 4*hidden Float32 ([μ1; σ²1; μ2; σ²2] of st.model)
    conv_check(ctx, ccall((:lrnde_conv_get_bn_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), ctx, pointer(like), length(like)))
    return like
end
function conv_node_forward(ctx, x, t0, t2, opts::SolveOpts, mode::Symbol, reg_type::Symbol, t1_or_rand; record=false)
    u_end = similar(x); reg = Ref{Float32}(); nfe = Ref{Int32}(); st = Stats(); t1 = Ref{Float32}()
    f = record ? :lrnde_conv_node_forward_record : :lrnde_conv_node_forward
    conv_check(ctx, ccall(sym(f), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int32, Float32, Float32, Ref{SolveOpts}, Int32, Int32, Float32,
         Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Ref{Stats}, Ptr{Float32}),
        ctx, pointer(x), nbatch(x), t0, t2, opts, MODE[mode], REG_TYPE[reg_type], t1_or_rand, pointer(u_end), reg, nfe, st, t1))
    return u_end, reg[], Int(nfe[]), st, t1[]
end
function conv_node_backward_recorded(ctx, B, ū, w_reg, dx, dp)
    sb = Stats()
    conv_check(ctx, ccall((:lrnde_conv_node_backward_recorded, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Float32}, Float32, Ptr{Float32}, Ptr{Float32}, Ref{Stats}),
        ctx, Int32(B), pointer(ū), Float32(w_reg), pointer(dx), pointer(dp), sb))
    return dx, dp, sb
end

# ---- NeuralDSDE steps (src/perform_step.jl:49-106, 108-170, 172-206), diagonal noise ----
function sde_create(D, H, act::Int32; diffusion_bias=true, device=0, stream=C_NULL)
    h = Ref{Ptr{Cvoid}}()
    rc = ccall((:lrnde_sde_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Ref{ModelDesc}, Int32, Cint, Ptr{Cvoid}),
               h, ModelDesc(D, H, false, act), diffusion_bias, device, stream)
    rc == 0 || error("lrnde_sde_create: status $rc")
    return h[]
end
sde_check(h, rc) = check(h, rc; last_error=:lrnde_sde_last_error)
sde_set_params!(h, pd, pg) = sde_check(h, ccall((:lrnde_sde_set_params, lib), Cint,
    (Ptr{Cvoid}, Ptr{Float32}, Csize_t, Ptr{Float32}, Csize_t), h, pointer(pd), length(pd), pointer(pg), length(pg)))
function euler_heun_step(h, uprev, dW, t, dt, abstol, reltol, delta)
    u = similar(uprev); ee = Ref{Float32}(); rv = Ref{Float32}()
    sde_check(h, ccall((:lrnde_sde_euler_heun_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, pointer(uprev), pointer(dW), nbatch(uprev), t, dt, abstol, reltol, delta, pointer(u), ee, rv))
    return u, rv[], 0, dt               # (u, EEst*dt, 0, dt) as src/perform_step.jl:205
end
function sri_step(h, tab::SriTableau, uprev, dW, dZ, t, dt, abstol, reltol, delta)
    u = similar(uprev); ee = Ref{Float32}(); rv = Ref{Float32}()
    sde_check(h, ccall((:lrnde_sde_sri_step, lib), Cint,
        (Ptr{Cvoid}, Ref{SriTableau}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32, Float32,
         Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, tab, pointer(uprev), pointer(dW), pointer(dZ), nbatch(uprev), t, dt, abstol, reltol, delta, pointer(u), ee, rv))
    return u, rv[], 0, dt               # src/perform_step.jl:105
end

# gradient path of NeuralDSDE (TrackerAdjoint in the reference, src/layers/neural_sde.jl:12): reverse sweep of the
# fixed-grid Euler-Heun solve, and d(EEst*dt)/d(p_drift, p_diffusion) of the local step (uprev, dW, dt constant)
function sde_solve_fixed_backward(h, u0, u_traj, dW, t0, dt, nsteps, du_end, dp_drift, dp_diff)
    dx = similar(u0)
    sde_check(h, ccall((:lrnde_sde_solve_fixed_backward, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, pointer(u0), pointer(u_traj), pointer(dW), nbatch(u0), t0, dt, nsteps, pointer(du_end), pointer(dx), pointer(dp_drift), pointer(dp_diff)))
    return dx, dp_drift, dp_diff
end
function sde_euler_heun_reg_grad(h, uprev, dW, t, dt, abstol, reltol, delta, dp_drift, dp_diff)
    rv = Ref{Float32}()
    sde_check(h, ccall((:lrnde_sde_euler_heun_reg_grad, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, pointer(uprev), pointer(dW), nbatch(uprev), t, dt, abstol, reltol, delta, pointer(dp_drift), pointer(dp_diff), rv))
    return dp_drift, dp_diff, rv[]
end

# ---- multi-GPU: one process per GPU, batch sharded (not in the reference) ----
comm_unique_id() = (id = zeros(UInt8, 128); ccall((:lrnde_comm_unique_id, lib), Cint, (Ptr{UInt8},), id) == 0 || error("unique id"); id)
comm_init!(ctx, id::Vector{UInt8}, rank, nranks) = check(ctx, ccall((:lrnde_comm_init, lib), Cint,
    (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx, id, rank, nranks))

# the same for the Milstein step (solver = RKMilCommute(), src/perform_step.jl:108-170): pullback of the fixed-grid solve and
# d(EEst*dt)/dp of one local step
function sde_solve_fixed_backward_rkmil(h, u0, utraj, dW, t0, dt, nsteps, du_end, npd, npg)
    dx = similar(u0); dpd = similar(u0, npd); dpg = similar(u0, npg)
    sde_check(h, ccall((:lrnde_sde_solve_fixed_backward_rkmil, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, pointer(u0), pointer(utraj), pointer(dW), nbatch(u0), t0, dt, Int32(nsteps), pointer(du_end), pointer(dx), pointer(dpd), pointer(dpg)))
    return dx, dpd, dpg
end
function sde_rkmil_reg_grad(h, uprev, dW, t, dt, abstol, reltol, npd, npg)
    dpd = similar(uprev, npd); dpg = similar(uprev, npg); rv = Ref{Float32}()
    sde_check(h, ccall((:lrnde_sde_rkmil_reg_grad, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, pointer(uprev), pointer(dW), nbatch(uprev), t, dt, abstol, reltol, pointer(dpd), pointer(dpg), rv))
    return dpd, dpg, rv[]
end
# reverse sweep of ONE four-stage SRI step (src/perform_step.jl:49-106; SOSRI's coefficients are the caller's, as for sde_sri_step):
# loss = <du_new, u'> + w_reg * EEst*dt; dp_drift / dp_diff are ADDED to (zero them first)
function sde_sri_step_backward!(h, tab::SriTableau, uprev, dW, dZ, t, dt, abstol, reltol, delta, du_new, w_reg, dpd, dpg)
    dx = similar(uprev); rv = Ref{Float32}()
    sde_check(h, ccall((:lrnde_sde_sri_step_backward, lib), Cint,
        (Ptr{Cvoid}, Ref{SriTableau}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Float32, Float32, Float32, Float32, Float32,
         Ptr{Float32}, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        h, tab, pointer(uprev), pointer(dW), pointer(dZ), nbatch(uprev), t, dt, abstol, reltol, delta,
        pointer(du_new), w_reg, pointer(dx), pointer(dpd), pointer(dpg), rv))
    return dx, rv[]
end

end # module
