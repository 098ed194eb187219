# LRNDELayer.jl — the drop-in: the reference's own `NeuralODE` layer call, routed through liblrnde.
#
# UNTESTED (no Julia in the build image).  `include("LRNDEBackend.jl"); include("LRNDELayer.jl")` inside
# `module LocalRegNeuralDE` (after src/layers/neural_ode.jl) adds, for models of the supported shape on ROCm arrays,
#   * `(n::NeuralODE)(x::ROCArray, ps, st) -> (sol, st)`   with the reference's return contract
#     (src/layers/neural_ode.jl:62-100): `sol.u::Vector`, `sol.t`, `sol.destats.nf / naccept / nreject`, `sol(t)` at the
#     saved knots, state `(; model, nfe, reg_val, rng, training)`;
#   * its `ChainRulesCore.rrule`, so that `Zygote.pullback` of the experiment losses (experiments/src/construct.jl:20-35,
#     experiments/src/utils.jl:104-115) runs the recorded forward + device-side continuous adjoint of the library —
#     cotangents may arrive for `sol.u[end]` only (`diffeqsol_to_array`) or for every saved state
#     (`diffeqsol_to_timeseries`, src/utils.jl:42-46); `reg_val`'s gradient goes to `ps` only (neural_ode.jl:40).
# Everything else (other solvers, other model shapes, CPU arrays) falls through to the reference's own methods.
#
# Supported models (solver `Tsit5()`; the MLP field also with `VCAB3()` / `VCABM3()`):
#   * `TDChain(Dense(D+1 => H, act), Dense(H+1 => D))` / `Chain(Dense(D => H, act), Dense(H => D))`, act ∈ (identity, tanh,
#     gelu) — the MNIST-ODE field of experiments/src/construct.jl:180-189;
#   * the CIFAR10 `node_core` of experiments/src/construct.jl:213-218, `TDChain(Chain(Conv((3,3), 9 => 64; pad=1,
#     use_bias=false), BatchNorm(64, gelu)), Chain(Conv((3,3), 65 => 64; ...), BatchNorm(64, gelu)), Conv((3,3), 65 => 8; ...))`
#     on W×H×8×B arrays — BatchNorm running statistics travel in `st.model` exactly as the reference's closure leaves them
#     (src/layers/neural_ode.jl:44-48);
#   * `NeuralDSDE(Chain(Dense(D => H, act), Dense(H => D)), Dense(D => D); solver = LambaEulerHeun())` — the adaptive
#     Euler-Heun solve on a Brownian path drawn from `st.rng` on a uniform grid (see lrnde.h, lrnde_sde_node_forward_record).
# The pullback closures check the handle's RECORD GENERATION (lrnde_record_generation) and re-run their forward when another
# forward of the same layer has replaced the record in between (an evaluation pass, a second pullback in flight).

import ChainRulesCore
const CRC_ = ChainRulesCore
using .LRNDEBackend: LRNDEBackend, SolveOpts, Stats, MODE, REG_TYPE

# ---- the solution object: the fields the reference reads (src/utils.jl:7-9,25-46; neural_ode.jl:34) ----
struct LRNDEDestats; nf::Int; naccept::Int; nreject::Int; end
struct LRNDESolution{A}
    u::Vector{A}
    t::Vector{Float32}
    destats::LRNDEDestats
    retcode::Symbol
end
(sol::LRNDESolution)(t) = sol.u[findfirst(==(Float32(t)), sol.t)]    # saveat-only solution: answers at its knots
Base.ndims(sol::LRNDESolution) = ndims(first(sol.u)) + 1
_get_destats(sol::LRNDESolution) = sol.destats.nf
_get_destats(sol::LRNDESolution, x::Symbol) = getproperty(sol.destats, x)
diffeqsol_to_array(sol::LRNDESolution) = sol.u[end]
diffeqsol_to_timeseries(sol::LRNDESolution) = diffeqsol_to_timeseries(Array, sol)
diffeqsol_to_timeseries(::Type{Array}, sol::LRNDESolution) = _cat(unsqueeze.(sol.u; dims=ndims(sol) - 1), Val(ndims(sol) - 1))
diffeqsol_to_timeseries(::Type{Tuple}, sol::LRNDESolution) = Tuple(sol.u)

# ---- which layers go through the library ----
const _LRNDE_ACT = IdDict{Any, Int32}(identity => Int32(0), tanh => Int32(1), NNlib.tanh_fast => Int32(1), NNlib.gelu => Int32(2))

"(D, H, time_dep, act) of a supported vector field, or nothing"
function lrnde_field_shape(model)
    layers = model isa TDChain ? values(model.layers) : (model isa Lux.Chain ? values(model.layers) : nothing)
    layers === nothing && return nothing
    (length(layers) == 2 && all(l -> l isa Lux.Dense, layers)) || return nothing
    l1, l2 = layers
    td = model isa TDChain
    D, H = l1.in_dims - td, l1.out_dims
    (l2.in_dims == H + td && l2.out_dims == D && l2.activation === identity && haskey(_LRNDE_ACT, l1.activation)) || return nothing
    (l1.use_bias && l2.use_bias) || return nothing
    return (D, H, td, _LRNDE_ACT[l1.activation])
end

const _lrnde_handles = IdDict{Any, Ptr{Cvoid}}()   # one handle per layer object (one task, one stream: SURVEY.md §8b)
# n.solver -> lrnde_set_solver's code: the three choices of experiments/src/construct.jl:154-164 (`_ode_solver`)
_lrnde_alg(solver) = solver isa Tsit5 ? Int32(0) : nameof(typeof(solver)) === :VCAB3 ? Int32(1) :
                     nameof(typeof(solver)) === :VCABM3 ? Int32(2) : nothing

function lrnde_handle(n::NeuralODE)
    get!(_lrnde_handles, n) do
        D, H, td, act = lrnde_field_shape(n.model)
        ctx = LRNDEBackend.create(D, H, td, act)
        LRNDEBackend.set_solver!(ctx, _lrnde_alg(n.solver))   # the global solve's method; the local step stays Tsit5 (:75, :93)
        ctx
    end
end

lrnde_supported(n::NeuralODE, x) = _lrnde_alg(n.solver) !== nothing && lrnde_field_shape(n.model) !== nothing &&
                                   nameof(typeof(x)) === :ROCArray && eltype(x) === Float32

_lrnde_opts(n::NeuralODE) = SolveOpts(Float32(get(n.kwargs, :abstol, 1f-6)), Float32(get(n.kwargs, :reltol, 1f-3)),
                                       Int32(n.maxiters), Int32(get(n.kwargs, :save_start, true)), Int32(0), Int32(0))
_lrnde_mode(::NeuralODE{R}, training::Val{T}) where {R, T} = T ? R : :none
_lrnde_regtype(::NeuralODE{R, RT}) where {R, RT} = RT

# one layer call = one recorded forward of the library.  Returns everything the layer returns plus what the pullback needs.
function lrnde_layer_forward(n::NeuralODE, x, ps, st)
    ctx = lrnde_handle(n)
    p = ComponentArrays.getdata(ps)
    LRNDEBackend.set_params!(ctx, p)
    t0, t2 = Float32.(n.tspan)
    mode = _lrnde_mode(n, st.training)
    rng = mode === :none ? st.rng : Lux.replicate(st.rng)
    r = mode === :none ? 0f0 : rand(rng, Float32)          # ONE uniform draw: t1 = r*(t2-t0)+t0 (:71), or the index
    t1_or_rand = mode === :unbiased ? r * (t2 - t0) + t0 : r  #   floor(r*m) into sol.t[1:end-1] (:92)
    saveat = Float32.(collect(get(n.kwargs, :saveat, Float32[])))
    cap = length(saveat) + 3 + (isempty(saveat) && mode === :biased ? min(n.maxiters, 510) : 0)
    useries = similar(x, size(x)..., cap); tseries = zeros(Float32, cap)
    ns = Ref{Int32}(); reg = Ref{Float32}(); nfe = Ref{Int32}(); stats = Stats(); t1u = Ref{Float32}()
    LRNDEBackend.check(ctx, ccall((:lrnde_node_forward_record_ts, LRNDEBackend.lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int32, Float32, Float32, Ref{SolveOpts}, Int32, Int32, Float32, Ptr{Float32}, Int32,
         Ptr{Float32}, Ptr{Float32}, Int32, Ptr{Int32}, Ptr{Float32}, Ptr{Int32}, Ref{Stats}, Ptr{Float32}),
        ctx, pointer(x), LRNDEBackend.nbatch(x), t0, t2, _lrnde_opts(n), MODE[mode], REG_TYPE[_lrnde_regtype(n)], t1_or_rand,
        saveat, Int32(length(saveat)), pointer(useries), tseries, Int32(cap), ns, reg, nfe, stats, t1u))
    k = Int(ns[])
    us = [copy(selectdim(useries, ndims(useries), i)) for i in 1:k]
    sol = LRNDESolution(us, tseries[1:k], LRNDEDestats(stats.nf, stats.naccept, stats.nreject), :Success)
    st_ = (; model=st.model, nfe=Int(nfe[]), reg_val=reg[], rng, st.training)
    return sol, st_, ctx
end

"the handle's record generation (0: no usable record)"
function lrnde_generation(ctx; f=:lrnde_record_generation)
    g = Ref{UInt64}(0)
    ccall(LRNDEBackend.sym(f), Cint, (Ptr{Cvoid}, Ptr{UInt64}), ctx, g)
    return g[]
end

function (n::NeuralODE)(x::AbstractArray{Float32}, ps, st::NamedTuple)
    lrnde_supported(n, x) || return n(x, ps, st, st.training)      # the reference's own path (neural_ode.jl:62)
    sol, st_, _ = lrnde_layer_forward(n, x, ps, st)
    return sol, st_
end

# Zygote.pullback through the layer call.  Cotangent shapes Zygote produces for `(sol, st_)`: a Tangent / NamedTuple with
# `u` (a vector of array cotangents, `nothing` for unused states) for sol, and `reg_val` for st_.
function CRC_.rrule(n::NeuralODE, x::AbstractArray{Float32}, ps, st::NamedTuple)
    lrnde_supported(n, x) || return CRC_.rrule_via_ad(Zygote.ZygoteRuleConfig(), (a, b, c) -> n(a, b, c, c.training), x, ps, st)
    sol, st_, ctx = lrnde_layer_forward(n, x, ps, st)
    gen = lrnde_generation(ctx)
    function lrnde_layer_pullback(Δ)
        # another forward of this layer ran since ours (or the record was consumed): same inputs, same draws -> same record
        lrnde_generation(ctx) == gen || lrnde_layer_forward(n, x, ps, st)
        Δsol, Δst = Δ
        k = length(sol.u)
        du = zeros(Float32, size(x)..., k) |> z -> copyto!(similar(x, size(z)...), z)
        Δu = Δsol === nothing || Δsol isa CRC_.AbstractZero ? nothing : Δsol.u
        if Δu !== nothing
            for i in 1:k
                (Δu[i] === nothing || Δu[i] isa CRC_.AbstractZero) && continue
                copyto!(selectdim(du, ndims(du), i), Δu[i])
            end
        end
        w_reg = (Δst === nothing || Δst isa CRC_.AbstractZero || Δst.reg_val === nothing) ? 0f0 : Float32(Δst.reg_val)
        dx = similar(x); dp = similar(ComponentArrays.getdata(ps)); sb = Stats()
        LRNDEBackend.check(ctx, ccall((:lrnde_node_backward_recorded_ts, LRNDEBackend.lib), Cint,
            (Ptr{Cvoid}, Int32, Ptr{Float32}, Int32, Float32, Ptr{Float32}, Ptr{Float32}, Ref{Stats}),
            ctx, LRNDEBackend.nbatch(x), pointer(du), Int32(k), w_reg, pointer(dx), pointer(dp), sb))
        return CRC_.NoTangent(), dx, ComponentArrays.ComponentArray(dp, ComponentArrays.getaxes(ps)), CRC_.NoTangent()
    end
    return (sol, st_), lrnde_layer_pullback
end

# =====================================================================================================================
# The conv vector field of experiments/src/construct.jl:213-218 behind the same `NeuralODE` call
# =====================================================================================================================
"(W, H, C, Hc, act) of a supported conv field, or nothing"
function lrnde_conv_field_shape(model, x)
    model isa TDChain || return nothing
    ls = values(model.layers)
    length(ls) == 3 || return nothing
    c1, c2, c3 = ls
    isblock(l) = l isa Lux.Chain && length(l.layers) == 2 && values(l.layers)[1] isa Lux.Conv && values(l.layers)[2] isa Lux.BatchNorm
    (isblock(c1) && isblock(c2) && c3 isa Lux.Conv) || return nothing
    k1, b1 = values(c1.layers); k2, b2 = values(c2.layers)
    C = size(x, 3); Hc = k1.out_chs
    ok(k, cin, cout) = k.kernel_size == (3, 3) && k.in_chs == cin && k.out_chs == cout && !k.use_bias && k.pad == (1, 1, 1, 1) &&
                       k.stride == (1, 1) && k.dilation == (1, 1)
    (ok(k1, C + 1, Hc) && ok(k2, Hc + 1, Hc) && ok(c3, Hc + 1, C)) || return nothing
    (b1.activation === b2.activation && haskey(_LRNDE_ACT, b1.activation) && b1.affine && b2.affine) || return nothing
    (C == 8 && Hc == 64 && size(x, 1) % 4 == 0 && 4 <= size(x, 1) <= 124 && size(x, 2) >= 2) || return nothing   # lrnde_conv_desc
    return (size(x, 1), size(x, 2), C, Hc, _LRNDE_ACT[b1.activation])
end

const _lrnde_conv_handles = IdDict{Any, Ptr{Cvoid}}()
function lrnde_conv_handle(n::NeuralODE, x)
    get!(_lrnde_conv_handles, n) do
        W, H, C, Hc, act = lrnde_conv_field_shape(n.model, x)
        LRNDEBackend.conv_create(W, H; channels=C, hidden=Hc, act, bn_train=true)
    end
end
lrnde_conv_supported(n::NeuralODE, x) = n.solver isa Tsit5 && ndims(x) == 4 && nameof(typeof(x)) === :ROCArray && eltype(x) === Float32 &&
                                        !haskey(n.kwargs, :saveat) && lrnde_conv_field_shape(n.model, x) !== nothing

# st.model of the TDChain: (layer_1 = (layer_1 = (;), layer_2 = (running_mean, running_var, training)), layer_2 = ..., layer_3 = (;))
_bn_states(stm) = (stm.layer_1.layer_2, stm.layer_2.layer_2)
function _bn_pack(stm, like)      # -> device vector [mean1; var1; mean2; var2]
    a, b = _bn_states(stm)
    v = similar(like, 4 * length(a.running_mean))
    copyto!(v, vcat(vec(a.running_mean), vec(a.running_var), vec(b.running_mean), vec(b.running_var)))
    return v
end
function _bn_unpack(stm, v)       # the running statistics after the solve, as the reference's `st_` carries them
    a, b = _bn_states(stm); h = length(a.running_mean)
    vv = Array(v)
    a2 = merge(a, (; running_mean=copyto!(similar(a.running_mean), vv[1:h]), running_var=copyto!(similar(a.running_var), vv[h+1:2h])))
    b2 = merge(b, (; running_mean=copyto!(similar(b.running_mean), vv[2h+1:3h]), running_var=copyto!(similar(b.running_var), vv[3h+1:4h])))
    return merge(stm, (; layer_1=merge(stm.layer_1, (; layer_2=a2)), layer_2=merge(stm.layer_2, (; layer_2=b2))))
end

function lrnde_conv_layer_forward(n::NeuralODE, x, ps, st; record::Bool)
    ctx = lrnde_conv_handle(n, x)
    LRNDEBackend.conv_set_params!(ctx, ComponentArrays.getdata(ps))   # flat order = Lux order (include/lrnde.h: conv1.weight, bn1.scale, bn1.bias, ...)
    mode = _lrnde_mode(n, st.training)
    training = st.training === Val(true)
    LRNDEBackend.conv_check(ctx, ccall((:lrnde_conv_set_bn_mode, LRNDEBackend.lib), Cint, (Ptr{Cvoid}, Int32), ctx, Int32(training)))
    LRNDEBackend.conv_set_bn_state!(ctx, _bn_pack(st.model, ComponentArrays.getdata(ps)))
    t0, t2 = Float32.(n.tspan)
    rng = mode === :none ? st.rng : Lux.replicate(st.rng)
    r = mode === :none ? 0f0 : rand(rng, Float32)
    t1_or_rand = mode === :unbiased ? r * (t2 - t0) + t0 : r
    u_end, reg, nfe, stats, t1 = LRNDEBackend.conv_node_forward(ctx, x, t0, t2, _lrnde_opts(n), mode, _lrnde_regtype(n), t1_or_rand; record)
    stm = training ? _bn_unpack(st.model, LRNDEBackend.conv_get_bn_state(ctx, _bn_pack(st.model, ComponentArrays.getdata(ps)))) : st.model
    # saveat = [t1, t2] / [t2] (neural_ode.jl:102-111, no user saveat on this path): the caller reads sol.u[end]
    sol = LRNDESolution([u_end], Float32[t2], LRNDEDestats(stats.nf, stats.naccept, stats.nreject), :Success)
    st_ = (; model=stm, nfe, reg_val=reg, rng, st.training)
    return sol, st_, ctx
end

function (n::NeuralODE)(x::AbstractArray{Float32, 4}, ps, st::NamedTuple)
    lrnde_conv_supported(n, x) || return n(x, ps, st, st.training)
    sol, st_, _ = lrnde_conv_layer_forward(n, x, ps, st; record=false)
    return sol, st_
end

function CRC_.rrule(n::NeuralODE, x::AbstractArray{Float32, 4}, ps, st::NamedTuple)
    lrnde_conv_supported(n, x) || return CRC_.rrule_via_ad(Zygote.ZygoteRuleConfig(), (a, b, c) -> n(a, b, c, c.training), x, ps, st)
    sol, st_, ctx = lrnde_conv_layer_forward(n, x, ps, st; record=true)
    gen = lrnde_generation(ctx; f=:lrnde_conv_record_generation)
    function lrnde_conv_layer_pullback(Δ)
        lrnde_generation(ctx; f=:lrnde_conv_record_generation) == gen || lrnde_conv_layer_forward(n, x, ps, st; record=true)
        Δsol, Δst = Δ
        ū = similar(x); fill!(ū, 0f0)
        if !(Δsol === nothing || Δsol isa CRC_.AbstractZero) && !(Δsol.u[end] === nothing || Δsol.u[end] isa CRC_.AbstractZero)
            copyto!(ū, Δsol.u[end])
        end
        w_reg = (Δst === nothing || Δst isa CRC_.AbstractZero || Δst.reg_val === nothing) ? 0f0 : Float32(Δst.reg_val)
        dx = similar(x); dp = similar(ComponentArrays.getdata(ps))
        LRNDEBackend.conv_node_backward_recorded(ctx, LRNDEBackend.nbatch(x), ū, w_reg, dx, dp)
        return CRC_.NoTangent(), dx, ComponentArrays.ComponentArray(dp, ComponentArrays.getaxes(ps)), CRC_.NoTangent()
    end
    return (sol, st_), lrnde_conv_layer_pullback
end

# =====================================================================================================================
# NeuralDSDE (src/layers/neural_sde.jl:74-123): the adaptive Euler-Heun solve + local step, and its pullback
# =====================================================================================================================
struct SdeAdaptOpts     # lrnde_sde_adapt_opts
    abstol::Float32; reltol::Float32; delta::Float32; dt0::Float32
    gamma::Float32; qmin::Float32; qmax::Float32; beta1::Float32; beta2::Float32; maxiters::Int32
end

"(D, H, act) of a supported drift / diffusion pair, or nothing"
function lrnde_sde_shape(n::NeuralDSDE)
    (n.drift isa Lux.Chain && n.diffusion isa Lux.Dense) || return nothing
    ls = values(n.drift.layers)
    (length(ls) == 2 && all(l -> l isa Lux.Dense, ls)) || return nothing
    l1, l2 = ls
    D, H = l1.in_dims, l1.out_dims
    (l2.in_dims == H && l2.out_dims == D && l2.activation === identity && haskey(_LRNDE_ACT, l1.activation) && l1.use_bias && l2.use_bias) || return nothing
    (n.diffusion.in_dims == D && n.diffusion.out_dims == D && n.diffusion.activation === identity) || return nothing
    return (D, H, _LRNDE_ACT[l1.activation], n.diffusion.use_bias)
end
const _lrnde_sde_handles = IdDict{Any, Ptr{Cvoid}}()
function lrnde_sde_handle(n::NeuralDSDE)
    get!(_lrnde_sde_handles, n) do
        D, H, act, gbias = lrnde_sde_shape(n)
        LRNDEBackend.sde_create(D, H, act; diffusion_bias=gbias)
    end
end
# LambaEulerHeun is the solver whose step the library integrates with (src/perform_step.jl:172-206); the reference's
# default SOSRI needs StochasticDiffEq's tableau and RSWM and stays on the reference's own path
lrnde_sde_supported(n::NeuralDSDE, x) = nameof(typeof(n.solver)) === :LambaEulerHeun && lrnde_sde_shape(n) !== nothing &&
                                         nameof(typeof(x)) === :ROCArray && eltype(x) === Float32 && ndims(x) == 2

const LRNDE_SDE_NFINE = Ref(256)    # grid intervals of the Brownian path drawn per layer call

function lrnde_sde_layer_forward(n::NeuralDSDE, x, ps, st)
    h = lrnde_sde_handle(n)
    LRNDEBackend.sde_set_params!(h, ComponentArrays.getdata(ps.drift), ComponentArrays.getdata(ps.diffusion))
    t0, t2 = Float32.(n.tspan)
    mode = st.training === Val(true) ? _sde_mode(n) : :none
    rng = Lux.replicate(st.rng)
    nfine = LRNDE_SDE_NFINE[]
    hh = (t2 - t0) / nfine
    # the Brownian path on the grid (W[0] = 0) and the local step's standard-normal draw, from the layer's own stream
    inc = randn(rng, Float32, size(x)..., nfine) .* sqrt(hh)
    Wh = cat(zeros(Float32, size(x)..., 1), cumsum(inc; dims=ndims(inc)); dims=ndims(inc))
    W = copyto!(similar(x, size(Wh)...), Wh)
    z = copyto!(similar(x), randn(rng, Float32, size(x)...))
    r = mode === :none ? 0f0 : rand(rng, Float32)
    t1_or_rand = mode === :unbiased ? r * (t2 - t0) + t0 : r
    saveat = Float32.(collect(get(n.kwargs, :saveat, Float32[])))
    opts = SdeAdaptOpts(Float32(get(n.kwargs, :abstol, 1f-2)), Float32(get(n.kwargs, :reltol, 1f-2)), Float32(1 / 6), 0f0,
                        0.9f0, 0.2f0, 1.125f0, 0.14f0, 0.08f0, Int32(n.maxiters))
    cap = length(saveat) + 3 + (isempty(saveat) && mode === :biased ? nfine + 1 : 0)
    useries = similar(x, size(x)..., cap); tseries = zeros(Float32, cap)
    ns = Ref{Int32}(); reg = Ref{Float32}(); nf = Ref{Int32}(); ng = Ref{Int32}(); stats = Stats(); t1u = Ref{Float32}()
    save_start = haskey(n.kwargs, :save_start) ? Int32(n.kwargs[:save_start]) : Int32(-1)
    LRNDEBackend.sde_check(h, ccall((:lrnde_sde_node_forward_record, LRNDEBackend.lib), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Int32, Float32, Float32, Ref{SdeAdaptOpts}, Int32, Float32, Ptr{Float32}, Int32,
         Ptr{Float32}, Int32, Ptr{Float32}, Ptr{Float32}, Int32, Ptr{Int32}, Ptr{Float32}, Ptr{Int32}, Ptr{Int32}, Ref{Stats}, Ptr{Float32}),
        h, pointer(x), pointer(W), Int32(nfine), LRNDEBackend.nbatch(x), t0, t2, opts, MODE[mode], t1_or_rand, pointer(z), save_start,
        saveat, Int32(length(saveat)), pointer(useries), tseries, Int32(cap), ns, reg, nf, ng, stats, t1u))
    k = Int(ns[])
    us = [copy(selectdim(useries, ndims(useries), i)) for i in 1:k]
    sol = LRNDESolution(us, tseries[1:k], LRNDEDestats(Int(nf[]), stats.naccept, stats.nreject), :Success)
    st_ = (; drift=st.drift, diffusion=st.diffusion, nfe_drift=Int(nf[]), nfe_diffusion=Int(ng[]), reg_val=reg[], rng, st.training)
    return sol, st_, h, W      # W must outlive the pullback: the record refers to it
end
_sde_mode(::NeuralDSDE{R}) where {R} = R

function (n::NeuralDSDE)(x::AbstractMatrix{Float32}, ps, st::NamedTuple)
    lrnde_sde_supported(n, x) || return n(x, ps, st, st.training)      # the reference's own path (neural_sde.jl:84)
    sol, st_, _, _ = lrnde_sde_layer_forward(n, x, ps, st)
    return sol, st_
end

function CRC_.rrule(n::NeuralDSDE, x::AbstractMatrix{Float32}, ps, st::NamedTuple)
    lrnde_sde_supported(n, x) || return CRC_.rrule_via_ad(Zygote.ZygoteRuleConfig(), (a, b, c) -> n(a, b, c, c.training), x, ps, st)
    sol, st_, h, W = lrnde_sde_layer_forward(n, x, ps, st)
    gen = lrnde_generation(h; f=:lrnde_sde_record_generation)
    function lrnde_sde_layer_pullback(Δ)
        if lrnde_generation(h; f=:lrnde_sde_record_generation) != gen
            _, _, _, W = lrnde_sde_layer_forward(n, x, ps, st)      # same st.rng -> same path, same record
        end
        Δsol, Δst = Δ
        k = length(sol.u)
        du = similar(x, size(x)..., k); fill!(du, 0f0)
        Δu = Δsol === nothing || Δsol isa CRC_.AbstractZero ? nothing : Δsol.u
        if Δu !== nothing
            for i in 1:k
                (Δu[i] === nothing || Δu[i] isa CRC_.AbstractZero) && continue
                copyto!(selectdim(du, ndims(du), i), Δu[i])
            end
        end
        w_reg = (Δst === nothing || Δst isa CRC_.AbstractZero || Δst.reg_val === nothing) ? 0f0 : Float32(Δst.reg_val)
        dx = similar(x); dpf = similar(ComponentArrays.getdata(ps.drift)); dpg = similar(ComponentArrays.getdata(ps.diffusion))
        GC.@preserve W LRNDEBackend.sde_check(h, ccall((:lrnde_sde_node_backward_recorded, LRNDEBackend.lib), Cint,
            (Ptr{Cvoid}, Int32, Ptr{Float32}, Int32, Float32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
            h, LRNDEBackend.nbatch(x), pointer(du), Int32(k), w_reg, pointer(dx), pointer(dpf), pointer(dpg)))
        dps = ComponentArrays.ComponentArray(vcat(dpf, dpg), ComponentArrays.getaxes(ps))   # reg_val: no cotangent for x (neural_sde.jl:42)
        return CRC_.NoTangent(), dx, dps, CRC_.NoTangent()
    end
    return (sol, st_), lrnde_sde_layer_pullback
end
