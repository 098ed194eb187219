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
# Supported model: `TDChain(Dense(D+1 => H, act), Dense(H+1 => D))` / `Chain(Dense(D => H, act), Dense(H => D))`,
# act ∈ (identity, tanh, gelu), solver `Tsit5()` — the MNIST-ODE field of experiments/src/construct.jl:180-189.

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
function lrnde_handle(n::NeuralODE)
    get!(_lrnde_handles, n) do
        D, H, td, act = lrnde_field_shape(n.model)
        LRNDEBackend.create(D, H, td, act)
    end
end

lrnde_supported(n::NeuralODE, x) = n.solver isa Tsit5 && lrnde_field_shape(n.model) !== nothing &&
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
    function lrnde_layer_pullback(Δ)
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
