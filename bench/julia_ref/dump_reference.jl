# Dumps outputs of the reference (avik-pal/LocalRegNeuralDE.jl) on the committed golden inputs.
# Usage: julia --project=<reference checkout> bench/julia_ref/dump_reference.jl tests/golden
# UN-RUN in this repository's build image (no julia there); see README.md in this directory.
using LocalRegNeuralDE, Lux, ComponentArrays, OrdinaryDiffEq, SciMLSensitivity, Random, Statistics, NPZ, Pkg

golden = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "..", "tests", "golden")
g = npzread(joinpath(golden, "mnist_mlp_b16.npz"))

D, H = 784, 100
x = permutedims(Float32.(g["x"]))                 # numpy (B, D) row-major  ->  Julia D x B
B = size(x, 2)
tol = Float32(g["tol"])
t_step, dt_step = Float32(g["t"]), Float32(g["dt"])

# the MNIST-ODE vector field of experiments/src/construct.jl:180-189
field = TDChain(; d1=Dense(D + 1 => H, tanh), d2=Dense(H + 1 => D))
make_node(reg) = NeuralODE(field; solver=Tsit5(), reltol=tol, abstol=tol, save_start=false, regularize=reg,
                           regularize_type=:error_estimate, maxiters=10_000,
                           sensealg=InterpolatingAdjoint(; autojacvec=ZygoteVJP()))

rng = Xoshiro(0)
node = make_node(:none)
ps, st = Lux.setup(rng, node)
ps = ComponentArray(ps)
@assert length(getdata(ps)) == length(g["params"]) "flat parameter layout differs"
copyto!(getdata(ps), Float32.(vec(g["params"])))   # [vec(W1) (H x (D+1), column-major); b1; vec(W2); b2]

# 1. one vector-field evaluation dudt(x, p, t)  (src/layers/neural_ode.jl:45-48)
function dudt(u, t)
  y, _ = Lux.apply(field, LocalRegNeuralDE.ArrayAndTime(u, t), ps.model, st.model)
  return LocalRegNeuralDE.get_array(y)
end
k1 = dudt(x, t_step)

# 2. the adaptive solve, regularize = :none  (sol.u[end], destats)
sol, st_none = node(x, ps, st)
u_end_none = sol.u[end]

# 3. regularize = :unbiased: the layer draws t1 from Lux.replicate(st.rng) (neural_ode.jl:70-71); draw it the same way
node_u = make_node(:unbiased)
ps_u, st_u = Lux.setup(Xoshiro(0), node_u)
rng_t1 = Lux.replicate(st_u.rng)
t1 = rand(rng_t1, Float32) * (node_u.tspan[2] - node_u.tspan[1]) + node_u.tspan[1]
sol_u, st_u2 = node_u(x, ps, st_u)

# 4. one local step from (x, k1) at (t, dt): _perform_step on an integrator initialised at t with the same dt
prob = ODEProblem((u, p, t) -> dudt(u, t), x, (t_step, 1.0f0), ps)
integ = init(prob, Tsit5(); abstol=tol, reltol=tol, dt=dt_step, save_start=false)
cacheT = integ.cache
u_step, reg_step, nf_step, dt_used = LocalRegNeuralDE._perform_step(integ, cacheT, ps, Val(:error_estimate))

# 5. CPU timing of the reference's own path (median of repeats), for cpu_baseline.kind = "reference"
node(x, ps, st)
times = [(@elapsed node(x, ps, st)) for _ in 1:10]

npzwrite(joinpath(golden, "reference_mnist_mlp_b16.npz"),
         Dict("k1" => permutedims(k1), "u_end_none" => permutedims(u_end_none), "nf_none" => Int64(sol.destats.nf),
              "naccept_none" => Int64(sol.destats.naccept), "nreject_none" => Int64(sol.destats.nreject),
              "t1" => Float32(t1), "u_end_unbiased" => permutedims(sol_u.u[end]), "nfe_unbiased" => Int64(st_u2.nfe),
              "reg_val_unbiased" => Float32(st_u2.reg_val), "step_u" => permutedims(u_step), "step_reg" => Float32(reg_step),
              "step_dt" => Float32(dt_used), "solve_seconds_median" => Float64(median(times)), "threads" => Int64(Threads.nthreads())))
open(joinpath(golden, "reference_versions.txt"), "w") do io
  Pkg.status(; io=io)
end
println("wrote ", joinpath(golden, "reference_mnist_mlp_b16.npz"))
