# make_golden.jl — reference-side golden vectors for the logpdf / posterior / VFE path.
#
#   julia --project=tests/golden -e 'using Pkg; Pkg.instantiate()'      (tests/golden/Project.toml: AbstractGPs 0.5.24)
#   julia --project=tests/golden tests/golden/make_golden.jl
#
# Reads tests/golden/julia_inputs/<case>.gpb (written by tests/golden/export_julia_inputs.py: the inputs of the committed
# fixtures, bit for bit) and runs the REAL AbstractGPs.jl on them:
#   logpdf(fx, y), logpdf(fx, Y)                         src/finite_gp_projection.jl:306-311
#   posterior(fx, y)  → data.α, data.C.U, data.δ         src/exact_gpr_posterior.jl:29-35
#   mean_and_var / cov of the posterior at xs            src/exact_gpr_posterior.jl:60-90
#   posterior(post(x₂, Σ₂), y₂)  (sequential)            src/exact_gpr_posterior.jl:46-56
#   posterior(VFE(fz), fx, y) → data.{m_ε, Λ_ε.U, U, α, b_y}, elbo, approx_log_evidence(DTC)
#                                                        src/sparse_approximations.jl:58-75, 248-254, 282-286
#   update_posterior (new observations / new pseudo-points)   src/sparse_approximations.jl:87-176
#   central differences of elbo along (variance, transform parameters, noise, pseudo-inputs): the pin of the accelerated path's ELBO gradient
# and writes tests/golden/julia/<case>.gpb.  tests/test_julia_golden.py compares the CPU oracle (-m "not gpu") and the HIP
# path (-m gpu) with those files when they are present and skips LOUDLY when they are not: Julia is not installed in the image
# this repository is built in, so a maintainer with Julia runs this once and commits tests/golden/julia/*.gpb.
#
# File format GPB1 (tests/golden/gpb.py): "GPB1" | UInt32 count | count × { UInt32 len | name | UInt32 ndim | Int64 dims | Float64 data },
# little-endian, column-major.

using AbstractGPs
using LinearAlgebra

const HERE = @__DIR__

function read_gpb(path)
    out = Dict{String,Any}()
    open(path, "r") do io
        magic = String(read(io, 4))
        magic == "GPB1" || error("$path: not a GPB1 file")
        count = ltoh(read(io, UInt32))
        for _ in 1:count
            len = ltoh(read(io, UInt32))
            name = String(read(io, len))
            nd = ltoh(read(io, UInt32))
            dims = ntuple(_ -> Int(ltoh(read(io, Int64))), nd)
            if nd == 0
                out[name] = ltoh(read(io, Float64))
            else
                a = Array{Float64}(undef, dims...)
                read!(io, a)
                out[name] = ltoh.(a)
            end
        end
    end
    return out
end

function write_gpb(path, pairs::Vector{Pair{String,Any}})
    open(path, "w") do io
        write(io, "GPB1")
        write(io, htol(UInt32(length(pairs))))
        for (name, v) in pairs
            write(io, htol(UInt32(sizeof(name))))
            write(io, name)
            if v isa Number
                write(io, htol(UInt32(0)))
                write(io, htol(Float64(v)))
            else
                a = Array{Float64}(v)
                write(io, htol(UInt32(ndims(a))))
                for d in size(a)
                    write(io, htol(Int64(d)))
                end
                write(io, htol.(a))
            end
        end
    end
end

# kernel descriptor of the fixtures (oracle/gp_oracle.py: SE = 0, MATERN12 = 1, MATERN32 = 2, MATERN52 = 3)
function build_kernel(kind, variance, scale)
    base = kind == 0 ? SqExponentialKernel() : kind == 1 ? Matern12Kernel() : kind == 2 ? Matern32Kernel() : Matern52Kernel()
    k = if scale isa Number
        isnan(scale) ? base : base ∘ ScaleTransform(scale)
    else
        base ∘ ARDTransform(vec(scale))
    end
    return variance == 1.0 ? k : variance * k
end

# points: a Vector for 1-D inputs, RowVecs of the n×d matrix otherwise (the layouts of src/finite_gp_projection.jl:33-37)
points(X) = ndims(X) == 1 ? Vector{Float64}(X) : RowVecs(Matrix{Float64}(X))
npoints(X) = size(X, 1)
rows(X, r) = ndims(X) == 1 ? X[r] : X[r, :]
noise(s2, r) = s2 isa Number ? s2 : Vector{Float64}(s2[r])

function run_case(inp)
    kind = Int(inp["kind"])
    k = build_kernel(kind, inp["variance"], inp["scale"])
    f = isnan(inp["mean"]) ? GP(k) : GP(inp["mean"], k)
    X, y, Y, Xs, Z = inp["x"], vec(inp["y"]), inp["Y"], inp["xs"], inp["z"]
    n, m = npoints(X), npoints(Z)
    s2 = inp["sigma2"]
    jitter = inp["jitter"]
    x, xs, z = points(X), points(Xs), points(Z)
    fx = f(x, noise(s2, 1:n))
    out = Pair{String,Any}[]

    # --- exact path
    push!(out, "logpdf" => logpdf(fx, y))
    push!(out, "logpdf_Y" => logpdf(fx, Matrix{Float64}(Y)))
    post = posterior(fx, y)
    push!(out, "alpha" => post.data.α)
    push!(out, "delta" => post.data.δ)
    push!(out, "U" => Matrix(post.data.C.U))
    pm, pv = mean_and_var(post, xs)
    push!(out, "post_mean" => pm)
    push!(out, "post_var" => pv)
    push!(out, "post_cov" => cov(post, xs))
    push!(out, "post_cross_cov" => cov(post, xs, x[1:min(n, 7)]))

    # --- sequential conditioning == batch (test/exact_gpr_posterior.jl:29-43)
    n1 = Int(inp["n1"])
    p1 = posterior(f(points(rows(X, 1:n1)), noise(s2, 1:n1)), y[1:n1])
    p2 = posterior(p1(points(rows(X, (n1 + 1):n)), noise(s2, (n1 + 1):n)), y[(n1 + 1):n])
    push!(out, "seq_alpha" => p2.data.α)
    push!(out, "seq_U" => Matrix(p2.data.C.U))

    # --- VFE / DTC
    fz = f(z, jitter)
    push!(out, "elbo" => elbo(VFE(fz), fx, y))
    push!(out, "dtc" => approx_log_evidence(DTC(fz), fx, y))
    ap = posterior(VFE(fz), fx, y)
    push!(out, "vfe_alpha" => ap.data.α)
    push!(out, "vfe_m_eps" => ap.data.m_ε)
    push!(out, "vfe_Lam_U" => Matrix(ap.data.Λ_ε.U))
    push!(out, "vfe_U" => Matrix(ap.data.U))
    push!(out, "vfe_b_y" => ap.data.b_y)
    vm, vv = mean_and_var(ap, xs)
    push!(out, "vfe_mean" => vm)
    push!(out, "vfe_var" => vv)
    push!(out, "vfe_cov" => cov(ap, xs))

    # --- derivatives of the ELBO along four directions by central differences of the reference's own `elbo` (t = ±1e-4): variance·(1 + t), every transform
    #     parameter·(1 + t), every noise variance·(1 + t), z + t·D with D_ij = sin(i + 3j) — what the accelerated path's gp_vfe_grad returns analytically,
    #     contracted with the same directions (examples/0-intro-1d/script.jl:385-394 differentiates this expression)
    function elbo_at(variance, scale, s2t, Zt)
        kt = build_kernel(kind, variance, scale)
        ft = isnan(inp["mean"]) ? GP(kt) : GP(inp["mean"], kt)
        return elbo(VFE(ft(points(Zt), jitter)), ft(x, noise(s2t, 1:n)), y)
    end
    t = 1e-4
    function central(g)
        return (g(t) - g(-t)) / (2t)
    end
    D = ndims(Z) == 1 ? [sin(i + 3.0) for i in 1:m] : [sin(i + 3.0 * j) for i in 1:m, j in 1:size(Z, 2)]
    sc = inp["scale"]
    has_scale = !(sc isa Number && isnan(sc))
    push!(out, "elbo_dir" => [central(u -> elbo_at(inp["variance"] * (1 + u), sc, s2, Z)),
                              has_scale ? central(u -> elbo_at(inp["variance"], sc .* (1 + u), s2, Z)) : 0.0,
                              central(u -> elbo_at(inp["variance"], sc, s2 .* (1 + u), Z)),
                              central(u -> elbo_at(inp["variance"], sc, s2, Z .+ u .* D))])

    # --- update_posterior: new observations on the same pseudo-points (src/sparse_approximations.jl:87-121)
    a1 = posterior(VFE(fz), f(points(rows(X, 1:n1)), noise(s2, 1:n1)), y[1:n1])
    a2 = update_posterior(a1, f(points(rows(X, (n1 + 1):n)), noise(s2, (n1 + 1):n)), y[(n1 + 1):n])
    push!(out, "upd_obs_alpha" => a2.data.α)
    push!(out, "upd_obs_m_eps" => a2.data.m_ε)
    um, uv = mean_and_var(a2, xs)
    push!(out, "upd_obs_mean" => um)
    push!(out, "upd_obs_var" => uv)

    # --- update_posterior: new pseudo-points appended (src/sparse_approximations.jl:130-176)
    m1 = Int(inp["m1"])
    b1 = posterior(VFE(f(points(rows(Z, 1:m1)), jitter)), fx, y)
    # The reference puts no jitter on the new diagonal block (:138), so this step throws PosDefException when K_zz is near-singular;
    # that outcome is recorded as NaN fields and the Python side requires the same outcome of the oracle and of the device.
    zfields = try
        b2 = update_posterior(b1, f(points(rows(Z, (m1 + 1):m)), jitter))
        zm, zv = mean_and_var(b2, xs)
        (b2.data.α, b2.data.m_ε, zm, zv)
    catch err
        err isa LinearAlgebra.PosDefException || rethrow()
        (fill(NaN, m), fill(NaN, m), fill(NaN, npoints(Xs)), fill(NaN, npoints(Xs)))
    end
    push!(out, "upd_z_alpha" => zfields[1])
    push!(out, "upd_z_m_eps" => zfields[2])
    push!(out, "upd_z_mean" => zfields[3])
    push!(out, "upd_z_var" => zfields[4])
    return out
end

function main()
    indir = joinpath(HERE, "julia_inputs")
    outdir = joinpath(HERE, "julia")
    mkpath(outdir)
    for fn in sort(filter(endswith(".gpb"), readdir(indir)))
        inp = read_gpb(joinpath(indir, fn))
        out = run_case(inp)
        write_gpb(joinpath(outdir, fn), out)
        println(fn, ": logpdf = ", first(v for (k, v) in out if k == "logpdf"), "  elbo = ", first(v for (k, v) in out if k == "elbo"))
    end
    println("AbstractGPs ", pkgversion(AbstractGPs), ", KernelFunctions ", pkgversion(AbstractGPs.KernelFunctions), ", Julia ", VERSION)
end

main()
