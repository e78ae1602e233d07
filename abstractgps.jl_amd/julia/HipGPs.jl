# HipGPs.jl — Julia host shim: keeps the AbstractGP / FiniteGP / PosteriorGP surface of AbstractGPs.jl
# and routes the logpdf / posterior hot path through `ccall` into libgpmi355.so (include/gpmi355.h).
#
# NOT EXECUTED in the build container (Julia is not installed there — SURVEY.md §0 F3); it is the
# reference-side binding a maintainer adds (INTEGRATION.md).  abstractgps.jl_amd/api.py is its
# line-for-line ctypes mirror and is what tests/ run.  Reference plug-in point: "subtype AbstractGP
# and implement the FiniteGP primary API" (docs/src/api.md:18-30, 49-73).
#
#   f   = HipGP(GP(SqExponentialKernel()))          # wraps a stock GP              src/base_gp.jl:57-64
#   fx  = f(x, 0.01)                                # stock FiniteGP ctor           src/finite_gp_projection.jl:13-37
#   logpdf(fx, y)                                   # -> gp_logpdf                  src/finite_gp_projection.jl:306-311
#   p   = posterior(fx, y)                          # -> gp_posterior_fit           src/exact_gpr_posterior.jl:29-35
#   mean_and_var(p(xs)); cov(p(xs))                 # -> gp_posterior_predict       src/exact_gpr_posterior.jl:60-90
#   posterior(VFE(f(z, 1e-6)), fx, y); elbo(...)    # -> gp_vfe_fit / gp_vfe_predict src/sparse_approximations.jl:58-75,248-254
module HipGPs

using AbstractGPs
using AbstractGPs: AbstractGP, FiniteGP, GP, ZeroMean, ConstMean, CustomMean, VFE, DTC, mean_vector
using KernelFunctions
using KernelFunctions: SqExponentialKernel, Matern12Kernel, ExponentialKernel, Matern32Kernel, Matern52Kernel,
    TransformedKernel, ScaledKernel, ScaleTransform, ARDTransform, ColVecs, RowVecs
using LinearAlgebra, FillArrays, Statistics, StatsBase, Distributions, Random

export HipGP, HipPosteriorGP, HipApproxPosteriorGP, HipContext

const libgpmi355 = get(ENV, "GPMI355_LIB", joinpath(@__DIR__, "..", "csrc", "libgpmi355.so"))

# ---- C structs (include/gpmi355.h) ---------------------------------------------------------------
struct CKernel          # gp_kernel
    kind::Int32
    dtype::Int32
    variance::Float64
    nscale::Int32
    scale::Ptr{Float64}
end
struct CPoints          # gp_points
    data::Ptr{Cvoid}
    n::Int64
    d::Int32
    layout::Int32
end
struct CNoise           # gp_noise
    kind::Int32
    s::Float64
    diag::Ptr{Cvoid}
end

struct GpmiError <: Exception
    status::Int32
    msg::String
end

# status convention of the ABI: 0 ok; k>0 LAPACK info -> PosDefException(k) exactly like `cholesky`
# at src/finite_gp_projection.jl:308; <0 argument / HIP error with text in gp_last_error().
function check(rc::Int32)
    rc == 0 && return nothing
    rc > 0 && throw(LinearAlgebra.PosDefException(rc))
    msg = unsafe_string(ccall((:gp_last_error, libgpmi355), Cstring, ()))
    rc > -1000 ? throw(ArgumentError(msg)) : throw(GpmiError(rc, msg))
end

# ---- context ---------------------------------------------------------------------------------------
mutable struct HipContext
    handle::Ptr{Cvoid}
    function HipContext(device::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:gp_ctx_create, libgpmi355), Int32, (Ref{Ptr{Cvoid}}, Int32, Ptr{Cvoid}), h, device, C_NULL))
        c = new(h[])
        finalizer(c -> ccall((:gp_ctx_destroy, libgpmi355), Int32, (Ptr{Cvoid},), c.handle), c)
        return c
    end
end
const _default_ctx = Ref{Union{Nothing,HipContext}}(nothing)
default_context() = something(_default_ctx[], (_default_ctx[] = HipContext(0)))

# ---- the GP wrapper -------------------------------------------------------------------------------
struct HipGP{Tg<:GP} <: AbstractGP
    gp::Tg
    ctx::HipContext
end
HipGP(gp::GP) = HipGP(gp, default_context())

# internal AbstractGP API delegates to the wrapped GP (src/base_gp.jl:68-74) so that everything that is not
# accelerated (rand, dense Σy, composite kernels) keeps working through the stock methods.
Statistics.mean(f::HipGP, x::AbstractVector) = mean(f.gp, x)
Statistics.cov(f::HipGP, x::AbstractVector) = cov(f.gp, x)
Statistics.var(f::HipGP, x::AbstractVector) = var(f.gp, x)
Statistics.cov(f::HipGP, x::AbstractVector, y::AbstractVector) = cov(f.gp, x, y)

# ---- kernel -> descriptor (dispatch; anything else falls back) ------------------------------------
kind_of(::SqExponentialKernel) = Int32(0)
kind_of(::ExponentialKernel) = Int32(1)       # Matern12Kernel is an alias
kind_of(::Matern32Kernel) = Int32(2)
kind_of(::Matern52Kernel) = Int32(3)
kind_of(::Any) = nothing

# returns (kind, variance, scales::Vector{Float64}) or nothing if the kernel is not accelerated
descriptor(k) = (kd = kind_of(k); kd === nothing ? nothing : (kd, 1.0, Float64[]))
function descriptor(k::ScaledKernel)
    d = descriptor(k.kernel)
    return d === nothing ? nothing : (d[1], d[2] * only(k.σ²), d[3])
end
function descriptor(k::TransformedKernel{<:Any,<:ScaleTransform})
    d = descriptor(k.kernel)
    (d === nothing || !isempty(d[3])) && return nothing
    return (d[1], d[2], Float64[only(k.transform.s)])
end
function descriptor(k::TransformedKernel{<:Any,<:ARDTransform})
    d = descriptor(k.kernel)
    (d === nothing || !isempty(d[3])) && return nothing
    return (d[1], d[2], Vector{Float64}(k.transform.v))
end

# ---- input / noise marshalling ---------------------------------------------------------------------
# layout 0 Vector{T}; 1 ColVecs (D×N column-major); 2 RowVecs (N×D column-major)   src/finite_gp_projection.jl:32-37
points(x::Vector{T}) where {T<:Union{Float32,Float64}} = (x, CPoints(pointer(x), length(x), 1, 0), T)
function points(x::ColVecs{T,<:Matrix{T}}) where {T<:Union{Float32,Float64}}
    return (x.X, CPoints(pointer(x.X), size(x.X, 2), size(x.X, 1), 1), T)
end
function points(x::RowVecs{T,<:Matrix{T}}) where {T<:Union{Float32,Float64}}
    return (x.X, CPoints(pointer(x.X), size(x.X, 1), size(x.X, 2), 2), T)
end
points(::Any) = nothing

noise(Σ::Fill, ::Type{T}) where {T} = (nothing, CNoise(0, Float64(FillArrays.getindex_value(Σ)), C_NULL))
function noise(Σ::Diagonal{<:Any,<:Fill}, ::Type{T}) where {T}
    return (nothing, CNoise(0, Float64(FillArrays.getindex_value(Σ.diag)), C_NULL))
end
function noise(Σ::Diagonal, ::Type{T}) where {T}
    v = Vector{T}(Σ.diag)
    return (v, CNoise(1, 0.0, pointer(v)))
end
noise(::Any, ::Type) = nothing   # dense Σy: not accelerated

prior_mean(f::GP{<:ZeroMean}, x, ::Type{T}) where {T} = nothing
prior_mean(f::GP, x, ::Type{T}) where {T} = Vector{T}(mean_vector(f.mean, x))

# everything one call needs, or `nothing` => use the stock AbstractGPs path
function marshal(fx::FiniteGP{<:HipGP})
    desc = descriptor(fx.f.gp.kernel)
    px = points(fx.x)
    (desc === nothing || px === nothing) && return nothing
    xbuf, cx, T = px
    nz = noise(fx.Σy, T)
    nz === nothing && return nothing
    kind, variance, scales = desc
    (length(scales) > 1 && length(scales) != cx.d) &&
        throw(DimensionMismatch("ARDTransform has $(length(scales)) scales, inputs have D=$(cx.d)"))
    ck = CKernel(kind, T === Float64 ? 0 : 1, variance, length(scales), isempty(scales) ? C_NULL : pointer(scales))
    return (; T, xbuf, cx, scales, ck, nbuf=nz[1], cn=nz[2], m=prior_mean(fx.f.gp, fx.x, T))
end
stock(fx::FiniteGP{<:HipGP}) = FiniteGP(fx.f.gp, fx.x, fx.Σy)

# ---- logpdf (src/finite_gp_projection.jl:306-311) -------------------------------------------------
function Distributions.logpdf(fx::FiniteGP{<:HipGP}, Y::AbstractVecOrMat{<:Real})
    a = marshal(fx)
    a === nothing && return logpdf(stock(fx), Y)
    size(Y, 1) == length(fx) || throw(DimensionMismatch("length(fx) = $(length(fx)) but Y has $(size(Y, 1)) rows"))
    T = a.T
    Yd = Matrix{T}(reshape(Y, size(Y, 1), :))
    out = Vector{T}(undef, size(Yd, 2))
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a Yd out begin
        check(ccall((:gp_logpdf, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}),
            fx.f.ctx.handle, a.ck, a.cx, a.cn, mptr, Yd, size(Yd, 1), size(Yd, 2), out))
    end
    return Y isa AbstractVector ? out[1] : out
end

# ---- value + gradient (what a ChainRulesCore.rrule for the accelerated logpdf returns; the reference relies on AD through
# logpdf: test/finite_gp_projection.jl:152-178, examples/1-mauna-loa/script.jl:228-240) ------------------------------
function logpdf_and_grad(fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    a = marshal(fx)
    a === nothing && throw(ArgumentError("kernel / noise form is not accelerated"))
    T = a.T
    yv = Vector{T}(y)
    lp = Ref{T}(zero(T)); dvar = Ref{Float64}(0.0)
    dscale = zeros(Float64, max(length(a.scales), 1))
    dnoise = Vector{T}(undef, a.cn.kind == 0 ? 1 : length(yv))
    dy = Vector{T}(undef, length(yv))
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a yv dscale dnoise dy begin
        check(ccall((:gp_logpdf_grad, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{T}, Ref{Float64}, Ptr{Float64},
                Ptr{Cvoid}, Ptr{Cvoid}),
            fx.f.ctx.handle, a.ck, a.cx, a.cn, mptr, yv, lp, dvar, dscale, dnoise, dy))
    end
    return lp[], (variance=dvar[], scale=dscale[1:length(a.scales)], noise=a.cn.kind == 0 ? dnoise[1] : dnoise, y=dy, mean=-dy)
end

# ---- posterior (src/exact_gpr_posterior.jl:29-35) -------------------------------------------------
mutable struct DeviceCholesky          # stands where `C::Cholesky` sits in PosteriorGP.data (:34)
    handle::Ptr{Cvoid}
    n::Int
    T::DataType
end
# C.U on the host (parity / debugging): N×N copy
function Base.getproperty(C::DeviceCholesky, s::Symbol)
    s === :U || return getfield(C, s)
    U = Matrix{C.T}(undef, C.n, C.n)
    check(ccall((:gp_posterior_get_factor, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), getfield(C, :handle), U))
    return UpperTriangular(U)
end

struct HipPosteriorGP{Tprior<:HipGP,Tdata} <: AbstractGP
    prior::Tprior
    data::Tdata                        # (α, C::DeviceCholesky, x, δ) — same field names as the reference (:34)
    logpdf_value::Float64              # logpdf(fx, y) from the same factorisation
end

function AbstractGPs.posterior(fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    a = marshal(fx)
    a === nothing && return posterior(stock(fx), y)
    length(y) == length(fx) || throw(DimensionMismatch("length(fx) != length(y)"))
    T = a.T
    yv = Vector{T}(y)
    δ = a.m === nothing ? copy(yv) : yv - a.m
    α = Vector{T}(undef, length(yv))
    lp = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a yv α begin
        check(ccall((:gp_posterior_fit, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Ref{T}),
            fx.f.ctx.handle, a.ck, a.cx, a.cn, mptr, yv, h, α, lp))
    end
    C = DeviceCholesky(h[], length(yv), T)
    finalizer(c -> ccall((:gp_posterior_free, libgpmi355), Int32, (Ptr{Cvoid},), getfield(c, :handle)), C)
    return HipPosteriorGP(fx.f, (α=α, C=C, x=fx.x, δ=δ), Float64(lp[]))
end

# ---- sequential conditioning (src/exact_gpr_posterior.jl:46-56; update_chol src/util/common_covmat_ops.jl:38-42) ----
function AbstractGPs.posterior(fx::FiniteGP{<:HipPosteriorGP}, y::AbstractVector{<:Real})
    post = fx.f
    px = points(fx.x)
    px === nothing && throw(ArgumentError("unsupported input container for the accelerated posterior"))
    xbuf, cx, T = px
    nz = noise(fx.Σy, T)
    nz === nothing && throw(ArgumentError("dense Σy is not accelerated"))
    m2 = prior_mean(post.prior.gp, fx.x, T)
    δ2 = m2 === nothing ? Vector{T}(y) : Vector{T}(y) - m2                 # :48-49
    δ = vcat(post.data.δ, δ2)                                              # :52
    α = Vector{T}(undef, length(δ))
    lp = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve xbuf nz δ α begin
        check(ccall((:gp_posterior_update, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Ref{T}),
            getfield(post.data.C, :handle), cx, nz[2], δ, h, α, lp))
    end
    C = DeviceCholesky(h[], length(δ), T)
    finalizer(c -> ccall((:gp_posterior_free, libgpmi355), Int32, (Ptr{Cvoid},), getfield(c, :handle)), C)
    return HipPosteriorGP(post.prior, (α=α, C=C, x=vcat(post.data.x, fx.x), δ=δ), Float64(lp[]))   # :54-55
end

# ---- sampling (src/finite_gp_projection.jl:233-237): m .+ C.U' * randn(rng, n, N), product on the device ----------
function Random.rand(rng::Random.AbstractRNG, fx::FiniteGP{<:HipGP}, N::Int)
    a = marshal(fx)
    a === nothing && return rand(rng, stock(fx), N)
    T = a.T
    p0 = posterior(fx, zeros(T, length(fx)))          # factor of cov(fx); α = C \ (0 − m) is discarded
    ξ = randn(rng, T, length(fx), N)
    out = similar(ξ)
    GC.@preserve ξ out check(ccall((:gp_posterior_factor_mul, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        getfield(p0.data.C, :handle), ξ, N, out))
    return mean(fx) .+ out
end

# ---- predictive methods (src/exact_gpr_posterior.jl:60-90) ----------------------------------------
function predict(f::HipPosteriorGP, x::AbstractVector, what::Integer)
    px = points(x)
    px === nothing && throw(ArgumentError("unsupported input container for the accelerated posterior"))
    xbuf, cx, T = px
    ns = length(x)
    pm = prior_mean(f.prior.gp, x, T)
    m = (what & 1) != 0 ? Vector{T}(undef, ns) : T[]
    v = (what & 2) != 0 ? Vector{T}(undef, ns) : T[]
    c = (what & 4) != 0 ? Matrix{T}(undef, ns, ns) : Matrix{T}(undef, 0, 0)
    GC.@preserve xbuf pm m v c begin
        check(ccall((:gp_posterior_predict, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
            getfield(f.data.C, :handle), cx, pm === nothing ? C_NULL : pointer(pm), what, m, v, c))
    end
    return m, v, c
end
Statistics.mean(f::HipPosteriorGP, x::AbstractVector) = predict(f, x, 1)[1]            # :60-62
Statistics.var(f::HipPosteriorGP, x::AbstractVector) = predict(f, x, 2)[2]             # :68-70
Statistics.cov(f::HipPosteriorGP, x::AbstractVector) = predict(f, x, 4)[3]             # :64-66
StatsBase.mean_and_var(f::HipPosteriorGP, x::AbstractVector) = predict(f, x, 3)[1:2]   # :85-90
function StatsBase.mean_and_cov(f::HipPosteriorGP, x::AbstractVector)                  # :78-83
    m, _, c = predict(f, x, 5)
    return m, c
end
function Statistics.cov(f::HipPosteriorGP, x::AbstractVector, z::AbstractVector)       # :72-76
    c = predict(f, vcat(x, z), 4)[3]
    return c[1:length(x), (length(x) + 1):end]
end

# ---- VFE / DTC (src/sparse_approximations.jl:58-75, 183-217, 248-254, 282-286) ---------------------
struct HipApproxPosteriorGP{Tapprox,Tprior<:HipGP} <: AbstractGP
    approx::Tapprox
    prior::Tprior
    handle::Base.RefValue{Ptr{Cvoid}}
    T::DataType
end

function vfe_call(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}, want_post::Bool)
    @assert approx.fz.f === fx.f                                                       # :59, :249, :283
    length(fx) == length(y) || throw(DimensionMismatch("length(fx) != length(y)"))    # :290-294
    a = marshal(fx)
    pz = points(approx.fz.x)
    (a === nothing || pz === nothing) && return nothing
    jit = approx.fz.Σy
    jit isa Union{Fill,Diagonal{<:Any,<:Fill}} || return nothing
    jitter = Float64(jit isa Fill ? FillArrays.getindex_value(jit) : FillArrays.getindex_value(jit.diag))
    T = a.T
    zbuf, cz, _ = pz
    yv = Vector{T}(y)
    obj = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a zbuf yv begin
        check(ccall((:gp_vfe_fit, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CPoints}, Ref{CNoise}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Int32,
                Ptr{Ptr{Cvoid}}, Ref{T}),
            fx.f.ctx.handle, a.ck, a.cx, cz, a.cn, jitter, mptr, yv, approx isa VFE ? 0 : 1,
            want_post ? Base.unsafe_convert(Ptr{Ptr{Cvoid}}, h) : Ptr{Ptr{Cvoid}}(C_NULL), obj))
    end
    return (h, Float64(obj[]), T)
end

function AbstractGPs.posterior(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    r = vfe_call(approx, fx, y, true)
    r === nothing && return posterior(typeof(approx)(FiniteGP(fx.f.gp, approx.fz.x, approx.fz.Σy)), stock(fx), y)
    p = HipApproxPosteriorGP(approx, fx.f, r[1], r[3])
    finalizer(hh -> ccall((:gp_vfe_free, libgpmi355), Int32, (Ptr{Cvoid},), hh[]), p.handle)
    return p
end
function AbstractGPs.approx_log_evidence(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    r = vfe_call(approx, fx, y, false)
    r === nothing &&
        return approx_log_evidence(typeof(approx)(FiniteGP(fx.f.gp, approx.fz.x, approx.fz.Σy)), stock(fx), y)
    return r[2]
end
AbstractGPs.elbo(vfe::VFE, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}) = approx_log_evidence(vfe, fx, y)  # :254

function vfe_predict(f::HipApproxPosteriorGP, x::AbstractVector, what::Integer)
    xbuf, cx, T = points(x)
    ns = length(x)
    pm = prior_mean(f.prior.gp, x, T)
    m = (what & 1) != 0 ? Vector{T}(undef, ns) : T[]
    v = (what & 2) != 0 ? Vector{T}(undef, ns) : T[]
    GC.@preserve xbuf pm m v begin
        check(ccall((:gp_vfe_predict, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
            f.handle[], cx, pm === nothing ? C_NULL : pointer(pm), what, m, v))
    end
    return m, v
end
Statistics.mean(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 1)[1]          # :183-185
Statistics.var(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 2)[2]           # :192-195
StatsBase.mean_and_var(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 3)      # :212-217
AbstractGPs.inducing_points(f::HipApproxPosteriorGP) = f.approx.fz.x                            # :219

# update_posterior with new observations, same pseudo-points (src/sparse_approximations.jl:87-121)
function AbstractGPs.update_posterior(f::HipApproxPosteriorGP, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    @assert f.prior === fx.f
    xbuf, cx, T = points(fx.x)
    nz = noise(fx.Σy, T)
    nz === nothing && throw(ArgumentError("dense Σy is not accelerated"))
    m2 = prior_mean(f.prior.gp, fx.x, T)
    yv = Vector{T}(y)
    obj = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve xbuf nz m2 yv begin
        check(ccall((:gp_vfe_update, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{T}),
            f.handle[], cx, nz[2], m2 === nothing ? C_NULL : pointer(m2), yv, h, obj))
    end
    p = HipApproxPosteriorGP(f.approx, f.prior, h, T)
    finalizer(hh -> ccall((:gp_vfe_free, libgpmi355), Int32, (Ptr{Cvoid},), hh[]), p.handle)
    return p
end

end # module
