# HipGPs.jl — Julia host shim: keeps the AbstractGP / FiniteGP / PosteriorGP surface of AbstractGPs.jl
# and routes the logpdf / posterior hot path through `ccall` into libgpmi355.so (include/gpmi355.h, ABI v4).
#
# NOT EXECUTED in the build container (Julia is not installed there — SURVEY.md §0 F3); it is the
# reference-side binding a maintainer adds (INTEGRATION.md).  abstractgps.jl_amd/api.py is its ctypes
# mirror (same entry points, argument meaning and errors) and is what tests/ run.  Reference plug-in point:
# "subtype AbstractGP and implement the FiniteGP primary API" (docs/src/api.md:18-30, 49-73).
#
#   f   = HipGP(GP(SqExponentialKernel()))          # wraps a stock GP              src/base_gp.jl:57-64
#   f8  = HipGP(GP(k), HipContext([0,1,2,3,4,5,6,7])) # same API, fits partitioned over 8 devices inside the library
#   fx  = f(x, 0.01)                                # stock FiniteGP ctor           src/finite_gp_projection.jl:13-37
#   logpdf(fx, y)                                   # -> gp_logpdf                  src/finite_gp_projection.jl:306-311
#   Zygote.gradient(θ -> logpdf(build(θ)(x, σ²), y), θ)   # -> rrule below -> gp_logpdf_grad
#   p   = posterior(fx, y)                          # -> gp_posterior_fit           src/exact_gpr_posterior.jl:29-35
#   mean_and_var(p(xs)); cov(p(xs))                 # -> gp_posterior_predict       src/exact_gpr_posterior.jl:60-90
#   logpdf(p(xs, σ²), ys); rand(rng, p(xs, σ²), 3)  # -> gp_posterior_logpdf / gp_posterior_rand   (device, no refit)
#   posterior(VFE(f(z, 1e-6)), fx, y); elbo(...)    # -> gp_vfe_fit / gp_vfe_predict src/sparse_approximations.jl:58-75,248-254
#   Zygote.gradient(θ -> elbo(VFE(build(θ)(z(θ), 1e-6)), build(θ)(x, σ²), y), θ)   # -> rrule -> gp_vfe_fit + gp_vfe_grad
#   update_posterior(pa, fx2, y2); update_posterior(pa, f(z2, 1e-6))   # -> gp_vfe_update / gp_vfe_append   :87-176
module HipGPs

using AbstractGPs
using AbstractGPs: AbstractGP, FiniteGP, GP, ZeroMean, ConstMean, CustomMean, VFE, DTC, mean_vector
using KernelFunctions
using KernelFunctions: SqExponentialKernel, Matern12Kernel, ExponentialKernel, Matern32Kernel, Matern52Kernel,
    TransformedKernel, ScaledKernel, ScaleTransform, ARDTransform, ColVecs, RowVecs
using LinearAlgebra, FillArrays, Statistics, StatsBase, Distributions, Random
using ChainRulesCore

export HipGP, HipPosteriorGP, HipApproxPosteriorGP, HipContext, logpdf_and_grad, elbo_and_grad

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
# HipContext(device)                 one GPU
# HipContext(devices; P, Q, nb)      several GPUs of the node driven from this one Julia process (gp_ctx_create_multi):
#                                    fp64 logpdf / posterior fits are partitioned 2D block-cyclically inside the library
mutable struct HipContext
    handle::Ptr{Cvoid}
    function HipContext(device::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:gp_ctx_create, libgpmi355), Int32, (Ref{Ptr{Cvoid}}, Int32, Ptr{Cvoid}), h, device, C_NULL))
        c = new(h[])
        finalizer(c -> ccall((:gp_ctx_destroy, libgpmi355), Int32, (Ptr{Cvoid},), c.handle), c)
        return c
    end
    function HipContext(devices::AbstractVector{<:Integer}; P::Integer=0, Q::Integer=0, nb::Integer=0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        devs = Vector{Int32}(devices)
        check(ccall((:gp_ctx_create_multi, libgpmi355), Int32, (Ref{Ptr{Cvoid}}, Ptr{Int32}, Int32, Int32, Int32, Int32),
            h, devs, length(devs), P, Q, nb))
        c = new(h[])
        finalizer(c -> ccall((:gp_ctx_destroy, libgpmi355), Int32, (Ptr{Cvoid},), c.handle), c)
        return c
    end
end
const _default_ctx = Ref{Union{Nothing,HipContext}}(nothing)
default_context() = something(_default_ctx[], (_default_ctx[] = HipContext(0)))
set_param!(c::HipContext, name::AbstractString, v::Integer) =
    check(ccall((:gp_ctx_set_param, libgpmi355), Int32, (Ptr{Cvoid}, Cstring, Int64), c.handle, name, v))
function get_param(c::HipContext, name::AbstractString)
    v = Ref{Int64}(0)
    check(ccall((:gp_ctx_get_param, libgpmi355), Int32, (Ptr{Cvoid}, Cstring, Ref{Int64}), c.handle, name, v))
    return v[]
end
trim!(c::HipContext) = check(ccall((:gp_ctx_trim, libgpmi355), Int32, (Ptr{Cvoid},), c.handle))
# multi-device contexts: fit attempts, repetitions after a failed self-check, forward solves on the distributed factor
function multi_stats(c::HipContext)
    f = Ref{Int64}(0); r = Ref{Int64}(0); s = Ref{Int64}(0)
    check(ccall((:gp_ctx_multi_stats, libgpmi355), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), c.handle, f, r, s))
    return (fits=f[], retries=r[], solves=s[])
end

# ---- the GP wrapper -------------------------------------------------------------------------------
struct HipGP{Tg<:GP} <: AbstractGP
    gp::Tg
    ctx::HipContext
end
HipGP(gp::GP) = HipGP(gp, default_context())

# internal AbstractGP API delegates to the wrapped GP (src/base_gp.jl:68-74) so that everything that is not
# accelerated (dense Σy, composite kernels, exotic input containers) keeps working through the stock methods.
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
# points(x, T) returns (buffer kept alive by the caller, CPoints) in eltype T, converting when needed; `nothing` for
# containers the ABI has no layout for (vector of vectors, ...): the caller then takes the stock path.
const HipFloat = Union{Float32,Float64}
points(x::AbstractVector{<:Real}, ::Type{T}) where {T<:HipFloat} = (b = convert(Vector{T}, x); (b, CPoints(pointer(b), length(b), 1, 0)))
function points(x::ColVecs, ::Type{T}) where {T<:HipFloat}
    b = convert(Matrix{T}, x.X)
    return (b, CPoints(pointer(b), size(b, 2), size(b, 1), 1))
end
function points(x::RowVecs, ::Type{T}) where {T<:HipFloat}
    b = convert(Matrix{T}, x.X)
    return (b, CPoints(pointer(b), size(b, 1), size(b, 2), 2))
end
points(::Any, ::Type) = nothing
# element type of an input container (Float32 in -> Float32 out is a tested reference property,
# test/finite_gp_projection.jl:180-191); anything that is not Float32 computes in Float64
input_eltype(x::AbstractVector{<:Real}) = eltype(x)
input_eltype(x::Union{ColVecs,RowVecs}) = eltype(x.X)
input_eltype(::Any) = Float64
hip_eltype(Ts...) = (T = promote_type(map(t -> t <: AbstractFloat ? t : Float64, Ts)...); T === Float32 ? Float32 : Float64)

noise(Σ::Diagonal{<:Any,<:Fill}, ::Type{T}) where {T} = (nothing, CNoise(0, Float64(FillArrays.getindex_value(Σ.diag)), C_NULL))
function noise(Σ::Diagonal, ::Type{T}) where {T}
    v = Vector{T}(Σ.diag)
    return (v, CNoise(1, 0.0, pointer(v)))
end
noise(::Any, ::Type) = nothing   # dense Σy: not accelerated

prior_mean(f::GP{<:ZeroMean}, x, ::Type{T}) where {T} = nothing
prior_mean(f::GP, x, ::Type{T}) where {T} = Vector{T}(mean_vector(f.mean, x))

# everything one call needs, or `nothing` => use the stock AbstractGPs path.  The compute type follows the reference's
# promotion (src/finite_gp_projection.jl:309): T = promote_type(eltype(x), eltype(Y)), restricted to Float32 / Float64.
function marshal(fx::FiniteGP{<:HipGP}, Ty::Type=input_eltype(fx.x))
    desc = descriptor(fx.f.gp.kernel)
    desc === nothing && return nothing
    T = hip_eltype(input_eltype(fx.x), Ty)
    px = points(fx.x, T)
    px === nothing && return nothing
    xbuf, cx = px
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
    a = marshal(fx, eltype(Y))
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

# ---- the two terms of logpdf on their own (src/finite_gp_projection.jl:313-337): sqmahal, logdetcov, gradlogpdf ----------------
function logpdf_terms(fx::FiniteGP{<:HipGP}, Y::Union{Nothing,AbstractVecOrMat{<:Real}}; logdet::Bool, sq::Bool)
    a = marshal(fx, Y === nothing ? input_eltype(fx.x) : eltype(Y))
    a === nothing && return nothing
    T = a.T
    Yd = Y === nothing ? nothing : Matrix{T}(reshape(Y, size(Y, 1), :))
    Yd === nothing || size(Yd, 1) == length(fx) || throw(DimensionMismatch("length(fx) = $(length(fx)) but Y has $(size(Yd, 1)) rows"))
    ld = Ref{T}(zero(T))
    out = Vector{T}(undef, Yd === nothing ? 1 : size(Yd, 2))
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a Yd out begin
        check(ccall((:gp_logpdf_terms, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
            fx.f.ctx.handle, a.ck, a.cx, a.cn, mptr, Yd === nothing ? C_NULL : pointer(Yd), length(fx),
            Yd === nothing ? 0 : size(Yd, 2), logdet ? Base.unsafe_convert(Ptr{Cvoid}, ld) : C_NULL, sq ? pointer(out) : C_NULL))
    end
    return (ld[], out)
end
function Distributions.logdetcov(fx::FiniteGP{<:HipGP})                                  # :313
    r = logpdf_terms(fx, nothing; logdet=true, sq=false)
    return r === nothing ? logdetcov(stock(fx)) : r[1]
end
function Distributions.sqmahal(fx::FiniteGP{<:HipGP}, x::AbstractVector)                 # :315-318
    r = logpdf_terms(fx, x; logdet=false, sq=true)
    return r === nothing ? sqmahal(stock(fx), x) : r[2][1]
end
function Distributions.sqmahal(fx::FiniteGP{<:HipGP}, X::AbstractMatrix)                 # :320-323
    r = logpdf_terms(fx, X; logdet=false, sq=true)
    return r === nothing ? sqmahal(stock(fx), X) : r[2]
end
# gradlogpdf(f, x) = C \ (m .- x) (:328-337): −α of the posterior fit; further columns reuse that fit's resident factor
function Distributions.gradlogpdf(fx::FiniteGP{<:HipGP}, x::AbstractVector)
    marshal(fx, eltype(x)) === nothing && return gradlogpdf(stock(fx), x)
    return -posterior(fx, x).data.α
end
function Distributions.gradlogpdf(fx::FiniteGP{<:HipGP}, X::AbstractMatrix)
    a = marshal(fx, eltype(X))
    a === nothing && return gradlogpdf(stock(fx), X)
    post = posterior(fx, X[:, 1])
    T = a.T
    D = a.m === nothing ? Matrix{T}(X) : Matrix{T}(X) .- a.m
    out = similar(D)
    # `post` must outlive the call: its finalizer frees the device factor the solve reads
    GC.@preserve post D out check(ccall((:gp_posterior_solve, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        getfield(post.data.C, :handle), D, size(D, 2), out))
    finalize(post.data.C)   # an O(n²) device factor: release it now instead of leaving it to the GC
    return -out
end

# ---- value + gradient: one factorisation, C⁻¹ by blocked TRSM + MFMA SYRK, one fused ½Σ(αᵢαⱼ − C⁻¹ᵢⱼ)∂Cᵢⱼ pass ----------
function logpdf_and_grad(fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}; wrt_x::Bool=true)
    a = marshal(fx, eltype(y))
    a === nothing && throw(ArgumentError("kernel / noise form is not accelerated"))
    T = a.T
    yv = Vector{T}(y)
    lp = Ref{T}(zero(T)); dvar = Ref{Float64}(0.0)
    dscale = zeros(Float64, max(length(a.scales), 1))
    dnoise = Vector{T}(undef, a.cn.kind == 0 ? 1 : length(yv))
    dy = Vector{T}(undef, length(yv))
    dx = wrt_x ? similar(a.xbuf) : T[]                  # same container layout as the inputs (Vector / D×N / N×D)
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a yv dscale dnoise dy dx begin
        check(ccall((:gp_logpdf_grad, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{T}, Ref{Float64}, Ptr{Float64},
                Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
            fx.f.ctx.handle, a.ck, a.cx, a.cn, mptr, yv, lp, dvar, dscale, dnoise, dy, wrt_x ? pointer(dx) : C_NULL))
    end
    return lp[], (variance=dvar[], scale=dscale[1:length(a.scales)], noise=a.cn.kind == 0 ? dnoise[1] : dnoise, y=dy, mean=-dy,
        x=wrt_x ? dx : nothing)
end

# ---- reverse-mode rule: Zygote / any ChainRules-based AD differentiates THROUGH the ccall -----------------------------------
# The reference's users differentiate logpdf by AD (test/finite_gp_projection.jl:152-178, test/mean_function.jl:38-56,
# examples/1-mauna-loa/script.jl:201-240); a ccall is opaque to AD, so the accelerated logpdf carries its own pullback built
# from gp_logpdf_grad.  Tangents are structural, mirroring how `descriptor` walks the kernel:
#   ScaledKernel.σ²  (1-vector)  <- ∂/∂variance · (total variance / σ²)      TransformedKernel.transform.s / .v  <- ∂/∂scale
#   ConstMean.c <- Σ_i α_i        FiniteGP.Σy (Diagonal{Fill} value / Diagonal diag) <- ∂/∂σ² / ½(α_i² − C⁻¹_ii)        y <- −α
#   x (Vector / ColVecs.X / RowVecs.X) <- ∂/∂x, the input gradient a deep-kernel model back-propagates into its feature map
# CustomMean parameters are not differentiated here (@not_implemented); the prior mean is taken as constant in x.
kernel_tangent(k, dvar, variance, dscale) = NoTangent()
function kernel_tangent(k::ScaledKernel, dvar, variance, dscale)
    σ² = only(k.σ²)
    return Tangent{typeof(k)}(; kernel=kernel_tangent(k.kernel, dvar, variance, dscale), σ²=[dvar * variance / σ²])
end
function kernel_tangent(k::TransformedKernel{<:Any,<:ScaleTransform}, dvar, variance, dscale)
    return Tangent{typeof(k)}(; kernel=kernel_tangent(k.kernel, dvar, variance, dscale),
        transform=Tangent{typeof(k.transform)}(; s=[dscale[1]]))
end
function kernel_tangent(k::TransformedKernel{<:Any,<:ARDTransform}, dvar, variance, dscale)
    return Tangent{typeof(k)}(; kernel=kernel_tangent(k.kernel, dvar, variance, dscale),
        transform=Tangent{typeof(k.transform)}(; v=collect(dscale)))
end
mean_tangent(::ZeroMean, dm) = NoTangent()
mean_tangent(m::ConstMean, dm) = Tangent{typeof(m)}(; c=sum(dm))
mean_tangent(m, dm) = ChainRulesCore.@not_implemented("HipGPs: gradients w.r.t. CustomMean parameters go through the stock path")
input_tangent(x::AbstractVector{<:Real}, dx, Δ) = Δ .* dx
input_tangent(x::ColVecs, dx, Δ) = Tangent{typeof(x)}(; X=Δ .* dx)
input_tangent(x::RowVecs, dx, Δ) = Tangent{typeof(x)}(; X=Δ .* dx)
noise_tangent(Σ::Diagonal{<:Any,<:Fill}, dn) = Tangent{typeof(Σ)}(; diag=Tangent{typeof(Σ.diag)}(; value=dn))
noise_tangent(Σ::Diagonal, dn) = Tangent{typeof(Σ)}(; diag=dn)

function ChainRulesCore.rrule(config::RuleConfig{>:HasReverseMode}, ::typeof(Distributions.logpdf), fx::FiniteGP{<:HipGP},
    y::AbstractVector{<:Real})
    desc = descriptor(fx.f.gp.kernel)
    if desc === nothing || marshal(fx, eltype(y)) === nothing   # not accelerated: the stock path keeps its own AD
        lp0, back = rrule_via_ad(config, (g_, x_, Σ_, y_) -> logpdf(FiniteGP(g_, x_, Σ_), y_), fx.f.gp, fx.x, fx.Σy, y)
        return lp0, function (Δ)
            _, dg, dx, dΣ, dy = back(Δ)
            return NoTangent(), Tangent{typeof(fx)}(; f=Tangent{typeof(fx.f)}(; gp=dg, ctx=NoTangent()), x=dx, Σy=dΣ), dy
        end
    end
    lp, g = logpdf_and_grad(fx, y)
    function logpdf_hip_pullback(Δ)
        Δr = unthunk(Δ)
        gp = fx.f.gp
        dk = kernel_tangent(gp.kernel, Δr * g.variance, desc[2], Δr .* g.scale)
        dgp = Tangent{typeof(gp)}(; mean=mean_tangent(gp.mean, Δr .* g.mean), kernel=dk)
        df = Tangent{typeof(fx.f)}(; gp=dgp, ctx=NoTangent())
        dfx = Tangent{typeof(fx)}(; f=df, x=input_tangent(fx.x, g.x, Δr), Σy=noise_tangent(fx.Σy, Δr .* g.noise))
        return NoTangent(), dfx, Δr .* g.y
    end
    return lp, logpdf_hip_pullback
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
# `C \ B` of user code (what `post.data.C \ v` does on a LinearAlgebra.Cholesky): forward + backward sweeps over the resident factor —
# on the block-cyclic pieces when the factor comes from a multi-device fit (gp_posterior_solve)
function Base.:\(C::DeviceCholesky, B::AbstractVecOrMat{<:Real})
    size(B, 1) == getfield(C, :n) || throw(DimensionMismatch("C is $(getfield(C, :n))×$(getfield(C, :n)), B has $(size(B, 1)) rows"))
    T = getfield(C, :T)
    D = Matrix{T}(reshape(B, size(B, 1), :))
    out = similar(D)
    GC.@preserve D out check(ccall((:gp_posterior_solve, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        getfield(C, :handle), D, size(D, 2), out))
    return B isa AbstractVector ? vec(out) : out
end
# logdet(post.data.C) (what src/finite_gp_projection.jl:310 does with its Cholesky): kept from the fit, no sweep over the factor
function LinearAlgebra.logdet(C::DeviceCholesky)
    out = Ref{Float64}(0.0)
    GC.@preserve C check(ccall((:gp_posterior_logdet, libgpmi355), Int32, (Ptr{Cvoid}, Ref{Float64}), getfield(C, :handle), out))
    return convert(getfield(C, :T), out[])
end
function device_cholesky(h::Ptr{Cvoid}, n::Int, ::Type{T}) where {T}
    C = DeviceCholesky(h, n, T)
    finalizer(c -> ccall((:gp_posterior_free, libgpmi355), Int32, (Ptr{Cvoid},), getfield(c, :handle)), C)
    return C
end

struct HipPosteriorGP{Tprior<:HipGP,Tdata} <: AbstractGP
    prior::Tprior
    data::Tdata                        # (α, C::DeviceCholesky, x, δ) — same field names as the reference (:34)
    logpdf_value::Float64              # logpdf(fx, y) from the same factorisation
end

function AbstractGPs.posterior(fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    a = marshal(fx, eltype(y))
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
    return HipPosteriorGP(fx.f, (α=α, C=device_cholesky(h[], length(yv), T), x=fx.x, δ=δ), Float64(lp[]))
end

# inputs / noise of a FiniteGP over one of the posterior types, in the posterior's element type
function joint_args(fx::FiniteGP, gp::GP, ::Type{T}) where {T}
    px = points(fx.x, T)
    px === nothing && throw(ArgumentError("unsupported input container for the accelerated posterior (use a Vector, ColVecs or RowVecs)"))
    nz = noise(fx.Σy, T)
    nz === nothing && throw(ArgumentError("dense Σy is not accelerated"))
    return px[1], px[2], nz[1], nz[2], prior_mean(gp, fx.x, T)
end

# ---- sequential conditioning (src/exact_gpr_posterior.jl:46-56; update_chol src/util/common_covmat_ops.jl:38-42) ----
function AbstractGPs.posterior(fx::FiniteGP{<:HipPosteriorGP}, y::AbstractVector{<:Real})
    post = fx.f
    T = getfield(post.data.C, :T)
    xbuf, cx, nbuf, cn, m2 = joint_args(fx, post.prior.gp, T)
    δ2 = m2 === nothing ? Vector{T}(y) : Vector{T}(y) - m2                 # :48-49
    δ = vcat(post.data.δ, δ2)                                              # :52
    α = Vector{T}(undef, length(δ))
    lp = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve xbuf nbuf δ α begin
        check(ccall((:gp_posterior_update, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Ref{T}),
            getfield(post.data.C, :handle), cx, cn, δ, h, α, lp))
    end
    return HipPosteriorGP(post.prior, (α=α, C=device_cholesky(h[], length(δ), T), x=vcat(post.data.x, fx.x), δ=δ), Float64(lp[]))   # :54-55
end

# ---- sampling (src/finite_gp_projection.jl:233-237, 271-277): m .+ C.U' * randn(rng, n, N), factor and product on the device ----
function Random.rand(rng::Random.AbstractRNG, fx::FiniteGP{<:HipGP}, N::Int)
    a = marshal(fx)
    a === nothing && return rand(rng, stock(fx), N)
    T = a.T
    p0 = posterior(FiniteGP(HipGP(GP(fx.f.gp.kernel), fx.f.ctx), fx.x, fx.Σy), zeros(T, length(fx)))   # factor of cov(fx)
    ξ = randn(rng, T, length(fx), N)
    out = similar(ξ)
    GC.@preserve ξ out check(ccall((:gp_posterior_factor_mul, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
        getfield(p0.data.C, :handle), ξ, N, out))
    return mean(fx) .+ out
end
Random.rand(rng::Random.AbstractRNG, fx::FiniteGP{<:HipGP}) = vec(rand(rng, fx, 1))
Random.rand(fx::FiniteGP{<:HipGP}, N::Int) = rand(Random.default_rng(), fx, N)
Random.rand(fx::FiniteGP{<:HipGP}) = rand(Random.default_rng(), fx)

post_handle(f::HipPosteriorGP) = (getfield(f.data.C, :handle), getfield(f.data.C, :T), :gp_posterior_logpdf, :gp_posterior_rand)

# logpdf(post(x*, Σy*), Y*) and rand(post(x*, Σy*)) — the FiniteGP-over-posterior path of the reference (mean_and_cov of the
# posterior + Σy*, cholesky, logdet/_sqmahal or m + U'ξ) as ONE device call each; nothing is refitted
function joint_logpdf(f, fx::FiniteGP, Y::AbstractVecOrMat{<:Real})
    h, T, sym_lp, _ = post_handle(f)
    size(Y, 1) == length(fx) || throw(DimensionMismatch("length(fx) = $(length(fx)) but Y has $(size(Y, 1)) rows"))
    xbuf, cx, nbuf, cn, pm = joint_args(fx, f.prior.gp, T)
    Yd = Matrix{T}(reshape(Y, size(Y, 1), :))
    out = Vector{T}(undef, size(Yd, 2))
    GC.@preserve xbuf nbuf pm Yd out begin
        rc = sym_lp === :gp_posterior_logpdf ?
            ccall((:gp_posterior_logpdf, libgpmi355), Int32, (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Ref{CNoise}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}),
                h, cx, pm === nothing ? C_NULL : pointer(pm), cn, Yd, size(Yd, 1), size(Yd, 2), out) :
            ccall((:gp_vfe_logpdf, libgpmi355), Int32, (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Ref{CNoise}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}),
                h, cx, pm === nothing ? C_NULL : pointer(pm), cn, Yd, size(Yd, 1), size(Yd, 2), out)
        check(rc)
    end
    return Y isa AbstractVector ? out[1] : out
end
function joint_rand!(rng::Random.AbstractRNG, f, fx::FiniteGP, out::AbstractVecOrMat{<:Real})
    h, T, _, sym_r = post_handle(f)
    size(out, 1) == length(fx) || throw(DimensionMismatch("length(fx) = $(length(fx)) but the output has $(size(out, 1)) rows"))
    xbuf, cx, nbuf, cn, pm = joint_args(fx, f.prior.gp, T)
    N = size(out, 2)
    ξ = randn(rng, T, length(fx), N)
    res = Matrix{T}(undef, length(fx), N)
    GC.@preserve xbuf nbuf pm ξ res begin
        rc = sym_r === :gp_posterior_rand ?
            ccall((:gp_posterior_rand, libgpmi355), Int32, (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Ref{CNoise}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
                h, cx, pm === nothing ? C_NULL : pointer(pm), cn, ξ, N, res) :
            ccall((:gp_vfe_rand, libgpmi355), Int32, (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Ref{CNoise}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
                h, cx, pm === nothing ? C_NULL : pointer(pm), cn, ξ, N, res)
        check(rc)
    end
    out .= reshape(res, size(out))
    return out
end

# ---- predictive methods (src/exact_gpr_posterior.jl:60-90) ----------------------------------------
function predict(f::HipPosteriorGP, x::AbstractVector, what::Integer)
    T = getfield(f.data.C, :T)
    px = points(x, T)
    px === nothing && throw(ArgumentError("unsupported input container for the accelerated posterior (use a Vector, ColVecs or RowVecs)"))
    xbuf, cx = px
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
struct HipApproxPosteriorGP{Tapprox,Tprior<:HipGP,Tx,TΣ} <: AbstractGP
    approx::Tapprox
    prior::Tprior
    handle::Base.RefValue{Ptr{Cvoid}}
    T::DataType
    objective::Float64                 # ELBO / DTC evidence from the same streamed pass
    x::Tx                              # cache.x  (src/sparse_approximations.jl:73, :115): every observation input seen so far
    Σy::TΣ                             # cache.Σy (:73, :100): Diagonal over every observation seen so far
end
post_handle(f::HipApproxPosteriorGP) = (getfield(f, :handle)[], getfield(f, :T), :gp_vfe_logpdf, :gp_vfe_rand)
function approx_posterior(approx, prior, h::Base.RefValue{Ptr{Cvoid}}, ::Type{T}, obj, x, Σy) where {T}
    finalizer(hh -> ccall((:gp_vfe_free, libgpmi355), Int32, (Ptr{Cvoid},), hh[]), h)
    return HipApproxPosteriorGP(approx, prior, h, T, Float64(obj), x, Σy)
end
# `post.data` — the cache of the reference (src/sparse_approximations.jl:73), read field by field by its tests
# (test/sparse_approximations.jl:48-55, 76-83): m_ε, Λ_ε (a LinearAlgebra.Cholesky, so Λ_ε.U works), U, α, b_y from the device
# (gp_vfe_get, gp_vfe_get_factors, gp_vfe_get_by), x and Σy from the host side.  B_εf (M×N) is never materialised and is not a field.
# The view is LAZY: `post.data` costs nothing, `post.data.α` moves M numbers, `post.data.U` one M×M factor — a NamedTuple built on every
# access would download both factors and the N-vector b_y each time (256 MB per `post.data.α` at M = 4 096, N = 262 144).
struct HipVfeCache{Tf}
    f::Tf
end
Base.propertynames(::HipVfeCache) = (:m_ε, :Λ_ε, :U, :α, :b_y, :x, :Σy)
function Base.getproperty(c::HipVfeCache, s::Symbol)
    f = getfield(c, :f)
    h, T = getfield(f, :handle), getfield(f, :T)
    s === :x && return getfield(f, :x)
    s === :Σy && return getfield(f, :Σy)
    m = Int(ccall((:gp_vfe_m, libgpmi355), Int64, (Ptr{Cvoid},), h[]))
    if s === :α || s === :m_ε
        v = Vector{T}(undef, m)
        GC.@preserve f v begin
            p = convert(Ptr{Cvoid}, pointer(v))
            check(ccall((:gp_vfe_get, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h[], s === :α ? p : C_NULL, s === :m_ε ? p : C_NULL))
        end
        return v
    elseif s === :U || s === :Λ_ε
        A = Matrix{T}(undef, m, m)
        GC.@preserve f A begin
            p = convert(Ptr{Cvoid}, pointer(A))
            check(ccall((:gp_vfe_get_factors, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h[], s === :U ? p : C_NULL, s === :Λ_ε ? p : C_NULL))
        end
        return s === :U ? UpperTriangular(A) : Cholesky(A, 'U', 0)
    elseif s === :b_y
        n = Int(ccall((:gp_vfe_n, libgpmi355), Int64, (Ptr{Cvoid},), h[]))
        b_y = Vector{T}(undef, n)
        GC.@preserve f b_y check(ccall((:gp_vfe_get_by, libgpmi355), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), h[], b_y))
        return b_y
    end
    throw(ArgumentError("the VFE cache has no field $s (fields: $(propertynames(c)); B_εf is never materialised)"))
end
Base.getproperty(f::HipApproxPosteriorGP, s::Symbol) = s === :data ? HipVfeCache(f) : getfield(f, s)
Base.propertynames(::HipApproxPosteriorGP) = (:approx, :prior, :data, :objective)

# FiniteGP API of the two posterior types on the device
for PT in (:HipPosteriorGP, :HipApproxPosteriorGP)
    @eval begin
        Distributions.logpdf(fx::FiniteGP{<:$PT}, Y::AbstractVecOrMat{<:Real}) = joint_logpdf(fx.f, fx, Y)
        Random.rand(rng::Random.AbstractRNG, fx::FiniteGP{<:$PT}, N::Int) =
            joint_rand!(rng, fx.f, fx, Matrix{post_handle(fx.f)[2]}(undef, length(fx), N))
        Random.rand(rng::Random.AbstractRNG, fx::FiniteGP{<:$PT}) = joint_rand!(rng, fx.f, fx, Vector{post_handle(fx.f)[2]}(undef, length(fx)))
        Random.rand!(rng::Random.AbstractRNG, fx::FiniteGP{<:$PT}, y::AbstractVecOrMat{<:Real}) = joint_rand!(rng, fx.f, fx, y)   # :271-277
    end
end
function Random.rand!(rng::Random.AbstractRNG, fx::FiniteGP{<:HipGP}, y::AbstractVecOrMat{<:Real})            # :271-277
    y .= y isa AbstractVector ? rand(rng, fx) : rand(rng, fx, size(y, 2))
    return y
end

function vfe_call(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}, want_post::Bool)
    @assert approx.fz.f === fx.f                                                       # :59, :249, :283
    length(fx) == length(y) || throw(DimensionMismatch("length(fx) != length(y)"))    # :290-294
    a = marshal(fx, eltype(y))
    a === nothing && return nothing
    T = a.T
    pz = points(approx.fz.x, T)
    pz === nothing && return nothing
    jit = approx.fz.Σy
    jit isa Diagonal{<:Any,<:Fill} || return nothing
    jitter = Float64(FillArrays.getindex_value(jit.diag))
    zbuf, cz = pz
    yv = Vector{T}(y)
    obj = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    mptr = a.m === nothing ? C_NULL : pointer(a.m)
    GC.@preserve a zbuf yv h begin
        check(ccall((:gp_vfe_fit, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CKernel}, Ref{CPoints}, Ref{CPoints}, Ref{CNoise}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Int32,
                Ptr{Ptr{Cvoid}}, Ref{T}),
            fx.f.ctx.handle, a.ck, a.cx, cz, a.cn, jitter, mptr, yv, approx isa VFE ? 0 : 1,
            want_post ? h : Ptr{Ptr{Cvoid}}(C_NULL), obj))
    end
    return (h, Float64(obj[]), T)
end

function AbstractGPs.posterior(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    r = vfe_call(approx, fx, y, true)
    r === nothing && return posterior(typeof(approx)(FiniteGP(fx.f.gp, approx.fz.x, approx.fz.Σy)), stock(fx), y)
    return approx_posterior(approx, fx.f, r[1], r[3], r[2], fx.x, Diagonal(collect(diag(fx.Σy))))
end
function AbstractGPs.approx_log_evidence(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    r = vfe_call(approx, fx, y, false)
    r === nothing &&
        return approx_log_evidence(typeof(approx)(FiniteGP(fx.f.gp, approx.fz.x, approx.fz.Σy)), stock(fx), y)
    return r[2]
end
AbstractGPs.elbo(vfe::VFE, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}) = approx_log_evidence(vfe, fx, y)  # :254

# ---- value + gradient of the sparse objective: one fit + one backward pass over the retained observations (gp_vfe_grad) ------------------
# The reference's users maximise `elbo(VFE(f(z, jitter)), f(x, Σy), y)` over kernel parameters and pseudo-points by AD / finite differences
# (examples/0-intro-1d/script.jl:385-394); the ccall is opaque to AD, so the accelerated objective carries its own pullback.
# ∂/∂z needs an fp64 fit (an fp32 fit's streamed B Bᵀ does not carry it: the library returns −7), so fp32 inputs get no z-tangent.
function elbo_and_grad(approx::Union{VFE,DTC}, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}; wrt_x::Bool=false)
    r = vfe_call(approx, fx, y, true)
    r === nothing && throw(ArgumentError("kernel / noise / pseudo-input form is not accelerated"))
    h, obj, T = r
    a = marshal(fx, eltype(y))
    zbuf, cz = points(approx.fz.x, T)
    n = length(y)
    dvar = Ref{Float64}(0.0); dns = Ref{Float64}(0.0)
    dscale = zeros(Float64, max(length(a.scales), 1))
    dnoise = Vector{T}(undef, n); dy = Vector{T}(undef, n)
    dz = T === Float64 ? similar(zbuf, Float64) : Float64[]             # the container layout of the pseudo-inputs
    dx = wrt_x ? similar(a.xbuf) : T[]
    try
        GC.@preserve dscale dnoise dy dz dx check(ccall((:gp_vfe_grad, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ref{Float64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Cvoid}, Int32),
            h[], dvar, dscale, dns, dnoise, dy, T === Float64 ? pointer(dz) : Ptr{Float64}(C_NULL), cz.layout,
            wrt_x ? pointer(dx) : C_NULL, a.cx.layout))
    finally
        ccall((:gp_vfe_free, libgpmi355), Int32, (Ptr{Cvoid},), h[])
    end
    return obj, (variance=dvar[], scale=dscale[1:length(a.scales)], noise=a.cn.kind == 0 ? dns[] : dnoise, y=dy, mean=-dy,
        z=T === Float64 ? dz : nothing, x=wrt_x ? dx : nothing)
end

function ChainRulesCore.rrule(config::RuleConfig{>:HasReverseMode}, ::typeof(AbstractGPs.approx_log_evidence), approx::Union{VFE,DTC},
    fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    desc = descriptor(fx.f.gp.kernel)
    if desc === nothing || marshal(fx, eltype(y)) === nothing || !(approx.fz.Σy isa Diagonal{<:Any,<:Fill})   # stock path keeps its own AD
        v0, back = rrule_via_ad(config, (g_, z_, J_, x_, Σ_, y_) -> approx_log_evidence(typeof(approx)(FiniteGP(g_, z_, J_)), FiniteGP(g_, x_, Σ_), y_),
            fx.f.gp, approx.fz.x, approx.fz.Σy, fx.x, fx.Σy, y)
        return v0, function (Δ)
            _, dg, dzz, dJ, dx, dΣ, dy = back(Δ)
            dfz = Tangent{typeof(approx.fz)}(; f=Tangent{typeof(fx.f)}(; gp=dg, ctx=NoTangent()), x=dzz, Σy=dJ)
            return NoTangent(), Tangent{typeof(approx)}(; fz=dfz), Tangent{typeof(fx)}(; f=Tangent{typeof(fx.f)}(; gp=dg, ctx=NoTangent()), x=dx, Σy=dΣ), dy
        end
    end
    v, g = elbo_and_grad(approx, fx, y; wrt_x=true)
    function sparse_objective_pullback(Δ)
        Δr = unthunk(Δ)
        gp = fx.f.gp
        dk = kernel_tangent(gp.kernel, Δr * g.variance, desc[2], Δr .* g.scale)
        dgp = Tangent{typeof(gp)}(; mean=mean_tangent(gp.mean, Δr .* g.mean), kernel=dk)
        df = Tangent{typeof(fx.f)}(; gp=dgp, ctx=NoTangent())
        dfx = Tangent{typeof(fx)}(; f=df, x=input_tangent(fx.x, g.x, Δr), Σy=noise_tangent(fx.Σy, Δr .* g.noise))
        # the prior is shared (approx.fz.f === fx.f): its tangent travels with fx; the pseudo-inputs get theirs, the jitter is a constant of the fit
        dzt = g.z === nothing ? ChainRulesCore.@not_implemented("HipGPs: ∂/∂z needs Float64 inputs") : input_tangent(approx.fz.x, g.z, Δr)
        dfz = Tangent{typeof(approx.fz)}(; f=ZeroTangent(), x=dzt, Σy=ZeroTangent())
        return NoTangent(), Tangent{typeof(approx)}(; fz=dfz), dfx, Δr .* g.y
    end
    return v, sparse_objective_pullback
end
ChainRulesCore.rrule(config::RuleConfig{>:HasReverseMode}, ::typeof(AbstractGPs.elbo), vfe::VFE, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real}) =
    ChainRulesCore.rrule(config, AbstractGPs.approx_log_evidence, vfe, fx, y)

function vfe_predict(f::HipApproxPosteriorGP, x::AbstractVector, what::Integer)
    T = getfield(f, :T)
    px = points(x, T)
    px === nothing && throw(ArgumentError("unsupported input container for the accelerated posterior (use a Vector, ColVecs or RowVecs)"))
    xbuf, cx = px
    ns = length(x)
    pm = prior_mean(f.prior.gp, x, T)
    m = (what & 1) != 0 ? Vector{T}(undef, ns) : T[]
    v = (what & 2) != 0 ? Vector{T}(undef, ns) : T[]
    c = (what & 4) != 0 ? Matrix{T}(undef, ns, ns) : Matrix{T}(undef, 0, 0)
    GC.@preserve f xbuf pm m v c begin
        check(ccall((:gp_vfe_predict, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
            getfield(f, :handle)[], cx, pm === nothing ? C_NULL : pointer(pm), what, m, v, c))
    end
    return m, v, c
end
Statistics.mean(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 1)[1]          # :183-185
Statistics.cov(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 4)[3]           # :187-190
Statistics.var(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 2)[2]           # :192-195
function Statistics.cov(f::HipApproxPosteriorGP, x::AbstractVector, z::AbstractVector)         # :197-203
    c = vfe_predict(f, vcat(x, z), 4)[3]
    return c[1:length(x), (length(x) + 1):end]
end
function StatsBase.mean_and_cov(f::HipApproxPosteriorGP, x::AbstractVector)                    # :205-210
    m, _, c = vfe_predict(f, x, 5)
    return m, c
end
StatsBase.mean_and_var(f::HipApproxPosteriorGP, x::AbstractVector) = vfe_predict(f, x, 3)[1:2] # :212-217
AbstractGPs.inducing_points(f::HipApproxPosteriorGP) = f.approx.fz.x                            # :219

# update_posterior with new observations, same pseudo-points (src/sparse_approximations.jl:87-121)
function AbstractGPs.update_posterior(f::HipApproxPosteriorGP, fx::FiniteGP{<:HipGP}, y::AbstractVector{<:Real})
    @assert f.prior === fx.f
    T = getfield(f, :T)
    xbuf, cx, nbuf, cn, m2 = joint_args(fx, f.prior.gp, T)
    yv = Vector{T}(y)
    obj = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve f xbuf nbuf m2 yv begin
        check(ccall((:gp_vfe_update, libgpmi355), Int32,
            (Ptr{Cvoid}, Ref{CPoints}, Ref{CNoise}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{T}),
            getfield(f, :handle)[], cx, cn, m2 === nothing ? C_NULL : pointer(m2), yv, h, obj))
    end
    Σy = Diagonal(vcat(diag(getfield(f, :Σy)), diag(fx.Σy)))                                   # :99-100 (block-diagonal of two Diagonals)
    return approx_posterior(f.approx, f.prior, h, T, obj[], vcat(getfield(f, :x), fx.x), Σy)   # :115
end

# update_posterior with new pseudo-points (src/sparse_approximations.jl:131-176): bordered K_zz factor + re-streamed new block
# rows on the device; the approximation object is rebuilt with z = vcat(z_old, z_new) like _update_approx (:178-179)
function AbstractGPs.update_posterior(f::HipApproxPosteriorGP, fz::FiniteGP{<:HipGP})
    @assert f.prior === fz.f                                                                   # :132
    T = getfield(f, :T)
    pz = points(fz.x, T)
    pz === nothing && throw(ArgumentError("unsupported container for the new pseudo-points"))
    zbuf, cz = pz
    obj = Ref{T}(zero(T))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve f zbuf begin
        check(ccall((:gp_vfe_append, libgpmi355), Int32, (Ptr{Cvoid}, Ref{CPoints}, Ref{Ptr{Cvoid}}, Ref{T}), getfield(f, :handle)[], cz, h, obj))
    end
    fz_new = f.approx.fz.f(vcat(f.approx.fz.x, fz.x), f.approx.fz.Σy)                         # :160-162
    return approx_posterior(typeof(f.approx)(fz_new), f.prior, h, T, obj[], getfield(f, :x), getfield(f, :Σy))
end

end # module
