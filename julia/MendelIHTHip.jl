# MendelIHTHip.jl -- Julia glue a MendelIHT.jl maintainer adds to route the IHT hot path to
# libmendeliht_hip.so (C ABI: include/mendeliht_hip.h).  NOT executed in this repository's
# image (no Julia toolchain); kept thin so it can be checked by eye against the header.
#
# Usage:
#   using MendelIHT, SnpArrays, MendelIHTHip
#   x   = HipSnpLinAlg{Float64}(SnpArray("normal.bed"); center=true, scale=true, impute=true)
#   res = fit_iht(y, x, z; k=7)          # dispatches to the GPU: same IHTResult
#   mse = cv_iht(y, x, z; path=1:20, q=5, folds=folds)
module MendelIHTHip

using MendelIHT, SnpArrays, Distributions, GLM, LinearAlgebra
import MendelIHT: fit_iht, cv_iht, IHTResult

const LIB = get(ENV, "MENDELIHT_HIP_LIB", "libmendeliht_hip.so")

# ---- status codes -> the exceptions the reference throws --------------------------------
function check(rc::Cint)
    rc == 0 && return
    buf = Vector{UInt8}(undef, 512)
    ccall((:mih_last_error, LIB), Cint, (Ptr{UInt8}, Csize_t), buf, 512)
    msg = unsafe_string(pointer(buf))
    rc == 1 && throw(DimensionMismatch(msg))
    rc in (2, 3) && throw(ArgumentError(msg))
    error(msg)        # NaN/Inf loglikelihood (fit.jl:259-260), HIP errors, OOM, no device
end

# ---- the design-matrix type: the dispatch hook (IHTVariable{T,M}, data_structures.jl:4) ----
mutable struct HipSnpLinAlg{T} <: AbstractMatrix{T}
    handle::Ptr{Cvoid}
    n::Int
    p::Int
    center::Bool
    scale::Bool
    impute::Bool
end

function HipSnpLinAlg{Float64}(s::SnpArray; center::Bool=false, scale::Bool=false,
                               impute::Bool=true, device::Integer=0)
    n, p = size(s)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    # s.data is the mmapped .bed body: ceil(n/4) x p UInt8, column-major = p columns of stride bytes
    check(ccall((:mih_snp_create, LIB), Cint,
        (Ptr{UInt8}, Int64, Int64, Int64, Cint, Cint, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
        s.data, n, p, size(s.data, 1), center, scale, impute, 64, device, h))
    x = HipSnpLinAlg{Float64}(h[], n, p, center, scale, impute)
    finalizer(x -> ccall((:mih_mat_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), x)
    return x
end
Base.size(x::HipSnpLinAlg) = (x.n, x.p)

# mul!(out, Transpose(x), r)  (call site utilities.jl:133)
function LinearAlgebra.mul!(out::Vector{Float64}, xt::Transpose{Float64, HipSnpLinAlg{Float64}},
                            r::Vector{Float64})
    check(ccall((:mih_xtv, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), xt.parent.handle, r, out))
    return out
end

# ---- parameter / result structs: field order = include/mendeliht_hip.h ----------------------
struct MihFitParams
    k::Int64; J::Int64; dist::Int32; link::Int32; nb_r::Float64; tol::Float64
    max_iter::Int32; min_iter::Int32; max_step::Int32; est_r::Int32
    zkeep::Ptr{UInt8}; weight::Ptr{Float64}; group::Ptr{Int64}; ks::Ptr{Int64}; nks::Int64
    progress::Ptr{Cvoid}; progress_user::Ptr{Cvoid}; init_beta::Int32
    comm::Ptr{Cvoid}      # Ptr{MihComm} for a column-sharded fit, C_NULL otherwise
    debias::Int32
end
# mih_comm: exchange callbacks of a column-sharded fit (one Julia process per GPU, e.g. under mpiexec)
struct MihComm
    rank::Int32; world::Int32; col_offset::Int64; p_global::Int64
    allreduce::Ptr{Cvoid}; allgather::Ptr{Cvoid}; user::Ptr{Cvoid}
end
mutable struct MihFitResult
    time::Float64; logl::Float64; iter::Int64; pve::Float64; nb_r::Float64
    choose_fired::Int32; n_trace::Int32
    beta::Ptr{Float64}; c::Ptr{Float64}; logl_trace::Ptr{Float64}; tol_trace::Ptr{Float64}
    bt_trace::Ptr{Int32}; mu::Ptr{Float64}
end

# refuse to run against a library whose structs differ from the mirrors above (mih_abi_sizes)
function __init__()
    sz = zeros(Int64, 4)
    check(ccall((:mih_abi_sizes, LIB), Cint, (Ptr{Int64}, Int32), sz, 4))
    (sz[1] == sizeof(MihFitParams) && sz[2] == sizeof(MihFitResult) && sz[4] == sizeof(MihComm)) ||
        error("MendelIHTHip.jl struct mirrors do not match $LIB: $sz")
end

distcode(::Normal) = Int32(0); distcode(::Bernoulli) = Int32(1)
distcode(::Poisson) = Int32(2); distcode(::NegativeBinomial) = Int32(3)
distcode(::Gamma) = Int32(4); distcode(::InverseGaussian) = Int32(5)
linkcode(::IdentityLink) = Int32(0); linkcode(::LogitLink) = Int32(1); linkcode(::LogLink) = Int32(2)
linkcode(::ProbitLink) = Int32(3); linkcode(::CloglogLink) = Int32(4); linkcode(::CauchitLink) = Int32(5)
linkcode(::InverseLink) = Int32(6); linkcode(::InverseSquareLink) = Int32(7); linkcode(::SqrtLink) = Int32(8)

# fit_iht(y, x::HipSnpLinAlg, z; ...)  -- same keywords as src/fit.jl:60-82
function fit_iht(y::AbstractVector{Float64}, x::HipSnpLinAlg{Float64}, z::AbstractVecOrMat{Float64};
        k::Int=10, J::Int=1, d::UnivariateDistribution=Normal(), l::Link=IdentityLink(),
        weight::AbstractVector{Float64}=Float64[], zkeep::BitVector=trues(size(z, 2)),
        verbose::Bool=true, tol::Float64=1e-4, max_iter::Int=200, min_iter::Int=5, max_step::Int=3,
        kwargs...)
    x.center || error("x is not centered! Please construct SnpLinAlg{Float64}(::SnpArray, center=true, scale=true)")
    q = size(z, 2)
    zk = Vector{UInt8}(zkeep)
    beta = zeros(x.p); c = zeros(q)
    lt = zeros(max_iter); tt = zeros(max_iter); bt = zeros(Int32, max_iter)
    GC.@preserve zk weight beta c lt tt bt begin
        prm = MihFitParams(k, J, distcode(d), linkcode(l), d isa NegativeBinomial ? d.r : 1.0, tol,
            max_iter, min_iter, max_step, 0, pointer(zk),
            isempty(weight) ? Ptr{Float64}(C_NULL) : pointer(weight), C_NULL, C_NULL, 0, C_NULL, C_NULL,
            Int32(get(kwargs, :init_beta, false)), get(kwargs, :comm, C_NULL), Int32(get(kwargs, :debias, false)))
        res = MihFitResult(0, 0, 0, 0, 0, 0, 0, pointer(beta), pointer(c), pointer(lt), pointer(tt),
            pointer(bt), C_NULL)
        check(ccall((:mih_fit_iht, LIB), Cint,
            (Ptr{Cvoid}, Ref{MihFitParams}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{UInt8}, Ref{MihFitResult}),
            x.handle, prm, y, z, q, C_NULL, res))
        if verbose
            for i in 1:res.n_trace
                println("Iteration $i: loglikelihood = $(lt[i]), backtracks = $(bt[i]), tol = $(tt[i])")
            end
        end
        return IHTResult(res.time, res.logl, res.iter, beta, c, J, k, Int[], d, res.pve)
    end
end

# cv_iht(y, x::HipSnpLinAlg, z; path, q, folds, ...)  -- src/cross_validation.jl:60-79.
# rank/world select this process's share of the (fold,k) grid; combine with one MPI/RCCL sum.
function cv_iht(y::AbstractVector{Float64}, x::HipSnpLinAlg{Float64}, z::AbstractVecOrMat{Float64};
        d::UnivariateDistribution=Normal(), l::Link=IdentityLink(), path::AbstractVector{<:Integer}=1:20,
        q::Int=5, folds::AbstractVector{Int}=rand(1:q, size(x, 1)), max_iter::Int=100, min_iter::Int=5,
        rank::Int=0, world::Int=1, kwargs...)
    nz = size(z, 2)
    f32 = Vector{Int32}(folds); pth = Vector{Int64}(path)
    raw = zeros(q * length(pth)); mse = zeros(length(pth))
    prm = MihFitParams(1, 1, distcode(d), linkcode(l), d isa NegativeBinomial ? d.r : 1.0, 1e-4,
        max_iter, min_iter, 3, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0, C_NULL, C_NULL, Int32(0), C_NULL, Int32(get(kwargs, :debias, false)))
    check(ccall((:mih_cv_iht, LIB), Cint,
        (Ptr{Cvoid}, Ref{MihFitParams}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, Int32, Ptr{Int64},
         Int64, Int32, Int32, Ptr{Float64}),
        x.handle, prm, y, z, nz, f32, q, pth, length(pth), rank, world, raw))
    # world > 1: raw = MPI.Allreduce(raw, +, comm)  (each rank filled only its own combinations)
    check(ccall((:mih_cv_meanloss, LIB), Cint, (Ptr{Float64}, Ptr{Int32}, Int64, Int32, Int64, Ptr{Float64}),
        raw, f32, length(folds), q, length(pth), mse))
    return mse
end

# cv_iht over several GPUs from this one process: xs[g] is a replica of the matrix on GPU g-1
function cv_iht(y::AbstractVector{Float64}, xs::Vector{HipSnpLinAlg{Float64}}, z::AbstractVecOrMat{Float64};
        d::UnivariateDistribution=Normal(), l::Link=IdentityLink(), path::AbstractVector{<:Integer}=1:20,
        q::Int=5, folds::AbstractVector{Int}=rand(1:q, size(xs[1], 1)), max_iter::Int=100, min_iter::Int=5, kwargs...)
    nz = size(z, 2)
    f32 = Vector{Int32}(folds); pth = Vector{Int64}(path)
    raw = zeros(q * length(pth)); mse = zeros(length(pth))
    hs = [x.handle for x in xs]
    prm = MihFitParams(1, 1, distcode(d), linkcode(l), d isa NegativeBinomial ? d.r : 1.0, 1e-4,
        max_iter, min_iter, 3, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0, C_NULL, C_NULL, Int32(0), C_NULL,
        Int32(get(kwargs, :debias, false)))
    GC.@preserve xs check(ccall((:mih_cv_iht_multi, LIB), Cint,
        (Ptr{Ptr{Cvoid}}, Int32, Ref{MihFitParams}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int32}, Int32, Ptr{Int64},
         Int64, Ptr{Float64}),
        hs, length(hs), prm, y, z, nz, f32, q, pth, length(pth), raw))
    check(ccall((:mih_cv_meanloss, LIB), Cint, (Ptr{Float64}, Ptr{Int32}, Int64, Int32, Int64, Ptr{Float64}),
        raw, f32, length(folds), q, length(pth), mse))
    return mse
end

end # module
