# ESparseHIP.jl -- the reference-side binding of libesparse_hip.so (see INTEGRATION.md).
#
# A maintainer of ExtendableSparse.jl adds this file next to src/matrix/sparsematrixlnk.jl and
# `include`s it from src/ExtendableSparse.jl after line 32.  It plugs the device buffer into the
# package's own extension slot (src/matrix/abstractsparsematrixextension.jl:6-14): nothing else in
# the package changes.  NOT executed in the build container (no Julia there); kept short.

const libesparse = get(ENV, "ESPARSE_HIP_LIB", "libesparse_hip.so")
const ESP_SET, ESP_UPDATE, ESP_RAWUPDATE = Int32(0), Int32(1), Int32(2)
const ESP_FLUSH_PLUS = Int32(1)
const ESP_CHUNK = 1 << 20

function esp_check(h, rc::Int32)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:esp_last_error, libesparse), Cstring, (Ptr{Cvoid},), h))
    rc == -2 && throw(BoundsError())                   # sparsematrixcsc.jl:8-10
    error("esparse_hip error $rc: $msg")
end

"""
Device-resident COO append buffer replacing `SparseMatrixLNK` (Float64 / Int64 only; every other
`Tv`/`Ti` stays on the CPU buffers).  Updates are staged in a pinned host chunk owned by the
library and committed with one `ccall` per chunk.
"""
mutable struct SparseMatrixHIPCOO{Tv, Ti <: Integer} <: AbstractSparseMatrixExtension{Tv, Ti}
    m::Ti
    n::Ti
    handle::Ptr{Cvoid}
    rows::Vector{Int64}      # views of the pinned staging chunk
    cols::Vector{Int64}
    vals::Vector{Float64}
    kinds::Vector{UInt8}
    nstaged::Int
end

function SparseMatrixHIPCOO{Float64, Int64}(m, n; device = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(C_NULL, ccall((:esp_create, libesparse), Int32, (Int64, Int64, Int32, Int64, Ptr{Ptr{Cvoid}}),
                            m, n, device, 0, h))
    r, c, v, k, got = Ref{Ptr{Int64}}(), Ref{Ptr{Int64}}(), Ref{Ptr{Float64}}(), Ref{Ptr{UInt8}}(), Ref{Int64}()
    esp_check(h[], ccall((:esp_stage_begin, libesparse), Int32,
                         (Ptr{Cvoid}, Int64, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt8}}, Ptr{Int64}),
                         h[], ESP_CHUNK, r, c, v, k, got))
    x = SparseMatrixHIPCOO{Float64, Int64}(m, n, h[], unsafe_wrap(Array, r[], got[]), unsafe_wrap(Array, c[], got[]),
                                           unsafe_wrap(Array, v[], got[]), unsafe_wrap(Array, k[], got[]), 0)
    finalizer(y -> ccall((:esp_destroy, libesparse), Int32, (Ptr{Cvoid},), y.handle), x)
end

Base.size(x::SparseMatrixHIPCOO) = (x.m, x.n)

function commit!(x::SparseMatrixHIPCOO)
    x.nstaged == 0 && return x
    n, x.nstaged = x.nstaged, 0
    esp_check(x.handle, ccall((:esp_commit, libesparse), Int32, (Ptr{Cvoid}, Int64, Int32, Int32), x.handle, n, -1, 0))
    x
end

@inline function push_entry!(x::SparseMatrixHIPCOO, kind, v, i, j)
    (1 <= i <= x.m) & (1 <= j <= x.n) || throw(BoundsError(x, (i, j)))
    k = (x.nstaged += 1)
    @inbounds x.rows[k] = i; @inbounds x.cols[k] = j; @inbounds x.vals[k] = v; @inbounds x.kinds[k] = kind
    k == length(x.rows) && commit!(x)
    x
end

# nnz(ext) > 0 iff anything is pending: the flush! gate of genericextendablesparsematrixcsc.jl:32
function SparseArrays.nnz(x::SparseMatrixHIPCOO)
    c = Ref{Int64}(0)
    esp_check(x.handle, ccall((:esp_pending, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), x.handle, c))
    c[] + x.nstaged
end

Base.setindex!(x::SparseMatrixHIPCOO, v, i, j) = push_entry!(x, ESP_SET, Float64(v), i, j)
updateindex!(x::SparseMatrixHIPCOO, ::typeof(+), v, i, j) = push_entry!(x, ESP_UPDATE, Float64(v), i, j)
updateindex!(x::SparseMatrixHIPCOO, ::typeof(-), v, i, j) = push_entry!(x, ESP_UPDATE, -Float64(v), i, j)
rawupdateindex!(x::SparseMatrixHIPCOO, ::typeof(+), v, i, j, tid = 1) = push_entry!(x, ESP_RAWUPDATE, Float64(v), i, j)
rawupdateindex!(x::SparseMatrixHIPCOO, ::typeof(-), v, i, j, tid = 1) = push_entry!(x, ESP_RAWUPDATE, -Float64(v), i, j)
# pending entries live on the device: the wrapper flushes before reading (cf. genericmt...jl:80)
Base.getindex(x::SparseMatrixHIPCOO, i, j) = error("flush! the matrix before getindex on device-pending entries")

"""
`ext + csc -> SparseMatrixCSC`: THE flush (replaces sparsematrixlnk.jl:294-383).
Uploads `csc`, runs the HIP pipeline, downloads into Julia-owned vectors.
"""
function Base.:+(x::SparseMatrixHIPCOO{Float64, Int64}, csc::SparseMatrixCSC{Float64, Int64})
    @assert size(csc) == size(x)
    commit!(x)
    h = x.handle
    esp_check(h, ccall((:esp_set_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64),
                       h, csc.colptr, csc.rowval, csc.nzval, nnz(csc)))
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(h, ccall((:esp_flush, libesparse), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int32}), h, ESP_FLUSH_PLUS, z, changed))
    colptr = Vector{Int64}(undef, x.n + 1)
    rowval = Vector{Int64}(undef, z[])
    nzval = Vector{Float64}(undef, z[])
    esp_check(h, ccall((:esp_get_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), h, colptr, rowval, nzval))
    SparseMatrixCSC{Float64, Int64}(x.m, x.n, colptr, rowval, nzval)
end
Base.:+(csc::SparseMatrixCSC, x::SparseMatrixHIPCOO) = x + csc

# Base.sum(extmatrices, csc) of the plugin contract (abstractsparsematrixextension.jl:11):
# csc + x1 + x2 + ... left to right, like sparsematrixdilnkc.jl:397-435
function Base.sum(xs::Vector{SparseMatrixHIPCOO{Tv, Ti}}, csc::SparseMatrixCSC{Tv, Ti}) where {Tv, Ti}
    for x in xs
        nnz(x) > 0 && (csc = x + csc)
    end
    csc
end

# aliases in the style of src/ExtendableSparse.jl:34-39
const HIPExtendableSparseMatrixCSC{Tv, Ti} = GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}
const MTHIPExtendableSparseMatrixCSC{Tv, Ti} = GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}

# ------------------------------------------------------------------------------------------------
# North-star form (INTEGRATION.md): same fields and methods as ExtendableSparseMatrixCSC
# (src/matrix/extendable.jl:10-25,159-272), but buffer AND CSC stay on the GPU between flushes; the
# host copy is fetched on demand.  Every update goes to the device (no host findindex); esp_flush in
# ROUTED mode applies updates of stored positions in call order (extendable.jl:164-166).
const ESP_FLUSH_ROUTED = Int32(0)
mutable struct HIPResidentSparseMatrixCSC{Tv, Ti <: Integer} <: AbstractExtendableSparseMatrixCSC{Tv, Ti}
    buf::SparseMatrixHIPCOO{Tv, Ti}                     # in the role of lnkmatrix (handle + staging chunk)
    cscmatrix::Union{SparseMatrixCSC{Tv, Ti}, Nothing}  # host copy, valid until the next update
    phash::UInt64
end
HIPResidentSparseMatrixCSC{Float64, Int64}(m, n) =
    HIPResidentSparseMatrixCSC{Float64, Int64}(SparseMatrixHIPCOO{Float64, Int64}(m, n), spzeros(Float64, Int64, m, n), 0)
Base.size(A::HIPResidentSparseMatrixCSC) = size(A.buf)
touch!(A::HIPResidentSparseMatrixCSC) = (A.cscmatrix = nothing; A)
Base.setindex!(A::HIPResidentSparseMatrixCSC, v, i::Integer, j::Integer) = (setindex!(A.buf, v, i, j); touch!(A))
updateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j) = (updateindex!(A.buf, op, v, i, j); touch!(A))
rawupdateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j, part = 1) = (rawupdateindex!(A.buf, op, v, i, j); touch!(A))

function flush!(A::HIPResidentSparseMatrixCSC)                       # extendable.jl:248-255
    commit!(A.buf)
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(A.buf.handle, ccall((:esp_flush, libesparse), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int32}),
                                  A.buf.handle, ESP_FLUSH_ROUTED, z, changed))
    if changed[] != 0                                                # the CSC was rebuilt: new pattern hash (:252)
        hsh = Ref{UInt64}(0)
        esp_check(A.buf.handle, ccall((:esp_pattern_hash, libesparse), Int32, (Ptr{Cvoid}, Ptr{UInt64}), A.buf.handle, hsh))
        A.phash = hsh[]
    end
    A
end

function SparseArrays.sparse(A::HIPResidentSparseMatrixCSC{Float64, Int64})   # extendable.jl:258-261
    flush!(A)
    A.cscmatrix === nothing || return A.cscmatrix
    h, (m, n) = A.buf.handle, size(A)
    z = Ref{Int64}(0)
    esp_check(h, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
    colptr, rowval, nzval = Vector{Int64}(undef, n + 1), Vector{Int64}(undef, z[]), Vector{Float64}(undef, z[])
    esp_check(h, ccall((:esp_get_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), h, colptr, rowval, nzval))
    A.cscmatrix = SparseMatrixCSC{Float64, Int64}(m, n, colptr, rowval, nzval)
end

# consumers that never leave the GPU (SURVEY 8f): mul! sums every row in column order, like the column loop
function LinearAlgebra.mul!(r::Vector{Float64}, A::HIPResidentSparseMatrixCSC{Float64, Int64}, x::Vector{Float64})
    flush!(A)
    esp_check(A.buf.handle, ccall((:esp_mul, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), A.buf.handle, x, r, 0))
    r
end
function mark_dirichlet(A::HIPResidentSparseMatrixCSC; penalty = 1.0e20)      # sparsematrixcsc.jl:94-108
    flush!(A)
    marker = zeros(Bool, size(A, 2))
    esp_check(A.buf.handle, ccall((:esp_mark_dirichlet, libesparse), Int32, (Ptr{Cvoid}, Float64, Ptr{Bool}, Int32), A.buf.handle, penalty, marker, 0))
    marker
end
function eliminate_dirichlet!(A::HIPResidentSparseMatrixCSC, marker::Vector{Bool})   # sparsematrixcsc.jl:121-144
    flush!(A)
    esp_check(A.buf.handle, ccall((:esp_eliminate_dirichlet, libesparse), Int32, (Ptr{Cvoid}, Ptr{Bool}, Int32), A.buf.handle, marker, 0))
    touch!(A)
end
function reset!(A::HIPResidentSparseMatrixCSC)                                 # extendable.jl:269-272 (phash kept)
    A.buf.nstaged = 0
    esp_check(A.buf.handle, ccall((:esp_reset, libesparse), Int32, (Ptr{Cvoid},), A.buf.handle))
    touch!(A)
end
