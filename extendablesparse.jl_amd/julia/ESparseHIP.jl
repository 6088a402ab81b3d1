# ESparseHIP.jl -- the reference-side binding of libesparse_hip.so (see INTEGRATION.md).
#
# A maintainer of ExtendableSparse.jl adds this file next to src/matrix/sparsematrixlnk.jl and
# `include`s it from src/ExtendableSparse.jl after line 32.  It plugs the device buffer into the
# package's own extension slot (src/matrix/abstractsparsematrixextension.jl:6-14): nothing else in
# the package changes.  NOT executed in the build container (no Julia there); kept under 150 lines.

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
