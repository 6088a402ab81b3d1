# ESparseHIP.jl -- the reference-side binding of libesparse_hip.so (see INTEGRATION.md).
#
# A maintainer of ExtendableSparse.jl adds this file next to src/matrix/sparsematrixlnk.jl and
# `include`s it from src/ExtendableSparse.jl after line 32.  It plugs the device buffer into the
# package's own extension slot (src/matrix/abstractsparsematrixextension.jl:6-14): nothing else in
# the package changes.
# UNTESTED IN THE BUILD CONTAINER: there is no Julia there.  tests/test_julia_shim.py checks every `ccall` of this file
# against include/esparse_hip.h (symbol, arity, pointer / scalar kind of every argument, return type).

const libesparse = get(ENV, "ESPARSE_HIP_LIB", "libesparse_hip.so")
const ESP_SET, ESP_UPDATE, ESP_RAWUPDATE, ESP_COO = Int32(0), Int32(1), Int32(2), Int32(3)
const ESP_FLUSH_ROUTED, ESP_FLUSH_PLUS = Int32(0), Int32(1)
const ESP_CHUNK = 1 << 16        # staged updates per ccall (pinned; allocated on the first push)

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
    rows::Vector{Int64}      # views of the pinned staging chunk (empty until the first push)
    cols::Vector{Int64}
    vals::Vector{Float64}
    kinds::Vector{UInt8}
    nstaged::Int
    released::Bool           # consumed by a flush! (plus_consume!): any further use is an error, not a silent empty buffer
    # the host matrix whose PATTERN the handle's device CSC holds (the colptr / rowval vectors the last download handed out):
    # while ext.cscmatrix still carries these very vectors only nzval travels (esp_set_nzval / esp_get_nzval)
    mirror_colptr::Vector{Int64}
    mirror_rowval::Vector{Int64}
end
check_live(x) = x.released && error("SparseMatrixHIPCOO: the buffer was consumed by an earlier flush!")

function wrap_handle(m, n, h::Ptr{Cvoid})
    x = SparseMatrixHIPCOO{Float64, Int64}(m, n, h, Int64[], Int64[], Float64[], UInt8[], 0, false, Int64[], Int64[])
    finalizer(y -> (y.handle == C_NULL || ccall((:esp_destroy, libesparse), Int32, (Ptr{Cvoid},), y.handle); y.handle = C_NULL), x)
end

function SparseMatrixHIPCOO{Float64, Int64}(m, n; device = 0, capacity_hint = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(C_NULL, ccall((:esp_create, libesparse), Int32, (Int64, Int64, Int32, Int64, Ptr{Ptr{Cvoid}}),
                            m, n, device, capacity_hint, h))
    wrap_handle(m, n, h[])
end

Base.size(x::SparseMatrixHIPCOO) = (x.m, x.n)

# the pinned staging chunk: asked for on the first update (a buffer that a Generic wrapper creates after every flush!
# and never fills costs no pinned memory)
function stage!(x::SparseMatrixHIPCOO)
    r, c, v, k, got = Ref{Ptr{Int64}}(), Ref{Ptr{Int64}}(), Ref{Ptr{Float64}}(), Ref{Ptr{UInt8}}(), Ref{Int64}()
    esp_check(x.handle, ccall((:esp_stage_begin, libesparse), Int32,
                              (Ptr{Cvoid}, Int64, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt8}}, Ptr{Int64}),
                              x.handle, ESP_CHUNK, r, c, v, k, got))
    x.rows, x.cols = unsafe_wrap(Array, r[], got[]), unsafe_wrap(Array, c[], got[])
    x.vals, x.kinds = unsafe_wrap(Array, v[], got[]), unsafe_wrap(Array, k[], got[])
    x
end

function commit!(x::SparseMatrixHIPCOO)
    check_live(x)
    x.nstaged == 0 && return x
    n, x.nstaged = x.nstaged, 0
    esp_check(x.handle, ccall((:esp_commit, libesparse), Int32, (Ptr{Cvoid}, Int64, Int32, Int32), x.handle, n, -1, 0))
    x
end

@inline function push_entry!(x::SparseMatrixHIPCOO, kind, v, i, j)
    check_live(x)
    (1 <= i <= x.m) & (1 <= j <= x.n) || throw(BoundsError(x, (i, j)))
    isempty(x.rows) && stage!(x)
    k = (x.nstaged += 1)
    @inbounds x.rows[k] = i; @inbounds x.cols[k] = j; @inbounds x.vals[k] = v; @inbounds x.kinds[k] = kind
    k == length(x.rows) && commit!(x)
    x
end

# free the device and pinned memory NOW: the GC does not see it, and the Generic wrappers drop a buffer per flush!
function release!(x::SparseMatrixHIPCOO)
    x.nstaged = 0
    x.rows, x.cols, x.vals, x.kinds = Int64[], Int64[], Float64[], UInt8[]      # the chunk pointers die with the buffers
    esp_check(x.handle, ccall((:esp_release_buffers, libesparse), Int32, (Ptr{Cvoid},), x.handle))
    x
end

# nnz(ext) > 0 iff anything is pending: the flush! gate of genericextendablesparsematrixcsc.jl:32
function SparseArrays.nnz(x::SparseMatrixHIPCOO)
    check_live(x)
    c = Ref{Int64}(0)
    esp_check(x.handle, ccall((:esp_pending, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), x.handle, c))
    c[] + x.nstaged
end

Base.setindex!(x::SparseMatrixHIPCOO, v, i, j) = push_entry!(x, ESP_SET, Float64(v), i, j)
updateindex!(x::SparseMatrixHIPCOO, ::typeof(+), v, i, j) = push_entry!(x, ESP_UPDATE, Float64(v), i, j)
updateindex!(x::SparseMatrixHIPCOO, ::typeof(-), v, i, j) = push_entry!(x, ESP_UPDATE, -Float64(v), i, j)
rawupdateindex!(x::SparseMatrixHIPCOO, ::typeof(+), v, i, j, tid = 1) = push_entry!(x, ESP_RAWUPDATE, Float64(v), i, j)
rawupdateindex!(x::SparseMatrixHIPCOO, ::typeof(-), v, i, j, tid = 1) = push_entry!(x, ESP_RAWUPDATE, -Float64(v), i, j)

# The assembly loop of test/femtools.jl:61-69 as ONE call: for every cell and local row il the optional diag[il, icell]
# on (i, i), then elmat[il, jl, icell] on (i, cellnodes[jl, icell]) for every jl -- bit for bit the per-entry
# rawupdateindex!(A, +, ...) calls in that order.  cellnodes = grid[CellNodes] (nloc x ncells), elmat[:, :, icell] =
# vol * S of femtools.jl:67, diag[il, icell] = the mass term of femtools.jl:64 (or `nothing`).
function assemble_elements!(x::SparseMatrixHIPCOO{Float64, Int64}, cellnodes::Matrix{Int64}, elmat::Array{Float64, 3},
                            diag::Union{Matrix{Float64}, Nothing} = nothing; kind = ESP_RAWUPDATE)
    commit!(x)
    nloc, ncells = size(cellnodes)
    @assert size(elmat) == (nloc, nloc, ncells) && (diag === nothing || size(diag) == (nloc, ncells))
    esp_check(x.handle, ccall((:esp_append_elements_host, libesparse), Int32,
                              (Ptr{Cvoid}, Int32, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32),
                              x.handle, nloc, ncells, cellnodes, elmat, diag === nothing ? C_NULL : diag, kind, 0))
    x
end

# A time step of the same mesh (new element matrices over the connectivity of the last assemble_elements!): keep_plan!(x)
# once, then assemble_elements_again!(x, elmat, diag) per step -- no pass over the connectivity, no item partition
keep_plan!(x::SparseMatrixHIPCOO, on = true) =
    (esp_check(x.handle, ccall((:esp_elements_keep_plan, libesparse), Int32, (Ptr{Cvoid}, Int32), x.handle, on ? 1 : 0)); x)
function assemble_elements_again!(x::SparseMatrixHIPCOO{Float64, Int64}, elmat::Array{Float64, 3}, diag::Union{Matrix{Float64}, Nothing} = nothing;
                                  kind = ESP_RAWUPDATE)
    commit!(x)
    esp_check(x.handle, ccall((:esp_append_elements_again_host, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Int32),
                              x.handle, elmat, diag === nothing ? C_NULL : diag, kind, 0))
    x
end

# getindex(buffer,i,j) (sparsematrixlnk.jl:151-171; reached from genericextendablesparsematrixcsc.jl:57-66 for
# positions not yet in the CSC): the ordered fold of the pending calls at (i,j), on the device.  Slow path by design.
function Base.getindex(x::SparseMatrixHIPCOO, i::Integer, j::Integer)
    commit!(x)
    v, found = Ref{Float64}(0.0), Ref{Int32}(0)
    esp_check(x.handle, ccall((:esp_pending_getindex, libesparse), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Int32}),
                              x.handle, i, j, v, found))
    v[]
end

# Base.copy(buffer): same pending entries on a handle of its own (esp_clone, device-to-device)
function Base.copy(x::SparseMatrixHIPCOO{Float64, Int64})
    commit!(x)
    h2 = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(x.handle, ccall((:esp_clone, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), x.handle, h2))
    wrap_handle(x.m, x.n, h2[])
end

"""
`ext + csc -> SparseMatrixCSC`: THE flush (replaces sparsematrixlnk.jl:294-383).
Uploads `csc`, runs the HIP pipeline, downloads into Julia-owned vectors.  Like `lnk + csc` it has NO side effect on
`x`: it works on a clone (`copy(x)`, device-to-device), so `x + csc` may be evaluated again.  The wrappers' `flush!`
and `Base.sum` consume their buffers instead (`plus_consume!`).
"""
function Base.:+(x::SparseMatrixHIPCOO{Float64, Int64}, csc::SparseMatrixCSC{Float64, Int64})
    check_live(x)
    y = copy(x)                      # esp_clone: the pending entries, device-to-device; x itself is left alone
    out = plus_consume!(y, csc)
    out
end

# The device CSC of handle h := csc.  Values only (8 instead of 24 bytes per entry) when h still holds csc's pattern: the
# Generic wrappers edit cscmatrix.nzval in place on the host (genericextendablesparsematrixcsc.jl:44-54, and users write
# nonzeros(A) .= 0: test_parallel.jl:71-92), which nobody can see from outside -- so the values always travel.
function upload_csc!(x::SparseMatrixHIPCOO, csc::SparseMatrixCSC{Float64, Int64})
    h = x.handle
    z = Ref{Int64}(0)
    esp_check(h, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
    if csc.colptr === x.mirror_colptr && csc.rowval === x.mirror_rowval && z[] == nnz(csc)
        esp_check(h, ccall((:esp_set_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, csc.nzval))
    else
        esp_check(h, ccall((:esp_set_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64),
                           h, csc.colptr, csc.rowval, csc.nzval, nnz(csc)))
    end
    x
end
# ... and back: a flush that added no position returns csc's own pattern vectors with fresh values (esp_get_nzval)
function download_csc!(x::SparseMatrixHIPCOO, csc::SparseMatrixCSC{Float64, Int64}, z::Int64, changed::Bool)
    h = x.handle
    if !changed && z == nnz(csc)
        nzval = Vector{Float64}(undef, z)
        esp_check(h, ccall((:esp_get_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, nzval))
        out = SparseMatrixCSC{Float64, Int64}(x.m, x.n, csc.colptr, csc.rowval, nzval)
    else
        colptr, rowval, nzval = Vector{Int64}(undef, x.n + 1), Vector{Int64}(undef, z), Vector{Float64}(undef, z)
        esp_check(h, ccall((:esp_get_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), h, colptr, rowval, nzval))
        out = SparseMatrixCSC{Float64, Int64}(x.m, x.n, colptr, rowval, nzval)
    end
    x.mirror_colptr, x.mirror_rowval = out.colptr, out.rowval
    out
end

# csc + buffer where the buffer is consumed: its pending entries are folded into the result.  What flush! of the Generic
# wrappers needs: they drop the buffer right after `+` (genericextendablesparsematrixcsc.jl:31-37), see the flush! methods
# below.  keep = true: the handle (device CSC + scratch) lives on in the wrapper's NEXT buffer (adopt!); else its device
# memory is released at once (not at some later GC, which does not see device memory).
function plus_consume!(x::SparseMatrixHIPCOO{Float64, Int64}, csc::SparseMatrixCSC{Float64, Int64}; keep = false)
    @assert size(csc) == size(x)
    commit!(x)
    h = x.handle
    upload_csc!(x, csc)
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(h, ccall((:esp_flush, libesparse), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int32}), h, ESP_FLUSH_PLUS, z, changed))
    out = download_csc!(x, csc, z[], changed[] != 0)
    if !keep
        release!(x)                  # pending buffers, scratch, the device copy of the CSC: gone now
        x.released = true
    end
    out
end

# the wrapper's next buffer takes over the handle of the one it drops: the device CSC (and every scratch buffer) stays
# where it is between flushes; the old struct is left without a handle (its finalizer has nothing to do)
function adopt!(old::SparseMatrixHIPCOO{Float64, Int64})
    new = SparseMatrixHIPCOO{Float64, Int64}(old.m, old.n, old.handle, old.rows, old.cols, old.vals, old.kinds, 0, false,
                                             old.mirror_colptr, old.mirror_rowval)
    old.handle, old.released = C_NULL, true
    old.rows, old.cols, old.vals, old.kinds = Int64[], Int64[], Float64[], UInt8[]
    finalizer(y -> (y.handle == C_NULL || ccall((:esp_destroy, libesparse), Int32, (Ptr{Cvoid},), y.handle); y.handle = C_NULL), new)
end

# flush! of the Generic wrappers with a HIP buffer: the reference's `ext.cscmatrix = ext.xmatrix + ext.cscmatrix;
# ext.xmatrix = Tm(m, n)` (genericextendablesparsematrixcsc.jl:31-37) with the buffer consumed instead of copied and the
# device CSC kept attached: per flush! the values go up, and the values (no new position) or the matrix come back
function flush!(ext::GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Float64, Int64}, Float64, Int64})
    if nnz(ext.xmatrix) > 0
        ext.cscmatrix = plus_consume!(ext.xmatrix, ext.cscmatrix; keep = true)
        ext.xmatrix = adopt!(ext.xmatrix)
    end
    ext
end
Base.:+(csc::SparseMatrixCSC, x::SparseMatrixHIPCOO) = x + csc

# Base.sum(extmatrices, csc) of the plugin contract (abstractsparsematrixextension.jl:11): sparse!(I, J, V, m, n, +) over
# (csc entries, x1's, x2's, ...), i.e. ((csc + x1) + x2) + ... left to right (sparsematrixdilnkc.jl:397-435), as ONE
# device call: every buffer folds by itself, the folds meet the stored matrix in one flush (esp_flush_sum) -- the CSC
# travels once whatever np is (test_parallel.jl:41,74: 10, 15, 20).  home: the handle that keeps the CSC; the buffers
# come back EMPTY (flush! of the MT wrapper replaces them all right afterwards, genericmtextendablesparsematrixcsc.jl:47-49).
function sum_into!(home::SparseMatrixHIPCOO{Float64, Int64}, xs::Vector{SparseMatrixHIPCOO{Float64, Int64}}, csc::SparseMatrixCSC{Float64, Int64})
    foreach(commit!, xs)
    sum(nnz, xs) == 0 && return csc
    upload_csc!(home, csc)
    handles = Ptr{Cvoid}[x.handle for x in xs]
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(home.handle, ccall((:esp_flush_sum, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32, Ptr{Int64}, Ptr{Int32}),
                                 home.handle, handles, length(xs), z, changed))
    download_csc!(home, csc, z[], changed[] != 0)
end
function Base.sum(xs::Vector{SparseMatrixHIPCOO{Float64, Int64}}, csc::SparseMatrixCSC{Float64, Int64})
    home = SparseMatrixHIPCOO{Float64, Int64}(size(csc)...)
    out = sum_into!(home, xs, csc)
    release!(home)                   # (a one-off destination: its device memory goes now, not at some later GC)
    home.released = true
    out
end
# flush! of the MT wrapper with HIP buffers (genericmtextendablesparsematrixcsc.jl:45-51): the destination handle lives
# as long as the wrapper does (one per wrapper, found through its buffer vector's first element ... kept in a WeakKeyDict)
const HOME_OF = WeakKeyDict{Any, SparseMatrixHIPCOO{Float64, Int64}}()
function flush!(ext::GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Float64, Int64}, Float64, Int64})
    home = get!(() -> SparseMatrixHIPCOO{Float64, Int64}(size(ext.cscmatrix)...), HOME_OF, ext)
    ext.cscmatrix = sum_into!(home, ext.xmatrices, ext.cscmatrix)
    ext                              # (the buffers are empty T_ext(m, n) again: they are kept)
end

# aliases in the style of src/ExtendableSparse.jl:34-39
const HIPExtendableSparseMatrixCSC{Tv, Ti} = GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}
const MTHIPExtendableSparseMatrixCSC{Tv, Ti} = GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}

# ------------------------------------------------------------------------------------------------
# North-star form (INTEGRATION.md): same fields and methods as ExtendableSparseMatrixCSC
# (src/matrix/extendable.jl:10-25,159-272), but buffer AND CSC stay on the GPU between flushes; the
# host copy is fetched on demand.  Every update goes to the device (no host findindex); esp_flush in
# ROUTED mode applies updates of stored positions in call order (extendable.jl:164-166).
mutable struct HIPResidentSparseMatrixCSC{Tv, Ti <: Integer} <: AbstractExtendableSparseMatrixCSC{Tv, Ti}
    buf::SparseMatrixHIPCOO{Tv, Ti}                     # in the role of lnkmatrix (handle + staging chunk)
    cscmatrix::Union{SparseMatrixCSC{Tv, Ti}, Nothing}  # host copy, valid until the next update
    phash::UInt64
end
HIPResidentSparseMatrixCSC{Float64, Int64}(m, n; kwargs...) =
    HIPResidentSparseMatrixCSC{Float64, Int64}(SparseMatrixHIPCOO{Float64, Int64}(m, n; kwargs...), spzeros(Float64, Int64, m, n), 0)
Base.size(A::HIPResidentSparseMatrixCSC) = size(A.buf)
touch!(A::HIPResidentSparseMatrixCSC) = (A.cscmatrix = nothing; A)
Base.setindex!(A::HIPResidentSparseMatrixCSC, v, i::Integer, j::Integer) = (setindex!(A.buf, v, i, j); touch!(A))
assemble_elements!(A::HIPResidentSparseMatrixCSC, cellnodes, elmat, diag = nothing; kwargs...) =
    (assemble_elements!(A.buf, cellnodes, elmat, diag; kwargs...); touch!(A))
keep_plan!(A::HIPResidentSparseMatrixCSC, on = true) = (keep_plan!(A.buf, on); A)
assemble_elements_again!(A::HIPResidentSparseMatrixCSC, elmat, diag = nothing; kwargs...) =
    (assemble_elements_again!(A.buf, elmat, diag; kwargs...); touch!(A))
updateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j) = (updateindex!(A.buf, op, v, i, j); touch!(A))
rawupdateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j, part = 1) = (rawupdateindex!(A.buf, op, v, i, j); touch!(A))

# ExtendableSparseMatrix(I, J, V[, m, n]) (extendable.jl:85-104) = sparse(I,J,V,m,n,+): the triplets go through the
# device pipeline as COO entries (first value as it is, duplicates added in input order, numerical zeros kept)
function HIPResidentSparseMatrixCSC(I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64}, m = maximum(I), n = maximum(J))
    A = HIPResidentSparseMatrixCSC{Float64, Int64}(m, n; capacity_hint = length(I))
    esp_check(A.buf.handle, ccall((:esp_append_host, libesparse), Int32,
                                  (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{UInt8}, Int32, Int32, Int64),
                                  A.buf.handle, I, J, V, C_NULL, ESP_COO, 0, length(I)))
    flush!(touch!(A))
end

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

# getindex (extendable.jl:226-238): pending entries live on the device, so the lookup is flush! + findindex on the
# device CSC (`A[i,j] += v` works; assembly loops should call updateindex!, which is bitwise the same update)
function Base.getindex(A::HIPResidentSparseMatrixCSC, i::Integer, j::Integer)
    flush!(A)
    v, found = Ref{Float64}(0.0), Ref{Int32}(0)
    esp_check(A.buf.handle, ccall((:esp_getindex, libesparse), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Int32}),
                                  A.buf.handle, i, j, v, found))
    v[]
end

function SparseArrays.nnz(A::HIPResidentSparseMatrixCSC)                       # abstractextendablesparsematrixcsc.jl:80
    flush!(A)
    z = Ref{Int64}(0)
    esp_check(A.buf.handle, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), A.buf.handle, z))
    z[]
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

# dropzeros!(ext) (abstractextendablesparsematrixcsc.jl:282) and fdrand!'s zero!(A) = nonzeros(A) .= 0 (sprand.jl:82)
function SparseArrays.dropzeros!(A::HIPResidentSparseMatrixCSC)
    flush!(A)
    z = Ref{Int64}(0)
    esp_check(A.buf.handle, ccall((:esp_dropzeros, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), A.buf.handle, z))
    touch!(A)
end
function zero!(A::HIPResidentSparseMatrixCSC)
    flush!(A)
    esp_check(A.buf.handle, ccall((:esp_zero_values, libesparse), Int32, (Ptr{Cvoid},), A.buf.handle))
    touch!(A)
end

# Base.copy(ext) (extendable.jl:279-285): CSC, pending entries and phash; device-to-device
function Base.copy(A::HIPResidentSparseMatrixCSC{Float64, Int64})
    commit!(A.buf)
    h2 = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(A.buf.handle, ccall((:esp_clone, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), A.buf.handle, h2))
    HIPResidentSparseMatrixCSC{Float64, Int64}(wrap_handle(A.buf.m, A.buf.n, h2[]), A.cscmatrix === nothing ? nothing : copy(A.cscmatrix), A.phash)
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
# set-up of the point preconditioners on the device CSC (factorizations/jacobi.jl:5-20, ilu0.jl:8-41)
function jacobi_setup(A::HIPResidentSparseMatrixCSC{Float64, Int64})
    flush!(A)
    invdiag = Vector{Float64}(undef, size(A, 2))
    esp_check(A.buf.handle, ccall((:esp_jacobi_setup, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), A.buf.handle, invdiag, 0))
    invdiag
end
function ilu0_setup(A::HIPResidentSparseMatrixCSC{Float64, Int64})
    flush!(A)
    n = size(A, 2)
    xdiag, idiag = Vector{Float64}(undef, n), Vector{Int64}(undef, n)
    esp_check(A.buf.handle, ccall((:esp_ilu0_setup, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int32), A.buf.handle, xdiag, idiag, 0))
    xdiag, idiag
end
function reset!(A::HIPResidentSparseMatrixCSC)                                 # extendable.jl:269-272 (phash kept)
    A.buf.nstaged = 0
    esp_check(A.buf.handle, ccall((:esp_reset, libesparse), Int32, (Ptr{Cvoid},), A.buf.handle))
    touch!(A)
end

# ------------------------------------------------------------------------------------------------
# Column shards across the GPUs of a node, one Julia process per GPU (INTEGRATION.md section 3): what
# GenericMTExtendableSparseMatrixCSC does with one buffer per thread (genericmt...jl:45-51,87-99), across processes.
# `id` = esp_group_unique_id() made on rank 0 and broadcast by the host (MPI.Bcast!, Distributed).
function esp_group_unique_id()
    id = Vector{UInt8}(undef, 128)
    esp_check(C_NULL, ccall((:esp_group_unique_id, libesparse), Int32, (Ptr{UInt8},), id))
    id
end
mutable struct HIPShardedSparseMatrixCSC
    A::HIPResidentSparseMatrixCSC{Float64, Int64}       # this rank's shard: append ANY (i,j) to it
    group::Ptr{Cvoid}
end
function HIPShardedSparseMatrixCSC(m, n, nranks, rank, id::Vector{UInt8}; device = rank)
    A = HIPResidentSparseMatrixCSC{Float64, Int64}(m, n; device = device)
    g = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(A.buf.handle, ccall((:esp_group_create, libesparse), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ptr{Ptr{Cvoid}}),
                                  A.buf.handle, nranks, rank, id, g))
    S = HIPShardedSparseMatrixCSC(A, g[])
    finalizer(s -> (s.group == C_NULL || ccall((:esp_group_destroy, libesparse), Int32, (Ptr{Cvoid},), s.group); s.group = C_NULL), S)
end
updateindex!(S::HIPShardedSparseMatrixCSC, op, v, i, j) = updateindex!(S.A, op, v, i, j)
rawupdateindex!(S::HIPShardedSparseMatrixCSC, op, v, i, j, tid = 1) = rawupdateindex!(S.A, op, v, i, j)
function flush!(S::HIPShardedSparseMatrixCSC)            # COLLECTIVE: all-to-all-v entry routing (RCCL) + local flush
    commit!(S.A.buf)
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(S.A.buf.handle, ccall((:esp_group_flush, libesparse), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int32}),
                                    S.group, ESP_FLUSH_ROUTED, z, changed))
    touch!(S.A)
    S
end
function SparseArrays.nnz(S::HIPShardedSparseMatrixCSC)  # global nnz (collective on first use after a flush)
    tot, before = Ref{Int64}(0), Ref{Int64}(0)
    esp_check(S.A.buf.handle, ccall((:esp_group_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), S.group, tot, before))
    tot[]
end
# this rank's columns col_lo:col_hi of the global CSC: (col_lo, col_hi, colptr[col_lo:col_hi+1], rowval, nzval)
function local_csc(S::HIPShardedSparseMatrixCSC)
    lo, hi, z = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    esp_check(S.A.buf.handle, ccall((:esp_group_column_range, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), S.group, lo, hi))
    esp_check(S.A.buf.handle, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), S.A.buf.handle, z))
    colptr, rowval, nzval = Vector{Int64}(undef, hi[] - lo[] + 2), Vector{Int64}(undef, z[]), Vector{Float64}(undef, z[])
    esp_check(S.A.buf.handle, ccall((:esp_group_get_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}),
                                    S.group, colptr, rowval, nzval))
    lo[], hi[], colptr, rowval, nzval
end
