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

# index types the device path takes (extendable.jl:10-25 is generic in Ti; the device CSC is Int64 inside, Int32 arrays are
# narrowed / widened beside the transfer: esp_append_host_i32, esp_set_csc_i32, esp_get_csc_i32)
const HIPIndex = Union{Int32, Int64}
# the ops the device fold knows (fold.hpp): + and, as + of the negated value, -.  Any other function: see `updateindex!` below
const DeviceOp = Union{typeof(+), typeof(-)}

"""
Device-resident COO append buffer replacing `SparseMatrixLNK` (`Tv = Float64`, `Ti` Int32 or Int64; every other
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
    mirror_colptr::Vector{Ti}
    mirror_rowval::Vector{Ti}
end
check_live(x) = x.released && error("SparseMatrixHIPCOO: the buffer was consumed by an earlier flush!")

function wrap_handle(::Type{Ti}, m, n, h::Ptr{Cvoid}) where {Ti <: HIPIndex}
    x = SparseMatrixHIPCOO{Float64, Ti}(Ti(m), Ti(n), h, Int64[], Int64[], Float64[], UInt8[], 0, false, Ti[], Ti[])
    finalizer(y -> (y.handle == C_NULL || ccall((:esp_destroy, libesparse), Int32, (Ptr{Cvoid},), y.handle); y.handle = C_NULL), x)
end

function SparseMatrixHIPCOO{Float64, Ti}(m, n; device = 0, capacity_hint = 0) where {Ti <: HIPIndex}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(C_NULL, ccall((:esp_create, libesparse), Int32, (Int64, Int64, Int32, Int64, Ptr{Ptr{Cvoid}}),
                            m, n, device, capacity_hint, h))
    wrap_handle(Ti, m, n, h[])
end

# the three transfers that carry index arrays, by index type
set_csc_call(h, csc::SparseMatrixCSC{Float64, Int64}) =
    ccall((:esp_set_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64), h, csc.colptr, csc.rowval, csc.nzval, nnz(csc))
set_csc_call(h, csc::SparseMatrixCSC{Float64, Int32}) =
    ccall((:esp_set_csc_i32, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Int64), h, csc.colptr, csc.rowval, csc.nzval, nnz(csc))
get_csc_call(h, colptr::Vector{Int64}, rowval::Vector{Int64}, nzval::Vector{Float64}) =
    ccall((:esp_get_csc, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), h, colptr, rowval, nzval)
get_csc_call(h, colptr::Vector{Int32}, rowval::Vector{Int32}, nzval::Vector{Float64}) =
    ccall((:esp_get_csc_i32, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}), h, colptr, rowval, nzval)
append_host_call(h, I::Vector{Int64}, J::Vector{Int64}, V::Vector{Float64}, kind::Int32) =
    ccall((:esp_append_host, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{UInt8}, Int32, Int32, Int64),
          h, I, J, V, C_NULL, kind, 0, length(I))
append_host_call(h, I::Vector{Int32}, J::Vector{Int32}, V::Vector{Float64}, kind::Int32) =
    ccall((:esp_append_host_i32, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{UInt8}, Int32, Int32, Int64),
          h, I, J, V, C_NULL, kind, 0, length(I))

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
# Any other `op` (sparsematrixlnk.jl:210-253 takes every function): the pending entries live on the device, where only the
# ordered + fold exists -- the plug-in buffer refuses, loudly.  HIPResidentSparseMatrixCSC (below) moves the whole matrix to
# the package's own CPU path instead (to_cpu!); with the Generic wrappers use Tm = SparseMatrixLNK for such a matrix.
unsupported_op(op) = throw(ArgumentError("SparseMatrixHIPCOO folds + and - on the device; op = $op needs the CPU buffer (SparseMatrixLNK)"))
updateindex!(x::SparseMatrixHIPCOO, op, v, i, j) = unsupported_op(op)
rawupdateindex!(x::SparseMatrixHIPCOO, op, v, i, j, tid = 1) = unsupported_op(op)

# The assembly loop of test/femtools.jl:61-69 as ONE call: for every cell and local row il the optional diag[il, icell]
# on (i, i), then elmat[il, jl, icell] on (i, cellnodes[jl, icell]) for every jl -- bit for bit the per-entry
# rawupdateindex!(A, +, ...) calls in that order.  cellnodes = grid[CellNodes] (nloc x ncells), elmat[:, :, icell] =
# vol * S of femtools.jl:67, diag[il, icell] = the mass term of femtools.jl:64 (or `nothing`).
assemble_elements!(x::SparseMatrixHIPCOO, cellnodes::Matrix{Int32}, elmat::Array{Float64, 3}, diag = nothing; kwargs...) =
    assemble_elements!(x, Matrix{Int64}(cellnodes), elmat, diag; kwargs...)      # (the element-level call reads Int64 connectivity)
function assemble_elements!(x::SparseMatrixHIPCOO, cellnodes::Matrix{Int64}, elmat::Array{Float64, 3},
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
function assemble_elements_again!(x::SparseMatrixHIPCOO, elmat::Array{Float64, 3}, diag::Union{Matrix{Float64}, Nothing} = nothing;
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
function Base.copy(x::SparseMatrixHIPCOO{Float64, Ti}) where {Ti <: HIPIndex}
    commit!(x)
    h2 = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(x.handle, ccall((:esp_clone, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), x.handle, h2))
    wrap_handle(Ti, x.m, x.n, h2[])
end

"""
`ext + csc -> SparseMatrixCSC`: THE flush (replaces sparsematrixlnk.jl:294-383).
Uploads `csc`, runs the HIP pipeline, downloads into Julia-owned vectors.  Like `lnk + csc` it has NO side effect on
`x`: it works on a clone (`copy(x)`, device-to-device), so `x + csc` may be evaluated again.  The wrappers' `flush!`
and `Base.sum` consume their buffers instead (`plus_consume!`).
"""
function Base.:+(x::SparseMatrixHIPCOO{Float64, Ti}, csc::SparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    check_live(x)
    y = copy(x)                      # esp_clone: the pending entries, device-to-device; x itself is left alone
    out = plus_consume!(y, csc)
    out
end

# The device CSC of handle h := csc.  Values only (8 instead of 24 bytes per entry) when h still holds csc's pattern: the
# Generic wrappers edit cscmatrix.nzval in place on the host (genericextendablesparsematrixcsc.jl:44-54, and users write
# nonzeros(A) .= 0: test_parallel.jl:71-92), which nobody can see from outside -- so the values always travel.
# NEVER edit colptr / rowval of a matrix this file handed out in place: the values-only path recognises the pattern by the
# identity of those two vectors (a pattern edit that keeps both vectors and nnz would go unseen); assign a new matrix instead.
function upload_csc!(x::SparseMatrixHIPCOO{Float64, Ti}, csc::SparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    h = x.handle
    z = Ref{Int64}(0)
    esp_check(h, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
    if csc.colptr === x.mirror_colptr && csc.rowval === x.mirror_rowval && z[] == nnz(csc)
        esp_check(h, ccall((:esp_set_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, csc.nzval))
    else
        esp_check(h, set_csc_call(h, csc))
        x.mirror_colptr, x.mirror_rowval = csc.colptr, csc.rowval
    end
    x
end
# ... and back: a flush that added no position returns csc's own pattern vectors with fresh values (esp_get_nzval)
function download_csc!(x::SparseMatrixHIPCOO{Float64, Ti}, csc::SparseMatrixCSC{Float64, Ti}, z::Int64, changed::Bool) where {Ti <: HIPIndex}
    h = x.handle
    if !changed && z == nnz(csc)
        nzval = Vector{Float64}(undef, z)
        esp_check(h, ccall((:esp_get_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, nzval))
        out = SparseMatrixCSC{Float64, Ti}(x.m, x.n, csc.colptr, csc.rowval, nzval)
    else
        colptr, rowval, nzval = Vector{Ti}(undef, x.n + 1), Vector{Ti}(undef, z), Vector{Float64}(undef, z)
        esp_check(h, get_csc_call(h, colptr, rowval, nzval))
        out = SparseMatrixCSC{Float64, Ti}(x.m, x.n, colptr, rowval, nzval)
    end
    x.mirror_colptr, x.mirror_rowval = out.colptr, out.rowval
    out
end

# csc + buffer where the buffer is consumed: its pending entries are folded into the result.  What flush! of the Generic
# wrappers needs: they drop the buffer right after `+` (genericextendablesparsematrixcsc.jl:31-37), see the flush! methods
# below.  keep = true: the handle (device CSC + scratch) lives on in the wrapper's NEXT buffer (adopt!); else its device
# memory is released at once (not at some later GC, which does not see device memory).
function plus_consume!(x::SparseMatrixHIPCOO{Float64, Ti}, csc::SparseMatrixCSC{Float64, Ti}; keep = false) where {Ti <: HIPIndex}
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
function adopt!(old::SparseMatrixHIPCOO{Float64, Ti}) where {Ti <: HIPIndex}
    new = SparseMatrixHIPCOO{Float64, Ti}(old.m, old.n, old.handle, old.rows, old.cols, old.vals, old.kinds, 0, false,
                                             old.mirror_colptr, old.mirror_rowval)
    old.handle, old.released = C_NULL, true
    old.rows, old.cols, old.vals, old.kinds = Int64[], Int64[], Float64[], UInt8[]
    finalizer(y -> (y.handle == C_NULL || ccall((:esp_destroy, libesparse), Int32, (Ptr{Cvoid},), y.handle); y.handle = C_NULL), new)
end

# flush! of the Generic wrappers with a HIP buffer: the reference's `ext.cscmatrix = ext.xmatrix + ext.cscmatrix;
# ext.xmatrix = Tm(m, n)` (genericextendablesparsematrixcsc.jl:31-37) with the buffer consumed instead of copied and the
# device CSC kept attached: per flush! the values go up, and the values (no new position) or the matrix come back
function flush!(ext::GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Float64, Ti}, Float64, Ti}) where {Ti <: HIPIndex}
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
function sum_into!(home::SparseMatrixHIPCOO{Float64, Ti}, xs::Vector{SparseMatrixHIPCOO{Float64, Ti}}, csc::SparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    foreach(commit!, xs)
    sum(nnz, xs) == 0 && return csc
    upload_csc!(home, csc)
    handles = Ptr{Cvoid}[x.handle for x in xs]
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(home.handle, ccall((:esp_flush_sum, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32, Ptr{Int64}, Ptr{Int32}),
                                 home.handle, handles, length(xs), z, changed))
    download_csc!(home, csc, z[], changed[] != 0)
end
function Base.sum(xs::Vector{SparseMatrixHIPCOO{Float64, Ti}}, csc::SparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    home = SparseMatrixHIPCOO{Float64, Ti}(size(csc)...)
    out = sum_into!(home, xs, csc)
    release!(home)                   # (a one-off destination: its device memory goes now, not at some later GC)
    home.released = true
    out
end
# flush! of the MT wrapper with HIP buffers (genericmtextendablesparsematrixcsc.jl:45-51): the destination handle lives
# as long as the wrapper does (one per wrapper, found through its buffer vector's first element ... kept in a WeakKeyDict)
const HOME_OF = WeakKeyDict{Any, SparseMatrixHIPCOO}()
function flush!(ext::GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Float64, Ti}, Float64, Ti}) where {Ti <: HIPIndex}
    home = get!(() -> SparseMatrixHIPCOO{Float64, Ti}(size(ext.cscmatrix)...), HOME_OF, ext)::SparseMatrixHIPCOO{Float64, Ti}
    ext.cscmatrix = sum_into!(home, ext.xmatrices, ext.cscmatrix)
    ext                              # (the buffers are empty T_ext(m, n) again: they are kept)
end

# aliases in the style of src/ExtendableSparse.jl:34-39
const HIPExtendableSparseMatrixCSC{Tv, Ti} = GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}
const MTHIPExtendableSparseMatrixCSC{Tv, Ti} = GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO{Tv, Ti}, Tv, Ti}

# ------------------------------------------------------------------------------------------------
# North-star form (INTEGRATION.md): same fields and methods as ExtendableSparseMatrixCSC
# (src/matrix/extendable.jl:10-25,159-272), but buffer AND CSC stay on the GPU between flushes.  Every update goes to
# the device (no host findindex); esp_flush in ROUTED mode applies updates of stored positions in call order
# (extendable.jl:164-166).
#
# THE FIELD CONTRACT.  The reference's consumers read the FIELDS right after flush!: `p.A.cscmatrix` and `p.A.phash`
# (factorizations/ilu0.jl:126-136, umfpack_lu.jl:18-27, jacobi.jl:54-64).  Here `A.cscmatrix` is a PROPERTY
# (Base.getproperty below): it flushes, brings the host copy up to date -- nothing travels when it is current, nzval only
# (esp_get_nzval, INTO the vector handed out before: the reference updates cscmatrix.nzval in place as well) when no position
# was added since the last read, the whole matrix (esp_get_csc) otherwise -- and returns a valid SparseMatrixCSC, never
# `nothing`.  `A.phash` is a plain field, refreshed by flush! exactly when the CSC was rebuilt (extendable.jl:252).
# `A.cscmatrix = B` (setproperty!) attaches B as the stored matrix (esp_set_csc), as reset! of the reference assigns the field.
#
# HOST EDITS.  A caller may edit the values of the matrix it was handed (`nonzeros(A) .= 0`, sprand.jl:82,
# test_parallel.jl:71-92; `A.cscmatrix.nzval[k] = v`).  Once the property has been read the host copy counts as possibly
# edited: its nzval goes back to the device (esp_set_nzval, 8 bytes per entry) in front of the next update, flush! or device
# consumer -- what the Generic plug-in above does on every flush!.  Edits made while device updates are already pending are
# not merged (the reference applies both to one array in program order; here the upload happens in front of the first
# pending update).  A code that never edits the host copy constructs with `host_edits = false` and pays no upload.
# colptr / rowval of the handed-out matrix must not be edited in place (assign a new matrix: `A.cscmatrix = B`).
const HOST_CURRENT, HOST_VALUES_STALE, HOST_STALE = Int8(0), Int8(1), Int8(2)
mutable struct HIPResidentSparseMatrixCSC{Tv, Ti <: Integer} <: AbstractExtendableSparseMatrixCSC{Tv, Ti}
    buf::SparseMatrixHIPCOO{Tv, Ti}                     # in the role of lnkmatrix (handle + staging chunk)
    host::SparseMatrixCSC{Tv, Ti}                       # the host copy behind the `cscmatrix` property
    host_state::Int8                                    # HOST_CURRENT / HOST_VALUES_STALE (same pattern) / HOST_STALE
    handed_out::Bool                                    # read through the property since the last upload: nzval may carry edits
    host_edits::Bool                                    # false: the caller promises never to edit the host copy
    phash::UInt64
    cpu::Union{ExtendableSparseMatrixCSC{Tv, Ti}, Nothing}   # set by to_cpu!: the matrix lives on the package's CPU path from then on
end
HIPResidentSparseMatrixCSC{Float64, Ti}(m, n; host_edits = true, kwargs...) where {Ti <: HIPIndex} =
    HIPResidentSparseMatrixCSC{Float64, Ti}(SparseMatrixHIPCOO{Float64, Ti}(m, n; kwargs...), spzeros(Float64, Ti, m, n),
                                            HOST_CURRENT, false, host_edits, 0, nothing)
Base.size(A::HIPResidentSparseMatrixCSC) = size(getfield(A, :buf))
oncpu(A::HIPResidentSparseMatrixCSC) = getfield(A, :cpu)

# nzval of a handed-out host copy back to the device (the pattern on both sides is the same: host_state == HOST_CURRENT)
function push_edits!(A::HIPResidentSparseMatrixCSC)
    if getfield(A, :handed_out)
        setfield!(A, :handed_out, false)
        if getfield(A, :host_edits) && getfield(A, :host_state) == HOST_CURRENT && nnz(getfield(A, :host)) > 0
            h = getfield(A, :buf).handle
            esp_check(h, ccall((:esp_set_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, getfield(A, :host).nzval))
        end
    end
    A
end
# in front of anything that changes device values (state >= HOST_VALUES_STALE afterwards) or the pattern (HOST_STALE)
function touch!(A::HIPResidentSparseMatrixCSC, state::Int8 = HOST_VALUES_STALE)
    push_edits!(A)
    setfield!(A, :host_state, max(getfield(A, :host_state), state))
    A
end

function host_csc!(A::HIPResidentSparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    cpu = oncpu(A)
    cpu === nothing || return cpu.cscmatrix
    flush!(A)
    state = getfield(A, :host_state)
    if state != HOST_CURRENT
        h, (m, n) = getfield(A, :buf).handle, size(A)
        z = Ref{Int64}(0)
        esp_check(h, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
        host = getfield(A, :host)
        if state == HOST_VALUES_STALE && z[] == nnz(host)            # values only, into the vector handed out before
            esp_check(h, ccall((:esp_get_nzval, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}), h, host.nzval))
        else
            colptr, rowval, nzval = Vector{Ti}(undef, n + 1), Vector{Ti}(undef, z[]), Vector{Float64}(undef, z[])
            esp_check(h, get_csc_call(h, colptr, rowval, nzval))
            setfield!(A, :host, SparseMatrixCSC{Float64, Ti}(m, n, colptr, rowval, nzval))
        end
        setfield!(A, :host_state, HOST_CURRENT)
    end
    setfield!(A, :handed_out, true)
    getfield(A, :host)
end
function Base.getproperty(A::HIPResidentSparseMatrixCSC, s::Symbol)
    s === :cscmatrix && return host_csc!(A)
    s === :lnkmatrix && return (oncpu(A) === nothing ? nothing : oncpu(A).lnkmatrix)   # (pending entries are on the device)
    s === :phash && oncpu(A) !== nothing && return oncpu(A).phash
    getfield(A, s)
end
Base.propertynames(::HIPResidentSparseMatrixCSC) = (:cscmatrix, :lnkmatrix, :phash)
function Base.setproperty!(A::HIPResidentSparseMatrixCSC{Float64, Ti}, s::Symbol, v) where {Ti <: HIPIndex}
    s === :cscmatrix || return setfield!(A, s, convert(fieldtype(typeof(A), s), v))
    cpu = oncpu(A)
    cpu === nothing || return (cpu.cscmatrix = v)
    csc = convert(SparseMatrixCSC{Float64, Ti}, v)
    @assert size(csc) == size(A)
    buf = getfield(A, :buf)
    commit!(buf)
    esp_check(buf.handle, set_csc_call(buf.handle, csc))
    setfield!(A, :host, csc); setfield!(A, :host_state, HOST_CURRENT); setfield!(A, :handed_out, true)
    csc
end

# An `op` the device cannot fold (extendable.jl:159-197 takes any function): the WHOLE matrix moves to the package's own
# CPU path -- flush, download, release the device memory -- and every later call is forwarded to that
# ExtendableSparseMatrixCSC (with its SparseMatrixLNK).  Results are the reference's by construction.
function to_cpu!(A::HIPResidentSparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    cpu = oncpu(A)
    cpu === nothing || return cpu
    csc = host_csc!(A)
    cpu = ExtendableSparseMatrixCSC{Float64, Ti}(csc, nothing, getfield(A, :phash))
    setfield!(A, :cpu, cpu)
    release!(getfield(A, :buf))
    cpu
end

function Base.setindex!(A::HIPResidentSparseMatrixCSC, v::Number, i::Integer, j::Integer)
    cpu = oncpu(A); cpu === nothing || return setindex!(cpu, v, i, j)
    touch!(A); setindex!(getfield(A, :buf), v, i, j); v
end
function updateindex!(A::HIPResidentSparseMatrixCSC, op::DeviceOp, v, i, j)
    cpu = oncpu(A); cpu === nothing || (updateindex!(cpu, op, v, i, j); return A)
    touch!(A); updateindex!(getfield(A, :buf), op, v, i, j); A
end
function rawupdateindex!(A::HIPResidentSparseMatrixCSC, op::DeviceOp, v, i, j, part = 1)
    cpu = oncpu(A); cpu === nothing || (rawupdateindex!(cpu, op, v, i, j); return A)
    touch!(A); rawupdateindex!(getfield(A, :buf), op, v, i, j); A
end
updateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j) = (updateindex!(to_cpu!(A), op, v, i, j); A)
rawupdateindex!(A::HIPResidentSparseMatrixCSC, op, v, i, j, part = 1) = (rawupdateindex!(to_cpu!(A), op, v, i, j); A)

device_only(A::HIPResidentSparseMatrixCSC, what) =
    oncpu(A) === nothing || throw(ArgumentError("$what: this matrix was moved to the CPU path by an op other than + / - (to_cpu!)"))
assemble_elements!(A::HIPResidentSparseMatrixCSC, cellnodes, elmat, diag = nothing; kwargs...) =
    (device_only(A, "assemble_elements!"); touch!(A); assemble_elements!(getfield(A, :buf), cellnodes, elmat, diag; kwargs...); A)
keep_plan!(A::HIPResidentSparseMatrixCSC, on = true) = (device_only(A, "keep_plan!"); keep_plan!(getfield(A, :buf), on); A)
assemble_elements_again!(A::HIPResidentSparseMatrixCSC, elmat, diag = nothing; kwargs...) =
    (device_only(A, "assemble_elements_again!"); touch!(A); assemble_elements_again!(getfield(A, :buf), elmat, diag; kwargs...); A)

# ExtendableSparseMatrix(I, J, V[, m, n]) (extendable.jl:85-104) = sparse(I,J,V,m,n,+): the triplets go through the
# device pipeline as COO entries (first value as it is, duplicates added in input order, numerical zeros kept)
function HIPResidentSparseMatrixCSC(I::Vector{Ti}, J::Vector{Ti}, V::Vector{Float64}, m = maximum(I), n = maximum(J)) where {Ti <: HIPIndex}
    A = HIPResidentSparseMatrixCSC{Float64, Ti}(m, n; capacity_hint = length(I))
    h = getfield(A, :buf).handle
    esp_check(h, append_host_call(h, I, J, V, ESP_COO))
    flush!(touch!(A))
end

function flush!(A::HIPResidentSparseMatrixCSC)                       # extendable.jl:248-255
    cpu = oncpu(A); cpu === nothing || (flush!(cpu); return A)
    push_edits!(A)
    buf = getfield(A, :buf)
    commit!(buf)
    z, changed = Ref{Int64}(0), Ref{Int32}(0)
    esp_check(buf.handle, ccall((:esp_flush, libesparse), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int32}),
                                buf.handle, ESP_FLUSH_ROUTED, z, changed))
    if changed[] != 0                                                # the CSC was rebuilt: new pattern hash (:252)
        hsh = Ref{UInt64}(0)
        esp_check(buf.handle, ccall((:esp_pattern_hash, libesparse), Int32, (Ptr{Cvoid}, Ptr{UInt64}), buf.handle, hsh))
        setfield!(A, :phash, hsh[])
        setfield!(A, :host_state, HOST_STALE)
    end
    A
end

# getindex (extendable.jl:226-238): pending entries live on the device, so the lookup is flush! + findindex on the
# device CSC (`A[i,j] += v` works; assembly loops should call updateindex!, which is bitwise the same update)
function Base.getindex(A::HIPResidentSparseMatrixCSC, i::Integer, j::Integer)
    cpu = oncpu(A); cpu === nothing || return cpu[i, j]
    flush!(A)
    h = getfield(A, :buf).handle
    v, found = Ref{Float64}(0.0), Ref{Int32}(0)
    esp_check(h, ccall((:esp_getindex, libesparse), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Int32}), h, i, j, v, found))
    v[]
end

function SparseArrays.nnz(A::HIPResidentSparseMatrixCSC)                       # abstractextendablesparsematrixcsc.jl:80
    cpu = oncpu(A); cpu === nothing || return nnz(cpu)
    flush!(A)
    h = getfield(A, :buf).handle
    z = Ref{Int64}(0)
    esp_check(h, ccall((:esp_nnz, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
    z[]
end

SparseArrays.sparse(A::HIPResidentSparseMatrixCSC) = host_csc!(A)              # extendable.jl:258-261: flush!, then the field

# dropzeros!(ext) (abstractextendablesparsematrixcsc.jl:282) and fdrand!'s zero!(A) = nonzeros(A) .= 0 (sprand.jl:82)
function SparseArrays.dropzeros!(A::HIPResidentSparseMatrixCSC)
    cpu = oncpu(A); cpu === nothing || (dropzeros!(cpu); return A)
    flush!(A)
    h = getfield(A, :buf).handle
    z = Ref{Int64}(0)
    esp_check(h, ccall((:esp_dropzeros, libesparse), Int32, (Ptr{Cvoid}, Ptr{Int64}), h, z))
    touch!(A, HOST_STALE)
end
function zero!(A::HIPResidentSparseMatrixCSC)
    cpu = oncpu(A); cpu === nothing || (flush!(cpu); nonzeros(cpu.cscmatrix) .= 0; return A)
    flush!(A)
    h = getfield(A, :buf).handle
    esp_check(h, ccall((:esp_zero_values, libesparse), Int32, (Ptr{Cvoid},), h))
    touch!(A)
end

# Base.copy(ext) (extendable.jl:279-285): CSC, pending entries and phash; device-to-device
function Base.copy(A::HIPResidentSparseMatrixCSC{Float64, Ti}) where {Ti <: HIPIndex}
    device_only(A, "copy")
    push_edits!(A)
    buf = getfield(A, :buf)
    commit!(buf)
    h2 = Ref{Ptr{Cvoid}}(C_NULL)
    esp_check(buf.handle, ccall((:esp_clone, libesparse), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), buf.handle, h2))
    HIPResidentSparseMatrixCSC{Float64, Ti}(wrap_handle(Ti, buf.m, buf.n, h2[]), spzeros(Float64, Ti, buf.m, buf.n), HOST_STALE, false,
                                            getfield(A, :host_edits), getfield(A, :phash), nothing)
end

# consumers that never leave the GPU (SURVEY 8f): mul! sums every row in column order, like the column loop
function LinearAlgebra.mul!(r::Vector{Float64}, A::HIPResidentSparseMatrixCSC, x::Vector{Float64})
    cpu = oncpu(A); cpu === nothing || return mul!(r, cpu, x)
    flush!(A)
    h = getfield(A, :buf).handle
    esp_check(h, ccall((:esp_mul, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), h, x, r, 0))
    r
end
function mark_dirichlet(A::HIPResidentSparseMatrixCSC; penalty = 1.0e20)      # sparsematrixcsc.jl:94-108
    cpu = oncpu(A); cpu === nothing || return mark_dirichlet(cpu; penalty)
    flush!(A)
    h = getfield(A, :buf).handle
    marker = zeros(Bool, size(A, 2))
    esp_check(h, ccall((:esp_mark_dirichlet, libesparse), Int32, (Ptr{Cvoid}, Float64, Ptr{Bool}, Int32), h, penalty, marker, 0))
    marker
end
function eliminate_dirichlet!(A::HIPResidentSparseMatrixCSC, marker::Vector{Bool})   # sparsematrixcsc.jl:121-144
    cpu = oncpu(A); cpu === nothing || (eliminate_dirichlet!(cpu, marker); return A)
    flush!(A)
    h = getfield(A, :buf).handle
    esp_check(h, ccall((:esp_eliminate_dirichlet, libesparse), Int32, (Ptr{Cvoid}, Ptr{Bool}, Int32), h, marker, 0))
    touch!(A)
end
# set-up of the point preconditioners on the device CSC (factorizations/jacobi.jl:5-20, ilu0.jl:8-41)
function jacobi_setup(A::HIPResidentSparseMatrixCSC)
    device_only(A, "jacobi_setup")
    flush!(A)
    h = getfield(A, :buf).handle
    invdiag = Vector{Float64}(undef, size(A, 2))
    esp_check(h, ccall((:esp_jacobi_setup, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), h, invdiag, 0))
    invdiag
end
function ilu0_setup(A::HIPResidentSparseMatrixCSC)
    device_only(A, "ilu0_setup")
    flush!(A)
    h = getfield(A, :buf).handle
    n = size(A, 2)
    xdiag, idiag = Vector{Float64}(undef, n), Vector{Int64}(undef, n)
    esp_check(h, ccall((:esp_ilu0_setup, libesparse), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int32), h, xdiag, idiag, 0))
    xdiag, idiag
end
function reset!(A::HIPResidentSparseMatrixCSC)                                 # extendable.jl:269-272 (phash kept)
    cpu = oncpu(A); cpu === nothing || (reset!(cpu); return A)
    buf = getfield(A, :buf)
    buf.nstaged = 0
    esp_check(buf.handle, ccall((:esp_reset, libesparse), Int32, (Ptr{Cvoid},), buf.handle))
    setfield!(A, :handed_out, false)
    setfield!(A, :host_state, HOST_STALE)
    A
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
    A::HIPResidentSparseMatrixCSC{Float64, Int64}       # this rank's shard: append ANY (i,j) to it (host_edits = false)
    group::Ptr{Cvoid}
end
function HIPShardedSparseMatrixCSC(m, n, nranks, rank, id::Vector{UInt8}; device = rank)
    A = HIPResidentSparseMatrixCSC{Float64, Int64}(m, n; device = device, host_edits = false)
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
    touch!(S.A, HOST_STALE)
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
