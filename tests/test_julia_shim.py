"""Static check of the reference-side binding (extendablesparse.jl_amd/julia/ESparseHIP.jl): there is no Julia in the
build container, so every `ccall((:esp_xxx, libesparse), Ret, (ArgTypes...), args...)` of the shim is parsed and held
against the prototypes of include/esparse_hip.h: the symbol exists, the arity matches, every argument is a pointer
where C has a pointer and an integer / float of the right width where C has a scalar, and the return type fits.
Also: every C entry point INTEGRATION.md lists for the shim is really called by it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "extendablesparse.jl_amd", "julia", "ESparseHIP.jl")
HDR = os.path.join(ROOT, "include", "esparse_hip.h")

C_SCALAR = {"int64_t": "Int64", "int32_t": "Int32", "double": "Float64", "uint64_t": "UInt64", "uint8_t": "UInt8"}


def header_prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int32_t|const char \*)\s*(esp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "(*" in a:
                    kinds.append("ptr")
                elif "*" in a:
                    kinds.append("ptr")
                else:
                    t = a.replace("const ", "").split()[0]
                    kinds.append(C_SCALAR[t])
        protos[name] = (("Cstring" if "char" in ret else "Int32"), kinds)
    return protos


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def shim_ccalls():
    src = open(JL).read()
    src = re.sub(r"#.*", "", src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(esp_\w+),\s*libesparse\),\s*(\w+),\s*\(", src):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        argt = split_top(src[m.end():i - 1])
        # the call's actual arguments: up to the ccall's closing parenthesis
        j, depth = i, 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        actual = split_top(src[i:j - 1].lstrip(", \n"))
        calls.append((name, ret, [a for a in argt if a], actual))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    calls = shim_ccalls()
    assert len(calls) >= 30
    for name, ret, argt, actual in calls:
        assert name in protos, "%s is not declared in include/esparse_hip.h" % name
        cret, kinds = protos[name]
        assert ret == cret, (name, ret, cret)
        assert len(argt) == len(kinds), (name, argt, kinds)
        assert len(actual) == len(kinds), (name, actual, kinds)
        for t, k in zip(argt, kinds):
            if k == "ptr":
                assert t.startswith("Ptr{") or t.startswith("Ref{"), (name, t, k)
            else:
                assert t == k or (k == "UInt8" and t == "Bool"), (name, t, k)


def test_shim_binds_what_integration_md_lists():
    called = {c[0] for c in shim_ccalls()}
    need = {"esp_create", "esp_destroy", "esp_stage_begin", "esp_commit", "esp_pending", "esp_set_csc", "esp_flush",
            "esp_get_csc", "esp_nnz", "esp_getindex", "esp_pending_getindex", "esp_dropzeros", "esp_zero_values",
            "esp_clone", "esp_append_host", "esp_pattern_hash", "esp_mul", "esp_mark_dirichlet", "esp_eliminate_dirichlet",
            "esp_jacobi_setup", "esp_ilu0_setup", "esp_reset", "esp_release_buffers", "esp_last_error",
            "esp_group_unique_id", "esp_group_create", "esp_group_destroy", "esp_group_flush", "esp_group_nnz",
            "esp_group_column_range", "esp_group_get_csc",
            # round 4: one-call Base.sum, values-only transfers of the plug-in that keeps its CSC attached, element-level assembly
            "esp_flush_sum", "esp_set_nzval", "esp_get_nzval", "esp_append_elements_host",
            # round 5: Ti = Int32 dispatches to the _i32 transfers
            "esp_append_host_i32", "esp_set_csc_i32", "esp_get_csc_i32"}
    assert need <= called, sorted(need - called)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = open(HDR).read()
    for name in sorted(set(re.findall(r"`(esp_\w+)", doc))):
        if name.endswith("_t") or name.endswith("_"):       # a type of the header / a family of calls (`esp_shard_*`)
            assert name in hdr, name
            continue
        assert name in header_prototypes(), "INTEGRATION.md mentions %s, which the header does not declare" % name


def _shim_source():
    return re.sub(r"#.*", "", open(JL).read())


def _function_body(src, header_regex):
    """Text of the Julia function whose first line matches header_regex, up to the `end` in column 0 (one-liners: the line)."""
    m = re.search(header_regex, src, flags=re.M)
    assert m, header_regex
    rest = src[m.start():]
    if not rest.lstrip().startswith("function"):
        return rest.split("\n", 1)[0]
    e = re.search(r"^end\b", rest, flags=re.M)
    assert e, header_regex
    return rest[:e.end()]


def test_cscmatrix_is_a_property_that_downloads():
    """VERDICT r4, missing #2: the reference's consumers read the FIELD A.cscmatrix right after flush!
    (factorizations/ilu0.jl:126-136, umfpack_lu.jl:18-27, jacobi.jl:54-64).  The sibling struct answers that read with a
    valid host CSC: a getproperty method for :cscmatrix that flushes and downloads (values only when no position was added),
    never a field that may hold `nothing`."""
    src = _shim_source()
    struct = _function_body(src.replace("mutable struct", "function"), r"^function HIPResidentSparseMatrixCSC\{Tv, Ti")
    assert "cscmatrix" not in struct, "cscmatrix must not be a plain field any more"
    assert "Nothing}" not in struct.split("cpu::")[0], "no field of the host copy may be Nothing"
    getp = _function_body(src, r"^function Base\.getproperty\(A::HIPResidentSparseMatrixCSC")
    assert re.search(r"s === :cscmatrix\s*&&\s*return host_csc!\(A\)", getp)
    host = _function_body(src, r"^function host_csc!\(A::HIPResidentSparseMatrixCSC")
    assert "flush!(A)" in host
    assert ":esp_get_nzval" in host and "get_csc_call" in host and "HOST_VALUES_STALE" in host
    assert host.index("flush!(A)") < host.index(":esp_get_nzval")
    # sparse(A) is that same function; A.cscmatrix = B attaches B
    assert re.search(r"^SparseArrays\.sparse\(A::HIPResidentSparseMatrixCSC\) = host_csc!\(A\)", src, flags=re.M)
    setp = _function_body(src, r"^function Base\.setproperty!\(A::HIPResidentSparseMatrixCSC")
    assert "set_csc_call" in setp
    # host edits of a handed-out copy go back (esp_set_nzval) in front of every update / flush! / device consumer
    push = _function_body(src, r"^function push_edits!\(A::HIPResidentSparseMatrixCSC")
    assert ":esp_set_nzval" in push and "handed_out" in push
    flush = _function_body(src, r"^function flush!\(A::HIPResidentSparseMatrixCSC\)")
    assert "push_edits!(A)" in flush and flush.index("push_edits!(A)") < flush.index(":esp_flush")
    assert "HOST_STALE" in flush
    touch = _function_body(src, r"^function touch!\(A::HIPResidentSparseMatrixCSC")
    assert "push_edits!(A)" in touch
    for f in ("updateindex!", "rawupdateindex!"):
        body = _function_body(src, r"^function %s\(A::HIPResidentSparseMatrixCSC, op::DeviceOp" % f)
        assert body.index("touch!(A)") < body.index("getfield(A, :buf)")


def test_other_ops_fall_back_to_the_cpu_path():
    """VERDICT r4, missing #3: extendable.jl:159-197 takes any `op`.  + and - run on the device; any other function moves
    that matrix to the package's own ExtendableSparseMatrixCSC + SparseMatrixLNK (to_cpu!), and the bare plug-in buffer
    throws an ArgumentError -- never a MethodError."""
    src = _shim_source()
    assert re.search(r"^const DeviceOp = Union\{typeof\(\+\), typeof\(-\)\}", src, flags=re.M)
    assert re.search(r"^updateindex!\(A::HIPResidentSparseMatrixCSC, op, v, i, j\) = \(updateindex!\(to_cpu!\(A\), op, v, i, j\); A\)",
                     src, flags=re.M)
    assert re.search(r"^rawupdateindex!\(A::HIPResidentSparseMatrixCSC, op, v, i, j, part = 1\) = "
                     r"\(rawupdateindex!\(to_cpu!\(A\), op, v, i, j\); A\)", src, flags=re.M)
    tocpu = _function_body(src, r"^function to_cpu!\(A::HIPResidentSparseMatrixCSC")
    assert "ExtendableSparseMatrixCSC{Float64, Ti}(csc, nothing" in tocpu and "release!" in tocpu and "host_csc!(A)" in tocpu
    # every method of the sibling struct forwards once the matrix lives on the CPU
    for name in ("flush!", "Base.getindex", "SparseArrays.nnz", "SparseArrays.dropzeros!", "reset!", "LinearAlgebra.mul!",
                 "Base.setindex!", "eliminate_dirichlet!"):
        body = _function_body(src, r"^function %s\((r::Vector\{Float64\}, )?A::HIPResidentSparseMatrixCSC" % re.escape(name))
        assert "oncpu(A)" in body, name
    assert re.search(r"^updateindex!\(x::SparseMatrixHIPCOO, op, v, i, j\) = unsupported_op\(op\)", src, flags=re.M)
    assert re.search(r"^rawupdateindex!\(x::SparseMatrixHIPCOO, op, v, i, j, tid = 1\) = unsupported_op\(op\)", src, flags=re.M)
    assert "throw(ArgumentError(" in _function_body(src, r"^unsupported_op\(op\)")


def test_int32_indices_dispatch_to_the_i32_entry_points():
    """VERDICT r4, missing #3 (second half): Ti = Int32 binds esp_append_host_i32 / esp_set_csc_i32 / esp_get_csc_i32, and no
    method of the shim is pinned to {Float64, Int64} any more (the sharded struct aside: its group calls are Int64)."""
    src = _shim_source()
    assert re.search(r"^const HIPIndex = Union\{Int32, Int64\}", src, flags=re.M)
    calls = {(c[0], tuple(c[2])) for c in shim_ccalls()}
    names = {c[0]: c[1] for c in calls}
    for name in ("esp_append_host_i32", "esp_set_csc_i32", "esp_get_csc_i32"):
        assert name in names
        assert "Ptr{Int32}" in names[name] and "Ptr{Int64}" not in names[name], (name, names[name])
    for helper, sym in (("set_csc_call", "esp_set_csc"), ("get_csc_call", "esp_get_csc"), ("append_host_call", "esp_append_host")):
        assert re.search(r"^%s\(h, \w+::(SparseMatrixCSC\{Float64, |Vector\{)Int32\}.*=\s*\n\s*ccall\(\(:%s_i32," % (helper, sym),
                         src, flags=re.M), helper
        assert re.search(r"^%s\(h, \w+::(SparseMatrixCSC\{Float64, |Vector\{)Int64\}.*=\s*\n\s*ccall\(\(:%s," % (helper, sym),
                         src, flags=re.M), helper
    pinned = [l for l in src.splitlines() if "{Float64, Int64}" in l and "HIPSharded" not in l and "set_csc_call" not in l
              and "A::HIPResidentSparseMatrixCSC{Float64, Int64}" not in l and "A = HIPResidentSparseMatrixCSC{Float64, Int64}" not in l]
    assert not pinned, pinned
