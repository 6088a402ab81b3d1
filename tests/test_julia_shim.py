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
            "esp_flush_sum", "esp_set_nzval", "esp_get_nzval", "esp_append_elements_host"}
    assert need <= called, sorted(need - called)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = open(HDR).read()
    for name in sorted(set(re.findall(r"`(esp_\w+)", doc))):
        if name.endswith("_t") or name.endswith("_"):       # a type of the header / a family of calls (`esp_shard_*`)
            assert name in hdr, name
            continue
        assert name in header_prototypes(), "INTEGRATION.md mentions %s, which the header does not declare" % name
