"""ctypes binding of libesparse_hip.so -- the C ABI declared in include/esparse_hip.h.

The product has no CPU path: if the shared library is missing this module raises at
import of the symbols, and every call that needs a GPU returns ESP_ERR_NODEVICE
(surfaced as NoDeviceError) on a machine without one.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libesparse_hip.so")

ESP_OK = 0
ESP_ERR_INVALID, ESP_ERR_BOUNDS, ESP_ERR_HIP, ESP_ERR_NOMEM = -1, -2, -3, -4
ESP_ERR_UNSUPPORTED, ESP_ERR_STATE, ESP_ERR_NODEVICE = -5, -6, -7
ESP_SET, ESP_UPDATE, ESP_RAWUPDATE, ESP_COO = 0, 1, 2, 3
ESP_OP_ADD, ESP_OP_SUB = 0, 1
ESP_FLUSH_ROUTED, ESP_FLUSH_PLUS = 0, 1
STAGES = ("append", "hist", "scan", "scatter", "local", "fold", "colptr", "merge", "copy")
ESP_ST_COUNT = len(STAGES)


class EspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("esparse_hip error %d: %s" % (code, msg))
        self.code = code


class NoDeviceError(EspError):
    pass


class BoundsError(IndexError):
    pass


class esp_timing_t(C.Structure):
    _fields_ = [("ms", C.c_double * ESP_ST_COUNT), ("launches", C.c_int64 * ESP_ST_COUNT),
                ("flush_ms", C.c_double), ("flushes", C.c_int64)]


i32, i64, u64, f64 = C.c_int32, C.c_int64, C.c_uint64, C.c_double
vp = C.c_void_p
P = C.POINTER

# name -> (restype, argtypes); mirrors include/esparse_hip.h one to one
SIGNATURES = {
    "esp_create": (i32, [i64, i64, i32, i64, P(vp)]),
    "esp_destroy": (i32, [vp]),
    "esp_clone": (i32, [vp, P(vp)]),
    "esp_last_error": (C.c_char_p, [vp]),
    "esp_version": (C.c_char_p, []),
    "esp_set_stream": (i32, [vp, vp]),
    "esp_synchronize": (i32, [vp]),
    "esp_size": (i32, [vp, P(i64), P(i64)]),
    "esp_key_layout": (i32, [vp, P(i32), P(i32)]),
    "esp_stage_begin": (i32, [vp, i64, P(vp), P(vp), P(vp), P(vp), P(i64)]),
    "esp_commit": (i32, [vp, i64, i32, i32]),
    "esp_append_host": (i32, [vp, vp, vp, vp, vp, i32, i32, i64]),
    "esp_append_host_i32": (i32, [vp, vp, vp, vp, vp, i32, i32, i64]),
    "esp_append_device": (i32, [vp, vp, vp, vp, vp, i32, i32, i64]),
    "esp_append_packed": (i32, [vp, vp, vp, i64]),
    "esp_generate_fdrand": (i32, [vp, i64, i64, i64, u64, i32, i32]),
    "esp_generate_fem": (i32, [vp, i32, i64, u64, i32]),
    "esp_append_elements": (i32, [vp, i32, i64, vp, vp, vp, i32, i32]),
    "esp_append_elements_host": (i32, [vp, i32, i64, vp, vp, vp, i32, i32]),
    "esp_elements_keep_plan": (i32, [vp, i32]),
    "esp_append_elements_again": (i32, [vp, vp, vp, i32, i32]),
    "esp_append_elements_again_host": (i32, [vp, vp, vp, i32, i32]),
    "esp_generate_fem_mesh": (i32, [vp, i32, i64, u64, i32, i32, u64, i64, i64, vp, vp, vp]),
    "esp_generate_fdrand_range": (i32, [vp, i64, i64, i64, u64, i32, i32, i64, i64]),
    "esp_set_column_window": (i32, [vp, i64, i64]),
    "esp_pending": (i32, [vp, P(i64)]),
    "esp_set_csc": (i32, [vp, vp, vp, vp, i64]),
    "esp_flush": (i32, [vp, i32, P(i64), P(i32)]),
    "esp_nnz": (i32, [vp, P(i64)]),
    "esp_get_csc": (i32, [vp, vp, vp, vp]),
    "esp_get_nzval": (i32, [vp, vp]),
    "esp_set_csc_i32": (i32, [vp, vp, vp, vp, i64]),
    "esp_get_csc_i32": (i32, [vp, vp, vp, vp]),
    "esp_set_nzval": (i32, [vp, vp]),
    "esp_flush_sum": (i32, [vp, vp, i32, P(i64), P(i32)]),
    "esp_csc_device": (i32, [vp, P(vp), P(vp), P(vp)]),
    "esp_reset": (i32, [vp]),
    "esp_clear_pending": (i32, [vp]),
    "esp_zero_values": (i32, [vp]),
    "esp_dropzeros": (i32, [vp, P(i64)]),
    "esp_getindex": (i32, [vp, i64, i64, P(f64), P(i32)]),
    "esp_pattern_hash": (i32, [vp, P(u64)]),
    "esp_pending_getindex": (i32, [vp, i64, i64, P(f64), P(i32)]),
    "esp_release_buffers": (i32, [vp]),
    "esp_mul": (i32, [vp, vp, vp, i32]),
    "esp_mark_dirichlet": (i32, [vp, f64, vp, i32]),
    "esp_eliminate_dirichlet": (i32, [vp, vp, i32]),
    "esp_jacobi_setup": (i32, [vp, vp, i32]),
    "esp_ilu0_setup": (i32, [vp, vp, vp, i32]),
    "esp_shard_counts": (i32, [vp, i32, vp]),
    "esp_shard_export": (i32, [vp, i32, vp, vp, vp]),
    "esp_shard_exchange_begin": (i32, [vp, i32, i32, i64, i64, P(vp), P(vp), vp]),
    "esp_shard_exchange_place": (i32, [vp, i64, vp, vp, i64]),
    "esp_shard_partition": (i32, [vp, i32, i32, i64, P(i32), P(vp), P(vp), P(vp), vp, P(i64)]),
    "esp_shard_assemble": (i32, [vp, vp, vp, vp, vp, P(i32)]),
    "esp_shard_plan": (i32, [vp, i32, i32, i64]),
    "esp_debug_last_shard_source": (i32, [vp, P(i32)]),
    "esp_debug_last_local_small": (i32, [vp, P(i32)]),
    "esp_debug_last_lazy_items": (i32, [vp, P(i32)]),
    "esp_debug_last_sum_join": (i32, [vp, P(i32)]),
    "esp_debug_last_sum_ms": (i32, [vp, P(C.c_double), P(C.c_double)]),
    "esp_debug_last_sum_batched": (i32, [vp, P(C.c_int32)]),
    "esp_debug_last_sum_plan_bits": (i32, [vp, P(i32), P(i32)]),
    "esp_debug_last_rebuild": (i32, [vp, P(i32)]),
    "esp_group_unique_id": (i32, [vp]),
    "esp_group_create": (i32, [vp, i32, i32, vp, P(vp)]),
    "esp_group_create_comm": (i32, [vp, i32, i32, vp, P(vp)]),
    "esp_group_destroy": (i32, [vp]),
    "esp_group_handle": (i32, [vp, P(vp)]),
    "esp_group_last_error": (C.c_char_p, [vp]),
    "esp_group_column_range": (i32, [vp, P(i64), P(i64)]),
    "esp_group_flush": (i32, [vp, i32, P(i64), P(i32)]),
    "esp_group_nnz": (i32, [vp, P(i64), P(i64)]),
    "esp_group_get_csc": (i32, [vp, vp, vp, vp]),
    "esp_group_last_exchange": (i32, [vp, P(i32), P(i64)]),
    "esp_debug_group_loopback": (i32, [vp, i32, P(i64)]),
    "esp_timing_enable": (i32, [vp, i32]),
    "esp_timing": (i32, [vp, P(esp_timing_t), i32]),
    "esp_debug_plan_cap": (i32, [vp, C.c_double]),
    "esp_debug_fail_next_bucket_stage": (i32, [vp]),
    "esp_debug_force_path": (i32, [vp, i32]),
    "esp_debug_last_run_order": (i32, [vp, P(i32)]),
    "esp_debug_last_colptr_direct": (i32, [vp, P(i32)]),
    "esp_debug_last_key_bytes": (i32, [vp, P(i32)]),
    "esp_debug_last_fold_update": (i32, [vp, P(i32)]),
    "esp_debug_last_path": (i32, [vp, P(i32)]),
    "esp_debug_last_partition": (i32, [vp, P(i32)]),
    "esp_debug_last_plan_reused": (i32, [vp, P(i32)]),
}

# esp_comm_t: the host-supplied transport of esp_group_create_comm
ALLGATHER_FN = C.CFUNCTYPE(i32, vp, P(i64), i32, P(i64))
ALLTOALLV_FN = C.CFUNCTYPE(i32, vp, P(vp), P(i64), P(vp), P(i64), vp)


class esp_comm_t(C.Structure):
    _fields_ = [("ctx", vp), ("allgather_i64", ALLGATHER_FN), ("alltoallv_dev", ALLTOALLV_FN)]


_lib = None


def load():
    """dlopen the in-tree shared library and bind every declared symbol (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError("libesparse_hip.so is not built (%s); run __graft_entry__.build() -- "
                          "there is no CPU fallback" % SO)
    L = C.CDLL(SO)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError if the ABI and the header drift apart
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(h, rc):
    if rc == ESP_OK:
        return
    msg = load().esp_last_error(h)
    msg = msg.decode() if msg else ""
    if rc == ESP_ERR_BOUNDS:
        raise BoundsError(msg)
    if rc == ESP_ERR_NODEVICE:
        raise NoDeviceError(rc, msg)
    raise EspError(rc, msg)
