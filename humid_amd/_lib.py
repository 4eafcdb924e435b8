"""ctypes loader of libhumid_hip.so (include/humid_hip.h).

Fails loudly: there is no CPU fallback.  torch is imported first so that this library binds to
the HIP runtime torch already loaded (same soname libamdhip64.so.7) and device pointers /
streams are shared."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libhumid_hip.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class HumidSummary(C.Structure):
    _fields_ = [("total", C.c_uint64), ("usable", C.c_uint64), ("unique", C.c_uint64),
                ("clusters", C.c_uint64), ("edges", C.c_uint64), ("nonsingle", C.c_uint64),
                ("ms_count", C.c_float), ("ms_neighbours", C.c_float), ("ms_cluster", C.c_float),
                ("ms_map", C.c_float), ("ms_total", C.c_float), ("ms_h2d", C.c_float),
                ("ms_d2h", C.c_float), ("ms_k_insert", C.c_float), ("ms_k_pairs", C.c_float),
                ("ms_k_cluster", C.c_float), ("ms_k_map", C.c_float), ("ms_k_part", C.c_float),
                ("ms_k_unperm", C.c_float), ("count_mode_used", C.c_uint32)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        # bit 8 of count_mode_used: the word-ordered count ran on 8-byte records (kernels_part8.hip.h)
        d["records8"] = bool(d["count_mode_used"] >> 8)
        d["count_mode_used"] &= 0xff
        return d


HOST_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, C.c_int, C.c_void_p)


class HumidComm(C.Structure):
    """humid_comm of include/humid_hip.h: what moves bytes between ranks for humid_dedup_run_exchange"""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("host_all_gather", HOST_ALL_GATHER_FN), ("exchange", EXCHANGE_FN)]


class HumidExchangeInfo(C.Structure):
    _fields_ = [("unique_local", C.c_uint64), ("id_base", C.c_uint64), ("n_nodes", C.c_uint64),
                ("n_pairs", C.c_uint64), ("d_unique_count", C.c_void_p), ("d_unique_degree", C.c_void_p)]


# every symbol include/humid_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "humid_abi_version": (C.c_uint32, []),
    "humid_device_count": (C.c_int, []),
    "humid_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "humid_ctx_destroy": (None, [C.c_void_p]),
    "humid_last_error": (C.c_char_p, [C.c_void_p]),
    "humid_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "humid_ctx_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "humid_host_alloc": (C.c_void_p, [C.c_uint64]),
    "humid_host_free": (None, [C.c_void_p]),
    "humid_dedup_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.POINTER(HumidSummary)]),
    "humid_dedup_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_void_p, C.POINTER(HumidSummary)]),
    "humid_dedup_run_bases": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.POINTER(HumidSummary)]),
    "humid_get_packed_words": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "humid_get_leaves": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6),
    "humid_get_adjacency": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "humid_get_clusters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "humid_get_histogram": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_uint64, u64p]),
    "humid_cluster_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, u32p]),
    "humid_stage_histogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.c_uint32, C.c_void_p]),
    "humid_stage_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                    C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p]),
    "humid_stage_unique": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p)]),
    "humid_stage_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.POINTER(HumidSummary)]),
    "humid_stage_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                  C.c_void_p]),
    "humid_stage_count_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                          C.c_uint64, C.c_uint64, u64p, C.c_uint32, u64p, u64p, u64p]),
    "humid_stage_map_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), u64p]),
    "humid_stage_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), u64p]),
    "humid_stage_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), u32p]),
    "humid_stage_route_words": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "humid_stage_route": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p, C.c_uint32, u64p,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "humid_stage_route_check": (C.c_int, [C.c_void_p]),
    "humid_stage_exchange_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p)]),
    "humid_stage_plan_info": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, u32p, u32p]),
    "humid_stage_combo_route": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                          C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), u64p]),
    "humid_stage_pairs_keyed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64,
                                          C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                          C.POINTER(C.c_void_p), u64p]),
    "humid_stage_compact_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                            C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]),
    "humid_stage_pairs_edit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), u64p]),
    "humid_stage_unique_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                           C.POINTER(C.c_void_p), u64p]),
    "humid_stage_graph_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                          C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(HumidSummary)]),
    "humid_stage_owned_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64p, C.c_uint32,
                                            C.POINTER(C.c_void_p), u64p]),
    "humid_stage_owner_perm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p,
                                         C.c_uint32, C.POINTER(C.c_void_p), u64p]),
    "humid_stage_owner_perm_wide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, u64p, u64p,
                                              C.c_uint32, C.POINTER(C.c_void_p), u64p]),
    "humid_stage_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_void_p]),
    "humid_dedup_run_exchange": (C.c_int, [C.c_void_p, C.POINTER(HumidComm), C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.POINTER(HumidSummary), C.POINTER(HumidExchangeInfo)]),
    "humid_shm_open": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "humid_shm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "humid_shm_abort": (None, [C.c_void_p]),
    "humid_shm_close": (None, [C.c_void_p]),
    "humid_at_least_double": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]),
}

_LIB = None


class HumidLibraryError(RuntimeError):
    pass


def load(import_torch: bool = True):
    """dlopen the HIP library and bind every C-ABI symbol.  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if import_torch:
        try:
            import torch  # noqa: F401  (loads libamdhip64.so.7 first; see module docstring)
        except Exception:  # torch absent: the library still works with the system runtime
            pass
    if not os.path.exists(SO_PATH):
        raise HumidLibraryError(
            "%s not found: build it with `python -m humid_amd.build` (hipcc, gfx950). "
            "humid_amd has no CPU fallback." % SO_PATH)
    try:
        lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise HumidLibraryError("cannot load %s: %s" % (SO_PATH, e)) from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HumidLibraryError("%s does not export %s" % (SO_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.humid_abi_version() != 5:
        raise HumidLibraryError("ABI version mismatch")
    _LIB = lib
    return lib
