"""ctypes loader for libbwt_mi355x.so (the HIP kernels and their C-ABI).

There is no CPU fallback: if the shared library is missing this module raises,
and every compute call fails loudly when no gfx950 device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbwt_mi355x.so")

vp = C.c_void_p
u64 = C.c_uint64
u32 = C.c_uint32
sz = C.c_size_t


class BwtGpuError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/bwt_gpu.h one to one
SIGNATURES = {
    "bwtgpu_init": (C.c_int, [C.c_int]),
    "bwtgpu_last_error": (C.c_char_p, []),
    "bwtgpu_device_count": (C.c_int, []),
    "bwtgpu_timing_enable": (C.c_int, [C.c_int]),
    "bwtgpu_timing_reset": (C.c_int, []),
    "bwtgpu_timing_get": (C.c_int, [C.c_char_p, C.POINTER(u64), C.POINTER(C.c_double)]),
    "bwtgpu_timing_report": (C.c_int, [C.c_char_p, sz]),
    "bwtgpu_sa_workspace_bytes": (sz, [u32]),
    "bwtgpu_sa_build": (C.c_int, [vp, u32, vp, vp, sz, vp]),
    "bwtgpu_sa_last_stats": (C.c_int, [C.POINTER(u64)]),
    "bwtgpu_bwt_from_sa": (C.c_int, [vp, vp, u32, vp, vp, vp]),
    "bwtgpu_index_build": (vp, [vp, u32, vp]),
    "bwtgpu_index_build_dev": (vp, [vp, u32, vp]),
    "bwtgpu_release_cache": (C.c_int, []),
    "bwtgpu_index_from_parts": (vp, [vp, u32, u32, vp, vp]),
    "bwtgpu_index_free": (None, [vp]),
    "bwtgpu_index_info": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
    "bwtgpu_index_export": (C.c_int, [vp, vp, vp]),
    "bwtgpu_index_rank_rows16": (C.c_int, [vp, vp]),
    "bwtgpu_index_rank_rows": (C.c_int, [vp, u32, vp]),
    "bwtgpu_index_device_bytes": (sz, [vp]),
    "bwtgpu_index_kmer_table": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32)]),
    "bwtgpu_mms": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, vp]),
    "bwtgpu_search_reads": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, vp, vp]),
    "bwtgpu_align_reads": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, u32, vp, vp, vp, vp]),
    "bwtgpu_align_reads_host": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, vp, vp, vp]),
    "bwtgpu_dp_align_batch": (C.c_int, [vp, vp, vp, vp, u64, C.c_int, u32, u32, vp, vp, vp, vp]),
    "bwtgpu_mms_host": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp]),
    "bwtgpu_search_reads_host": (C.c_int, [vp, vp, vp, u64, C.c_int, vp]),
    "bwtgpu_occ_batch": (C.c_int, [vp, vp, vp, u64, C.c_int, vp, vp]),
    "bwtgpu_lf_batch": (C.c_int, [vp, vp, u64, vp, vp]),
    "bwtgpu_sort_suffixes_host": (C.c_int, [vp, u32, vp, u64]),
    "bwtgpu_sa_build_host": (C.c_int, [vp, u32, vp]),
    "bwtgpu_bwt_from_sa_host": (C.c_int, [vp, vp, u32, vp, vp]),
    "bwtgpu_reverse_search_host": (C.c_int, [vp, vp, u32, u64, u64, vp]),
    "bwtgpu_search_host": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, vp]),
    "bwtgpu_locate_rows_host": (C.c_int, [vp, vp, u64, vp]),
    "bwtgpu_occ_host": (C.c_int, [vp, vp, vp, u64, C.c_int, vp]),
    "bwtgpu_lf_host": (C.c_int, [vp, vp, u64, vp]),
    "bwtgpu_search_workspace_bytes": (sz, [u64, u32]),
    "bwtgpu_search": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, vp, sz, vp]),
    "bwtgpu_search_packed": (C.c_int, [vp, vp, u32, u64, vp, vp, vp, vp]),
    "bwtgpu_search_locate_workspace_bytes": (sz, [u64, u32]),
    "bwtgpu_search_locate": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, vp, sz, vp]),
    "bwtgpu_search_locate_packed_workspace_bytes": (sz, [u64]),
    "bwtgpu_search_locate_packed": (C.c_int, [vp, vp, u32, u64, vp, vp, vp, vp, sz, vp]),
    "bwtgpu_pack_reads_host": (C.c_int, [vp, u32, u64, u64, vp]),
    "bwtgpu_search_packed_host": (C.c_int, [vp, vp, u32, u64, vp, vp, vp, vp]),
    "bwtgpu_search_static": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, vp]),
    "bwtgpu_sa_bucket_counts": (C.c_int, [vp, u32, vp, vp]),
    "bwtgpu_sa_build_range": (C.c_int, [vp, u32, u32, u32, vp, vp, u64, vp, vp, vp, vp, sz, vp]),
    "bwtgpu_pack_bw2": (C.c_int, [vp, u64, vp, vp]),
    "bwtgpu_index_build_from_shards": (vp, [vp, u32, u32, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64),
                                            vp, vp, vp]),
    "bwtgpu_index_attach_sa": (C.c_int, [vp, vp, vp, vp]),
    "bwtgpu_index_keep_lean": (C.c_int, [vp]),
    "bwtgpu_index_build_from_sa": (vp, [vp, u32, vp, vp, u32, vp]),
    "bwtgpu_search_profile": (C.c_int, [vp, vp, u32, u64, u64, vp, vp, vp, sz, vp, vp]),
    "bwtgpu_index_compact": (C.c_int, [vp]),
    "bwtgpu_sa_certify_workspace_bytes": (sz, [u32]),
    "bwtgpu_sa_certify": (C.c_int, [vp, u32, vp, vp, sz, C.POINTER(u64), vp]),
    "bwtgpu_index_certify": (C.c_int, [vp, vp, vp, sz, C.POINTER(u64), vp]),
    "bwtgpu_index_prepare_search": (C.c_int, [vp, vp]),
    "bwtgpu_index_replicate": (vp, [vp, C.c_int]),
    "bwtgpu_index_attach_text": (C.c_int, [vp, vp, vp]),
    "bwtgpu_index_attach_text_dev": (C.c_int, [vp, vp, vp]),
    "bwtgpu_locate_workspace_bytes": (sz, []),
    "bwtgpu_locate_rows": (C.c_int, [vp, vp, u64, vp, vp, vp, sz, vp]),
    "bwtgpu_gen_genome": (C.c_int, [vp, u64, u64, vp]),
    "bwtgpu_gen_reads": (C.c_int, [vp, u64, vp, u64, u32, u64, vp, vp]),
    "bwtgpu_gen_reads_slice": (C.c_int, [vp, u64, vp, u64, u64, u32, u64, vp, vp]),
    "bwtgpu_diag_gather_rate": (C.c_int, [vp, u64, u64, C.POINTER(C.c_double), vp]),
    "bwtgpu_diag_scan": (C.c_int, [C.c_int, vp, vp, u64, vp, vp]),
    "bwtgpu_diag_radix_sort": (C.c_int, [vp, vp, u64, C.c_int, vp]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BwtGpuError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950); "
                "there is no CPU fallback"
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        err = BwtGpuError(f"libbwt_mi355x error {rc}: {load().bwtgpu_last_error().decode()}")
        err.code = rc
        raise err


def last_error():
    return load().bwtgpu_last_error().decode()
