"""ctypes binding of libmi355_lzss.so (the C ABI declared in include/mi355_lzss.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_lzss.so")

OK, E_ARG, E_DICT_SIZE, E_INPUT_SIZE, E_DEVICE, E_CAPACITY, E_FORMAT = 0, -1, -2, -3, -4, -5, -6

_vp, _sz, _u32, _u64, _i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int

# name -> (restype, argtypes); every symbol include/mi355_lzss.h declares
SIGNATURES = {
    "mi355_lzss_ctx_create": (_i32, [_vp, _sz, _i32, C.POINTER(_vp)]),
    "mi355_lzss_ctx_destroy": (None, [_vp]),
    "mi355_lzss_ctx_create_decoder": (_i32, [_vp, _sz, _i32, C.POINTER(_vp)]),
    "mi355_lzss_ctx_set_chunk_bytes": (_i32, [_vp, _u64]),
    "mi355_lzss_host_alloc": (_vp, [_sz]),
    "mi355_lzss_host_free": (None, [_vp]),
    "mi355_lzss_compressed_size_batch": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "mi355_lzss_compress_batch_multi": (_i32, [_vp, _i32, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_lzss_decompress_batch_multi": (_i32, [_vp, _i32, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_lzss_partition": (_i32, [_vp, _u32, _i32, _vp]),
    "mi355_lzss_ctx_set_overlap": (_i32, [_vp, _i32]),
    "mi355_lzss_last_error": (C.c_char_p, []),
    "mi355_lzss_dict_info": (_i32, [_vp, C.POINTER(_u64), C.POINTER(C.c_int32 * 2)]),
    "mi355_lzss_best_backrefs": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "mi355_lzss_write": (_i32, [_vp, _vp, _sz, _sz, C.c_uint, _vp, _sz, C.POINTER(_u64)]),
    "mi355_lzss_write_noalloc": (_i32, [_vp, _vp, _sz, _sz, C.c_uint, _vp, _sz, _vp, _vp, _vp]),
    "mi355_lzss_compressed_size": (_i32, [_vp, _vp, _sz, C.POINTER(_u64)]),
    "mi355_lzss_pack_blob": (_i32, [_vp, _vp, _u64, _u32, _u64, _u32, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64),
                                    C.POINTER(_i32)]),
    "mi355_lzss_compress_batch": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_lzss_decompress_batch": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "mi355_lzss_batch_create": (_i32, [_vp, _u32, _u64, C.POINTER(_vp)]),
    "mi355_lzss_batch_destroy": (None, [_vp]),
    "mi355_lzss_batch_upload": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "mi355_lzss_batch_compress": (_i32, [_vp, _i32]),
    "mi355_lzss_batch_sizes": (_i32, [_vp, _vp, _vp, C.POINTER(_u64)]),
    "mi355_lzss_batch_download": (_i32, [_vp, _vp, _vp, _vp]),
    "mi355_lzss_batch_results": (_i32, [_vp, _u32, _vp, _sz]),
    "mi355_lzss_batch_index": (_i32, [_vp, _u32, _vp, _vp, _vp, _sz]),
    "mi355_lzss_batch_upload_compressed": (_i32, [_vp, _vp, _vp, _u32, _vp]),
    "mi355_lzss_batch_decompress": (_i32, [_vp]),
    "mi355_lzss_batch_download_decompressed": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "mi355_lzss_batch_stage_times": (_i32, [_vp, _vp, _vp, _i32]),
}

_lib = None


def lib():
    """Loads the HIP library; raises (loudly) when it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C compress_amd/csrc` "
                "(or __graft_entry__.build()); compress_amd has no CPU path")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    return (lib().mi355_lzss_last_error() or b"").decode()
