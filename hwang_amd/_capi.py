"""ctypes binding of libhwang_amd.so (C ABI in include/hwang_amd.h).

The library is the product; this file only declares prototypes.  There is no
Python or CPU fallback: if the shared object is missing or a symbol is absent
the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HWANG_AMD_LIB") or os.path.join(_HERE, "libhwang_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "hwang_amd: %s is missing -- build it with `make lib` (or "
        "`python -c 'import __graft_entry__ as g; g.build()'`) at the repo "
        "root.  There is no fallback path." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class RetrieveRequestStruct(C.Structure):
    """hwang_retrieve_request (one clip of hwang_retriever_retrieve_batch)."""
    _fields_ = [
        ("index", C.c_void_p), ("file", C.c_void_p), ("file_size", C.c_size_t),
        ("rows", C.POINTER(C.c_uint64)), ("num_rows", C.c_size_t),
        ("out", C.c_void_p), ("out_capacity", C.c_size_t),
    ]


class EncodedDataStruct(C.Structure):
    """hwang_encoded_data (DecoderAutomata::EncodedData, decoder_automata.h:43-66)."""
    _fields_ = [
        ("encoded_video", vp), ("encoded_video_size", C.c_size_t),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("start_keyframe", C.c_uint64), ("end_keyframe", C.c_uint64),
        ("format", C.c_char_p),
        ("sample_offsets", u64p), ("sample_sizes", u64p),
        ("num_samples", C.c_size_t),
        ("keyframes", u64p), ("num_keyframes", C.c_size_t),
        ("valid_frames", u64p), ("num_valid_frames", C.c_size_t),
    ]


# name -> (restype, argtypes).  Mirrors include/hwang_amd.h one to one; the
# CPU test-suite checks that every prototype declared in the header is here
# and exported by the library.
PROTOTYPES = {
    "hwang_amd_version": (C.c_char_p, []),
    "hwang_register_decoder_backend": (C.c_int, [C.c_int, vp]),
    # VideoIndex
    "hwang_video_index_create": (vp, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.c_char_p, u64p, u64p, C.c_size_t, u64p,
                                      C.c_size_t, vp, C.c_size_t]),
    "hwang_video_index_destroy": (None, [vp]),
    "hwang_video_index_deserialize": (vp, [vp, C.c_size_t]),
    "hwang_video_index_serialize": (C.c_size_t, [vp, vp, C.c_size_t]),
    "hwang_video_index_timescale": (C.c_uint32, [vp]),
    "hwang_video_index_duration": (C.c_uint64, [vp]),
    "hwang_video_index_fps": (C.c_double, [vp]),
    "hwang_video_index_frame_width": (C.c_uint32, [vp]),
    "hwang_video_index_frame_height": (C.c_uint32, [vp]),
    "hwang_video_index_format": (C.c_char_p, [vp]),
    "hwang_video_index_frames": (C.c_uint64, [vp]),
    "hwang_video_index_problem": (C.c_char_p, [vp]),
    "hwang_video_index_sample_offsets": (u64p, [vp]),
    "hwang_video_index_sample_sizes": (u64p, [vp]),
    "hwang_video_index_num_keyframes": (C.c_size_t, [vp]),
    "hwang_video_index_keyframe_indices": (u64p, [vp]),
    "hwang_video_index_metadata_size": (C.c_size_t, [vp]),
    "hwang_video_index_metadata_bytes": (u8p, [vp]),
    # MP4IndexCreator
    "hwang_mp4_indexer_create": (vp, [C.c_uint64]),
    "hwang_mp4_indexer_destroy": (None, [vp]),
    "hwang_mp4_indexer_feed": (C.c_int, [vp, vp, C.c_size_t, u64p, u64p]),
    "hwang_mp4_indexer_is_done": (C.c_int, [vp]),
    "hwang_mp4_indexer_is_error": (C.c_int, [vp]),
    "hwang_mp4_indexer_error_message": (C.c_char_p, [vp]),
    "hwang_mp4_indexer_get_video_index": (vp, [vp]),
    # planner
    "hwang_slice_into_video_intervals": (vp, [vp, u64p, C.c_size_t]),
    "hwang_video_intervals_destroy": (None, [vp]),
    "hwang_video_intervals_count": (C.c_size_t, [vp]),
    "hwang_video_intervals_range": (None, [vp, C.c_size_t, u64p, u64p]),
    "hwang_video_intervals_num_valid": (C.c_size_t, [vp, C.c_size_t]),
    "hwang_video_intervals_valid": (u64p, [vp, C.c_size_t]),
    # decoder backend
    "hwang_decoder_type_supported": (C.c_int, [C.c_int]),
    "hwang_decoder_create": (vp, [C.c_int, C.c_int32, C.c_uint32, C.c_int]),
    "hwang_decoder_destroy": (None, [vp]),
    "hwang_decoder_configure": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_char_p, vp, C.c_size_t]),
    "hwang_decoder_feed": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "hwang_decoder_flush": (C.c_int, [vp]),
    "hwang_decoder_discard_frame": (C.c_int, [vp]),
    "hwang_decoder_get_frame": (C.c_int, [vp, vp, C.c_size_t]),
    "hwang_decoder_decoded_frames_buffered": (C.c_int, [vp]),
    "hwang_decoder_wait_until_frames_copied": (C.c_int, [vp]),
    "hwang_decoder_last_error": (C.c_char_p, [vp]),
    "hwang_decoder_set_wanted": (C.c_int, [vp, u64p, C.c_size_t]),
    "hwang_decoder_set_interval_keyframes": (C.c_int, [vp, u64p, C.c_size_t]),
    "hwang_decoder_get_frame_yuv": (C.c_int, [vp, vp, C.c_size_t]),
    # automata
    "hwang_automata_create": (vp, [C.c_int, C.c_int32, C.c_int32, C.c_int]),
    "hwang_automata_destroy": (None, [vp]),
    "hwang_automata_initialize": (C.c_int, [vp, C.POINTER(EncodedDataStruct), C.c_size_t, vp, C.c_size_t]),
    "hwang_automata_get_frames": (C.c_int, [vp, vp, C.c_int32]),
    "hwang_automata_last_error": (C.c_char_p, [vp]),
    "hwang_automata_stats": (None, [vp, i64p]),
    # parallel retriever
    "hwang_retriever_create": (vp, [C.c_int, C.c_int32, C.c_int, C.c_int32]),
    "hwang_retriever_create_multi": (vp, [C.c_int, i32p, C.c_size_t, C.c_int, C.c_int32]),
    "hwang_retriever_set_shard": (C.c_int, [vp, C.c_int32, C.c_int32]),
    "hwang_retriever_num_lanes": (C.c_int32, [vp]),
    "hwang_retriever_retrieve_batch": (C.c_int, [vp, C.POINTER(RetrieveRequestStruct), C.c_size_t, C.c_int]),
    "hwang_retriever_device_stats": (None, [vp, C.c_int32, i64p]),
    "hwang_retriever_destroy": (None, [vp]),
    "hwang_retriever_retrieve": (C.c_int, [vp, vp, vp, C.c_size_t, u64p, C.c_size_t, vp, C.c_size_t, C.c_int]),
    "hwang_retriever_last_error": (C.c_char_p, [vp]),
    "hwang_retriever_stats": (None, [vp, i64p]),
    "hwang_retriever_host_counters": (None, [vp, i64p]),
    # kernels
    "hwang_k_yuv420_to_rgb24": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "hwang_h264_probe": (C.c_int, [vp, C.c_size_t, vp, u64p, u64p, C.c_size_t, u64p, C.c_size_t,
                                   i64p, C.c_char_p, C.c_size_t]),
    "hwang_h264_probe_workers": (C.c_int, [vp, C.c_size_t, vp, u64p, u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, C.c_int,
                                           i64p, C.c_char_p, C.c_size_t]),
    "hwang_hevc_describe_config": (C.c_int, [vp, C.c_size_t, C.c_char_p, C.c_size_t]),
    "hwang_hevc_run_test_picture": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    # profiling
    "hwang_prof_enable": (None, [C.c_int]),
    "hwang_prof_num_kernels": (C.c_int, []),
    "hwang_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "hwang_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "hwang_prof_reset": (None, []),
    "hwang_shutdown": (None, []),
    "hwang_replay_keep": (C.c_int, [C.c_int, C.c_int]),
    "hwang_replay_run": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    try:
        _fn = getattr(lib, _name)  # AttributeError here == library out of date
    except AttributeError:
        # development aid (tools/experiments/exp_ab_round.sh: an older build of the library under the current Python
        # for an A/B on one box): entry points the old build lacks stay unbound
        if os.environ.get("HWANG_AMD_LIB") and os.environ.get("HWANG_AMD_LIB_ALLOW_OLD"):
            continue
        raise
    _fn.restype = _res
    _fn.argtypes = _args


# Stop the executor threads while the interpreter (and the HIP runtime, and any
# profiler wrapped around it) is still fully alive.
import atexit  # noqa: E402

atexit.register(lib.hwang_shutdown)


def u64_array(values):
    values = list(values)
    return (C.c_uint64 * len(values))(*values)


def as_void_p(buf):
    """Host bytes-like / numpy array / int address -> c_void_p (+ keep-alive)."""
    if isinstance(buf, int):
        return C.c_void_p(buf), None
    if isinstance(buf, bytes):
        return buf, buf  # ctypes passes the bytes object's own buffer
    if isinstance(buf, (bytearray, memoryview)):
        keep = (C.c_char * len(buf)).from_buffer(buf)
        return C.cast(keep, C.c_void_p), keep
    # numpy
    return C.c_void_p(buf.ctypes.data), buf
