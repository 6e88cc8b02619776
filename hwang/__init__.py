"""`hwang`, the reference's Python package name, on the MI355X build.

Same surface as scanner-research/hwang's python/hwang (`index_video`,
`Decoder`, `VideoIndex.from_file / to_file`, and everything in the pybind11
module `hwang._python`), so code written against the reference imports and
runs unchanged; the GPU device type selects `VideoDecoderType.AMD`
(python/hwang/decoder.py:24-27 hard-wires NVIDIA there).  The extension
`hwang/_python*.so` is built by `make pymod` from
hwang_amd/csrc/pybind/hwang_python.cpp and links libhwang_amd.so; there is no
fallback if it is missing.
"""
import ctypes as _ctypes
import os

# libhwang_amd.so (and through it ROCm's HIP runtime) goes in with RTLD_GLOBAL before the extension that links it, as
# hwang_amd/_capi.py does: a Python extension is loaded RTLD_LOCAL, and a HIP runtime loaded that way is not shared with
# a PyTorch imported later in the same process - which then brings up its bundled copy, and whichever of the two
# initialises second finds no device ("no ROCm-capable device is detected").
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hwang_amd", "libhwang_amd.so")
if not os.path.exists(_lib):
    raise ImportError("hwang: %s is missing -- build it with `make lib pymod` at the repo root; there is no fallback" % _lib)
_ctypes.CDLL(_lib, mode=_ctypes.RTLD_GLOBAL)

from ._python import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData, MP4IndexCreator, VideoDecoderType,
                      VideoIndex, slice_into_video_intervals, supported_decoder_types)
from .video_index import *  # noqa: F401,F403  (attaches from_file / to_file)
from .decoder import Decoder  # noqa: E402


def index_video(source):
    """VideoIndex of an MP4 given as a path or a seekable binary file, by
    driving MP4IndexCreator's pull protocol (python/hwang/__init__.py:5-26:
    start with the first KiB, then read whatever range the indexer asks for)."""
    if isinstance(source, str):
        with open(source, "rb") as f:
            return index_video(f)
    size = source.seek(0, os.SEEK_END)
    indexer = MP4IndexCreator(size)
    offset, want = 0, 1024
    while not indexer.is_done():
        source.seek(offset)
        _more, offset, want = indexer.feed(source.read(want), want)
    if indexer.is_error():
        raise Exception(indexer.error_message())
    return indexer.get_video_index()
