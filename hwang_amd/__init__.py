"""hwang_amd -- MI355X-native sparse H.264 frame retrieval behind
scanner-research/hwang's API.

The ctypes twin of the `hwang` package (which sits on the pybind11 module): `index_video`, the `_python` surface
re-hosted on the C ABI (api.py), and a Decoder whose retrieve() is ONE native call (decoder.py).
"""
import os

from .api import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData,
                  MP4IndexCreator, VideoDecoderType, VideoIndex,
                  slice_into_video_intervals)
from .decoder import BatchRetriever, Decoder


def index_video(source):
    """VideoIndex of an MP4 given as a path or a seekable binary file (the job of python/hwang/__init__.py:5-26):
    MP4IndexCreator pulls - it is handed the first KiB and then names the byte range it wants next, until it has
    seen `ftyp` + `moov` (+ every `moof`)."""
    if isinstance(source, str):
        with open(source, "rb") as f:
            return index_video(f)
    indexer = MP4IndexCreator(source.seek(0, os.SEEK_END))
    offset, want = 0, 1024
    while not indexer.is_done():
        source.seek(offset)
        _more, offset, want = indexer.feed(source.read(want), want)
    if indexer.is_error():
        raise Exception(indexer.error_message())
    return indexer.get_video_index()
