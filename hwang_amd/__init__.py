"""hwang_amd -- MI355X-native sparse H.264 frame retrieval behind
scanner-research/hwang's API.

Mirrors python/hwang/__init__.py:1-26 (index_video) and re-exports the
`_python` surface plus Decoder.
"""
import os

from .api import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData,
                  MP4IndexCreator, VideoDecoderType, VideoIndex,
                  slice_into_video_intervals)
from .decoder import BatchRetriever, Decoder


def index_video(f_or_string):
    """Build a VideoIndex by driving MP4IndexCreator's pull protocol
    (python/hwang/__init__.py:5-26)."""
    def w(f):
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(0, 0)
        indexer = MP4IndexCreator(size)
        offset = 0
        size_to_read = 1024
        while not indexer.is_done():
            f.seek(offset, 0)
            data = f.read(size_to_read)
            ret, offset, new_size = indexer.feed(data, size_to_read)
            size_to_read = new_size
        if indexer.is_error():
            raise Exception(indexer.error_message())
        return indexer.get_video_index()

    if isinstance(f_or_string, str):
        with open(f_or_string, 'rb') as f:
            return w(f)
    return w(f_or_string)
