"""VideoIndex.from_file / to_file (python/hwang/video_index.py:5-15): the
serialized form is the reference's proto3 wire format."""
from ._python import VideoIndex


def _from_file(f):
    return VideoIndex.deserialize(f.read())


def _to_file(self, f):
    f.write(self.serialize())


VideoIndex.from_file = staticmethod(_from_file)
VideoIndex.to_file = _to_file
