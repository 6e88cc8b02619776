"""hwang.Decoder with the reference's semantics (python/hwang/decoder.py:5-69):
`retrieve(rows)` walks the seek plan one keyframe interval at a time through
ONE DecoderAutomata -- initialize, then get_frames for the interval's wanted
rows -- and returns H x W x 3 uint8 arrays, frame rows[k] at position k.
This is the drop-in path; the batched multi-lane path is `hwang_amd.Decoder`."""
import hwang

from ._python import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData, VideoDecoderType,
                      slice_into_video_intervals)


class Decoder(object):
    def __init__(self, f_or_path, video_index=None, device_type=DeviceType.CPU, device_id=0, decoder_type=None):
        self.video_index = video_index if video_index is not None else hwang.index_video(f_or_path)
        self.f = open(f_or_path, "rb") if isinstance(f_or_path, str) else f_or_path
        handle = DeviceHandle()
        handle.type = device_type
        handle.id = device_id
        if decoder_type is None:
            # reference: GPU -> NVIDIA, CPU -> SOFTWARE.  This build: GPU -> AMD; a CPU handle has no backend
            # and the constructor below raises.
            decoder_type = VideoDecoderType.AMD if device_type == DeviceType.GPU else VideoDecoderType.SOFTWARE
        self._decoder = DecoderAutomata(handle, 1, decoder_type)

    def retrieve(self, rows):
        index = self.video_index
        offsets = index.sample_offsets()
        sizes = index.sample_sizes()
        keyframes = index.keyframe_indices()
        end_of_file = offsets[-1] + sizes[-1]
        frames = []
        for (first, last), wanted in slice_into_video_intervals(index, rows):
            begin = offsets[first]
            end = offsets[last] if last < len(offsets) else end_of_file
            self.f.seek(begin)
            segment = EncodedData()
            segment.encoded_video = self.f.read(end - begin)
            segment.width, segment.height, segment.format = index.frame_width(), index.frame_height(), index.format()
            segment.start_keyframe, segment.end_keyframe = first, last
            segment.sample_offsets = [o - begin for o in offsets[first:last]]
            segment.sample_sizes = sizes[first:last]
            segment.keyframes = [k for k in keyframes if first <= k <= last]
            segment.valid_frames = wanted
            self._decoder.initialize([segment], index.metadata_bytes())
            frames += self._decoder.get_frames(index, len(wanted))
        return frames
