"""hwang.Decoder with the reference's semantics (python/hwang/decoder.py:5-69):
`retrieve(rows)` walks the seek plan through ONE DecoderAutomata and returns
H x W x 3 uint8 arrays, frame rows[k] at position k.  This is the drop-in path;
the batched multi-lane path is `hwang_amd.Decoder`.

Two things differ from the reference's 69 lines, both inside its C++ API:

* `num_devices` is a constructor argument (the reference hard-wires 1,
  decoder.py:28).  It reaches the backend through
  VideoDecoderFactory::make_from_config exactly as in the reference
  (video_decoder_factory.cpp:90-91), where the software backend reads it as its
  decoder thread count (software_video_decoder.cpp:148); the AMD backend reads it
  the same way: that many threads entropy decode consecutive pictures at once.
* the intervals of the seek plan go to `DecoderAutomata.initialize` in groups
  (a list of EncodedData per call - what the C++ API takes and the reference's own
  tests pass, decoder_automata_test.cpp:82-155) instead of one per call
  (decoder.py:65).  Every `get_frames` ends with wait_until_frames_copied(), i.e.
  with the GPU drained: per interval that costs the full latency of the interval's
  dependent pictures every time, per group it is paid once.  `interval_group_bytes`
  sizes the groups: the first is cut at a quarter of it (decoding starts early), the
  following ones at twice it, and an interval is never split, so a group holds up to
  2 x interval_group_bytes + one interval of encoded bytes.  Three groups exist at a
  time (one decoding, one queued, one being read): about 6 x interval_group_bytes
  (768 MB at the 128 MB default) of encoded video is resident beside the answer.
  0 restores the reference's loop (one interval per initialize)."""
import bisect

import hwang

from ._python import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData, VideoDecoderType,
                      slice_into_video_intervals)


class Decoder(object):
    def __init__(self, f_or_path, video_index=None, device_type=DeviceType.CPU, device_id=0, decoder_type=None,
                 num_devices=1, interval_group_bytes=128 << 20):
        self.video_index = video_index if video_index is not None else hwang.index_video(f_or_path)
        self.f = open(f_or_path, "rb") if isinstance(f_or_path, str) else f_or_path
        handle = DeviceHandle()
        handle.type = device_type
        handle.id = device_id
        if decoder_type is None:
            # reference: GPU -> NVIDIA, CPU -> SOFTWARE.  This build: GPU -> AMD; a CPU handle has no backend
            # and the constructor below raises.
            decoder_type = VideoDecoderType.AMD if device_type == DeviceType.GPU else VideoDecoderType.SOFTWARE
        self._decoder = DecoderAutomata(handle, num_devices, decoder_type)
        self._group_bytes = interval_group_bytes

    def _segment(self, index, offsets, sizes, keyframes, end_of_file, first, last, wanted):
        begin = offsets[first]
        end = offsets[last] if last < len(offsets) else end_of_file
        self.f.seek(begin)
        segment = EncodedData()
        segment.encoded_video = self.f.read(end - begin)
        segment.width, segment.height, segment.format = index.frame_width(), index.frame_height(), index.format()
        segment.start_keyframe, segment.end_keyframe = first, last
        segment.sample_offsets = [o - begin for o in offsets[first:last]]
        segment.sample_sizes = sizes[first:last]
        # (the sync samples first <= k <= last; a scan of the whole table per interval was 0.1 ms of a one-hour clip's 9 ms)
        segment.keyframes = keyframes[bisect.bisect_left(keyframes, first):bisect.bisect_right(keyframes, last)]
        segment.valid_frames = wanted
        return segment, end - begin

    def _groups(self, rows):
        """The seek plan as lists of EncodedData, each list up to interval_group_bytes of encoded video (one interval
        per list with 0): [(segments, frames wanted of them)]."""
        index = self.video_index
        offsets = index.sample_offsets()
        sizes = index.sample_sizes()
        keyframes = index.keyframe_indices()
        end_of_file = offsets[-1] + sizes[-1]
        group, group_frames, group_bytes = [], 0, 0
        # the first group is a quarter of the bound (decoding starts while the helper thread reads on), the ones behind
        # it twice the bound (each group ends with the GPU drained: fewer of them, fewer bubbles)
        limit = self._group_bytes // 4
        for (first, last), wanted in slice_into_video_intervals(index, rows):
            if not wanted:   # (the planner's answer to an empty row list: one interval without frames)
                continue
            segment, nbytes = self._segment(index, offsets, sizes, keyframes, end_of_file, first, last, wanted)
            group.append(segment)
            group_frames += len(wanted)
            group_bytes += nbytes
            if group_bytes >= limit:
                yield group, group_frames
                group, group_frames, group_bytes = [], 0, 0
                limit = self._group_bytes * 2
        if group:
            yield group, group_frames

    def retrieve(self, rows):
        extradata = self.video_index.metadata_bytes()
        if not self._group_bytes:
            # the reference's loop as it stands (decoder.py:39-67): read an interval, initialize, fetch its frames -
            # one after the other on this thread
            frames = []
            for group, nframes in self._groups(rows):
                self._decoder.initialize(group, extradata, True)
                frames += self._decoder.get_frames(self.video_index, nframes)
            return frames
        # The next group is read and packed (file reads, EncodedData) by a helper thread while the current one decodes:
        # the extension releases the GIL inside initialize / get_frames.
        import queue
        import threading
        ready = queue.Queue(maxsize=1)
        stop = threading.Event()

        def put(item):
            # (every hand-over gives way to `stop`: the consumer may have left with an error while the queue is full -
            # a plain put() of the end marker then blocked forever and retrieve() never returned from its join)
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for item in self._groups(rows):
                    if not put(item):
                        return
                put(None)
            except BaseException as e:  # noqa: BLE001  (handed to the caller below)
                put(e)

        producer = threading.Thread(target=produce, name="hwang-read", daemon=True)
        producer.start()
        frames = []
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                group, nframes = item
                self._decoder.initialize(group, extradata, True)   # (consume: the group's EncodedData are ours to give)
                frames += self._decoder.get_frames(self.video_index, nframes)
        finally:
            stop.set()
            producer.join()
        return frames
