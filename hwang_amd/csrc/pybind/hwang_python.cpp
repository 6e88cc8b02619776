// pybind11 module `hwang._python`: the reference's Python boundary
// (hwang/hwang_python.cpp:102-167) over THIS repo's C++ classes, so that code
// written against scanner-research/hwang (python/hwang/decoder.py:24-28 picks a
// backend through hwang._python.VideoDecoderType) can select the MI355X
// backend: VideoDecoderType.AMD is the one addition to the surface.
//
// Same names, argument meaning and error behaviour as the reference module:
//   VideoIndex (13 methods), MP4IndexCreator(file_size).feed(data, size) ->
//   (more, next_offset, next_size), slice_into_video_intervals(index, rows) ->
//   [((start, end), [frames])], DeviceType, DeviceHandle, VideoDecoderType,
//   EncodedData (10 fields), DecoderAutomata(handle, num_devices, type)
//   .initialize(list, extradata) / .get_frames(index, n) -> [ndarray(H, W, 3)].
// Differences, all on the safe side: a failing backend raises RuntimeError
// from the constructor instead of dereferencing null later; frames are views
// into ONE buffer per call kept alive by a capsule (the reference mallocs and
// copies each frame twice, hwang_python.cpp:75-94); get_frames releases the
// GIL while the backend works.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdlib.h>
#include <sys/mman.h>

#include <cstdlib>

#include "../decoder_automata.h"
#include "../mp4_index_creator.h"
#include "../video_decoder_factory.h"
#include "../video_index.h"

#include <memory>
#include <stdexcept>

namespace py = pybind11;
using namespace hwang;

namespace {

py::bytes index_serialize(const VideoIndex& index) {
  std::vector<uint8_t> wire = index.serialize();
  return py::bytes(reinterpret_cast<const char*>(wire.data()), wire.size());
}

VideoIndex index_deserialize(const std::string& data) {
  VideoIndex out;
  if (!VideoIndex::deserialize(reinterpret_cast<const uint8_t*>(data.data()), data.size(), &out))
    throw std::runtime_error("VideoIndex.deserialize: malformed data");
  return out;
}

py::bytes index_metadata(const VideoIndex& index) {
  const std::vector<uint8_t>& m = index.metadata_bytes();
  return py::bytes(reinterpret_cast<const char*>(m.data()), m.size());
}

std::tuple<bool, uint64_t, uint64_t> indexer_feed(MP4IndexCreator& indexer, const std::string& data, size_t size) {
  py::gil_scoped_release release;
  uint64_t next_offset = 0, next_size = 0;
  const bool more = indexer.feed(reinterpret_cast<const uint8_t*>(data.data()), std::min(size, data.size()),
                                 next_offset, next_size);
  return std::make_tuple(more, next_offset, next_size);
}

std::vector<std::tuple<std::tuple<uint64_t, uint64_t>, std::vector<uint64_t>>> plan(const VideoIndex& index,
                                                                                    const std::vector<uint64_t>& rows) {
  bool ok = true;
  VideoIntervals v = slice_into_video_intervals(index, rows, &ok);
  if (!ok) throw std::runtime_error("slice_into_video_intervals: row out of range or inconsistent index");
  std::vector<std::tuple<std::tuple<uint64_t, uint64_t>, std::vector<uint64_t>>> out;
  for (size_t i = 0; i < v.sample_index_intervals.size(); ++i)
    out.emplace_back(std::make_tuple(std::get<0>(v.sample_index_intervals[i]), std::get<1>(v.sample_index_intervals[i])),
                     v.valid_frames[i]);
  return out;
}

std::unique_ptr<DecoderAutomata> make_automata(DeviceHandle handle, int32_t num_devices, VideoDecoderType type) {
  DecoderAutomata* a = DecoderAutomata::make_instance(handle, num_devices, type);
  if (!a)
    throw std::runtime_error(
        "DecoderAutomata: this build has no backend for the requested decoder type on that device "
        "(only VideoDecoderType.AMD on a GPU handle; there is no CPU fallback)");
  return std::unique_ptr<DecoderAutomata>(a);
}

void automata_initialize(DecoderAutomata& dec, const py::list& segment_list, const py::object& extradata, bool consume) {
  // The Python objects keep their data (the reference's API leaves them with the caller); the one copy made here is
  // handed on to the automata, which used to copy it once more.  consume=True (not in the reference; hwang.Decoder
  // builds its EncodedData per call and drops them) hands the encoded bytes over instead: the objects are left
  // without them.
  std::vector<DecoderAutomata::EncodedData> segments;
  segments.reserve(segment_list.size());
  for (const py::handle& h : segment_list) {
    DecoderAutomata::EncodedData& e = h.cast<DecoderAutomata::EncodedData&>();
    if (consume) {
      segments.push_back(std::move(e));
      e = DecoderAutomata::EncodedData();
    } else {
      segments.push_back(e);
    }
  }
  std::vector<uint8_t> extra;
  if (py::isinstance<py::bytes>(extradata) || py::isinstance<py::bytearray>(extradata)) {
    const std::string s = py::cast<std::string>(py::bytes(extradata));
    extra.assign(s.begin(), s.end());
  } else {
    extra = py::cast<std::vector<uint8_t>>(extradata);
  }
  Result r;
  {
    py::gil_scoped_release release;
    r = dec.initialize(std::move(segments), extra);
  }
  if (!r.ok) throw std::runtime_error(r.message);
}

std::vector<py::array_t<uint8_t>> automata_get_frames(DecoderAutomata& dec, const VideoIndex& index, uint32_t num_frames) {
  const size_t h = index.frame_height(), w = index.frame_width(), frame_size = w * h * 3;
  // One buffer for the frames of the call, NOT zero-filled: a std::vector would touch every page first (gigabytes
  // for a few hundred 1080p frames, a second of memset before the first sample is fed); every byte handed out below is
  // written by get_frames, and on failure nothing is handed out.
  // Large answers ask for transparent huge pages (where the kernel grants them on request): the frames land in pages
  // nobody has touched, and a 2 MB fault per 2 MB beats 512 faults of 4 KB under the copier threads.
  const size_t bytes = std::max<size_t>(1, frame_size * num_frames);
  uint8_t* storage = nullptr;
  constexpr size_t kHuge = size_t(2) << 20;
  if (bytes >= kHuge) {  // (one 1080p frame is three of them: 3 faults instead of 1 500 under the copy-out)
    void* p = nullptr;
    if (posix_memalign(&p, kHuge, bytes) == 0) {
      storage = static_cast<uint8_t*>(p);
      (void)madvise(p, bytes, MADV_HUGEPAGE);
    }
  }
  if (!storage) storage = static_cast<uint8_t*>(std::malloc(bytes));
  if (!storage) throw std::bad_alloc();
  py::capsule owner(storage, [](void* p) { std::free(p); });
  Result r;
  {
    py::gil_scoped_release release;
    r = dec.get_frames(storage, static_cast<int32_t>(num_frames));
  }
  if (!r.ok) throw std::runtime_error(r.message);
  std::vector<py::array_t<uint8_t>> frames;
  for (uint32_t i = 0; i < num_frames; ++i)
    frames.emplace_back(std::vector<py::ssize_t>{(py::ssize_t)h, (py::ssize_t)w, 3},
                        std::vector<py::ssize_t>{(py::ssize_t)(w * 3), 3, 1}, storage + i * frame_size, owner);
  return frames;
}

}  // namespace

PYBIND11_MODULE(_python, m) {
  m.doc() = "hwang C++ library (MI355X backend build)";
  m.attr("__name__") = "hwang._python";

  py::class_<VideoIndex>(m, "VideoIndex")
      .def_static("deserialize", &index_deserialize)
      .def("serialize", &index_serialize)
      .def("timescale", &VideoIndex::timescale)
      .def("duration", &VideoIndex::duration)
      .def("fps", &VideoIndex::fps)
      .def("frame_width", &VideoIndex::frame_width)
      .def("frame_height", &VideoIndex::frame_height)
      .def("format", &VideoIndex::format)
      .def("frames", &VideoIndex::frames)
      .def("sample_offsets", &VideoIndex::sample_offsets)
      .def("sample_sizes", &VideoIndex::sample_sizes)
      .def("keyframe_indices", &VideoIndex::keyframe_indices)
      .def("metadata_bytes", &index_metadata);

  py::class_<MP4IndexCreator>(m, "MP4IndexCreator")
      .def(py::init<uint64_t>())
      .def("feed", &indexer_feed)
      .def("is_done", &MP4IndexCreator::is_done)
      .def("is_error", &MP4IndexCreator::is_error)
      .def("error_message", &MP4IndexCreator::error_message)
      .def("get_video_index", &MP4IndexCreator::get_video_index);

  m.def("slice_into_video_intervals", &plan);

  py::enum_<DeviceType>(m, "DeviceType", py::arithmetic()).value("CPU", DeviceType::CPU).value("GPU", DeviceType::GPU);

  py::class_<DeviceHandle>(m, "DeviceHandle")
      .def(py::init([]() { return DeviceHandle{DeviceType::CPU, 0}; }))
      .def_readwrite("type", &DeviceHandle::type)
      .def_readwrite("id", &DeviceHandle::id);

  py::enum_<VideoDecoderType>(m, "VideoDecoderType", py::arithmetic())
      .value("SOFTWARE", VideoDecoderType::SOFTWARE)
      .value("NVIDIA", VideoDecoderType::NVIDIA)
      .value("INTEL", VideoDecoderType::INTEL)
      .value("AMD", VideoDecoderType::AMD);  // values >= 100: backends registered at run time (include/hwang_amd.h)

  py::class_<DecoderAutomata::EncodedData>(m, "EncodedData")
      .def(py::init<>())
      .def_property(
          "encoded_video",
          [](const DecoderAutomata::EncodedData& d) {
            return py::bytes(reinterpret_cast<const char*>(d.encoded_video.data()), d.encoded_video.size());
          },
          [](DecoderAutomata::EncodedData& d, const py::bytes& v) {  // straight from the bytes object: one copy, not two
            char* p = nullptr;
            py::ssize_t n = 0;
            if (PyBytes_AsStringAndSize(v.ptr(), &p, &n) != 0) throw py::error_already_set();
            d.encoded_video.assign(reinterpret_cast<const uint8_t*>(p), reinterpret_cast<const uint8_t*>(p) + n);
          })
      .def_readwrite("width", &DecoderAutomata::EncodedData::width)
      .def_readwrite("height", &DecoderAutomata::EncodedData::height)
      .def_readwrite("format", &DecoderAutomata::EncodedData::format)
      .def_readwrite("start_keyframe", &DecoderAutomata::EncodedData::start_keyframe)
      .def_readwrite("end_keyframe", &DecoderAutomata::EncodedData::end_keyframe)
      .def_readwrite("sample_offsets", &DecoderAutomata::EncodedData::sample_offsets)
      .def_readwrite("sample_sizes", &DecoderAutomata::EncodedData::sample_sizes)
      .def_readwrite("keyframes", &DecoderAutomata::EncodedData::keyframes)
      .def_readwrite("valid_frames", &DecoderAutomata::EncodedData::valid_frames);

  py::class_<DecoderAutomata>(m, "DecoderAutomata")
      .def(py::init(&make_automata))
      .def("initialize", &automata_initialize, py::arg("encoded_data"), py::arg("extradata"), py::arg("consume") = false)
      .def("get_frames", &automata_get_frames);

  m.def("supported_decoder_types", []() { return VideoDecoderFactory::get_supported_decoder_types(); });
}
