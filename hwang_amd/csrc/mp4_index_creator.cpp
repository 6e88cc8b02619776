#include "mp4_index_creator.h"

#include <cstring>
#include <functional>

namespace hwang {
namespace {

constexpr uint32_t fourcc(const char (&s)[5]) {
  return (uint32_t(uint8_t(s[0])) << 24) | (uint32_t(uint8_t(s[1])) << 16) |
         (uint32_t(uint8_t(s[2])) << 8) | uint32_t(uint8_t(s[3]));
}

std::string fourcc_str(uint32_t t) {
  std::string s(4, ' ');
  for (int i = 0; i < 4; ++i) s[i] = static_cast<char>(t >> (24 - 8 * i));
  return s;
}

// Bounds-checked big-endian cursor.  Reading past the end sets `bad` and
// yields zeros; callers test `bad` once per box.
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool bad = false;

  uint64_t left() const { return static_cast<uint64_t>(end - p); }
  uint64_t be(int nbytes) {
    if (left() < static_cast<uint64_t>(nbytes)) {
      bad = true;
      p = end;
      return 0;
    }
    uint64_t v = 0;
    for (int i = 0; i < nbytes; ++i) v = (v << 8) | *p++;
    return v;
  }
  uint32_t u8() { return static_cast<uint32_t>(be(1)); }
  uint32_t u16() { return static_cast<uint32_t>(be(2)); }
  uint32_t u32() { return static_cast<uint32_t>(be(4)); }
  uint64_t u64() { return be(8); }
  void skip(uint64_t n) {
    if (left() < n) {
      bad = true;
      p = end;
    } else {
      p += n;
    }
  }
};

struct Box {
  uint32_t type = 0;
  const uint8_t* start = nullptr;  // first byte of the size field
  uint64_t size = 0;               // whole box
  uint64_t header = 0;             // bytes before the payload
  const uint8_t* body() const { return start + header; }
  uint64_t body_size() const { return size - header; }
};

// Reads a box header at [p, end).  `size == 0` ("to end of enclosing
// container") is resolved against `end`.  Returns false when not even the
// header is available.  The box may extend past `end` (caller decides).
bool read_box_header(const uint8_t* p, const uint8_t* end, Box& b) {
  Cursor c{p, end};
  b.start = p;
  b.size = c.u32();
  b.type = c.u32();
  if (b.size == 1) b.size = c.u64();
  if (b.type == fourcc("uuid")) c.skip(16);
  if (c.bad) return false;
  b.header = static_cast<uint64_t>(c.p - p);
  if (b.size == 0) b.size = static_cast<uint64_t>(end - p);
  return b.size >= b.header;
}

// Calls fn(box) for every child box fully contained in [body, body+size);
// stops early when fn returns false.
void for_each_child(const uint8_t* body, uint64_t size,
                    const std::function<bool(const Box&)>& fn) {
  const uint8_t* p = body;
  const uint8_t* end = body + size;
  while (end - p >= 8) {
    Box b;
    if (!read_box_header(p, end, b)) return;
    if (b.size > static_cast<uint64_t>(end - p)) return;
    if (!fn(b)) return;
    p += b.size;
  }
}

bool find_child(const uint8_t* body, uint64_t size, uint32_t type, Box& out) {
  bool found = false;
  for_each_child(body, size, [&](const Box& b) {
    if (b.type != type) return true;
    out = b;
    found = true;
    return false;
  });
  return found;
}

// version(8) + flags(24) of a FullBox payload.
struct FullHeader {
  uint32_t version;
  uint32_t flags;
};
FullHeader read_full(Cursor& c) {
  uint32_t v = c.u32();
  return {v >> 24, v & 0xffffff};
}

}  // namespace

MP4IndexCreator::MP4IndexCreator(uint64_t file_size) : file_size_(file_size) {}

bool MP4IndexCreator::fail(const std::string& msg) {
  if (!error_) error_message_ = msg;
  error_ = true;
  done_ = true;
  return false;
}

bool MP4IndexCreator::want_exact(uint64_t offset, uint64_t size,
                                 uint64_t& next_offset, uint64_t& next_size) {
  // overflow-safe: a 64-bit box size close to 2^64 must not wrap past the test
  if (offset > file_size_ || size > file_size_ - offset) return fail("EOF in middle of box");
  next_offset = offset;
  next_size = size;
  return true;
}

bool MP4IndexCreator::want_probe(uint64_t offset, uint64_t& next_offset,
                                 uint64_t& next_size) {
  uint64_t size = 1024;
  if (offset + size > file_size_) {
    size = offset >= file_size_ ? 0 : file_size_ - offset;
    if (size == 0) {
      if (parsed_ftyp_ && parsed_moov_ && fragments_present_) {
        done_ = true;  // ran out of moofs: the fragmented file is indexed
        return false;
      }
      return fail("Reached EOF without being done");
    }
  }
  next_offset = offset;
  next_size = size;
  return true;
}

bool MP4IndexCreator::feed(const uint8_t* data, size_t size,
                           uint64_t& next_offset, uint64_t& next_size) {
  const uint8_t* p = data;
  const uint8_t* end = data + size;
  while (p < end && !is_done()) {
    Box b;
    if (!read_box_header(p, end, b)) {
      // Header straddles the chunk end: ask again from this box.
      if (pos_ + static_cast<uint64_t>(end - p) >= file_size_) {
        return fail("Truncated box header at end of file");
      }
      return want_probe(pos_, next_offset, next_size);
    }
    const uint64_t have = static_cast<uint64_t>(end - p);
    const bool is_ftyp = !parsed_ftyp_ && b.type == fourcc("ftyp");
    const bool is_moov = !parsed_moov_ && b.type == fourcc("moov");
    const bool is_moof = b.type == fourcc("moof");
    if (is_ftyp || is_moov || is_moof) {
      if (have < b.size) return want_exact(pos_, b.size, next_offset, next_size);
      bool ok = is_ftyp   ? parse_ftyp(p, b.size, b.header)
                : is_moov ? parse_moov(p, b.size, b.header)
                          : parse_moof(p, b.size, b.header, pos_);
      if (!ok) return false;
      p += b.size;
      pos_ += b.size;
    } else {
      pos_ += b.size;
      return want_probe(pos_, next_offset, next_size);
    }
  }
  if (is_done()) return false;
  return want_probe(pos_, next_offset, next_size);
}

bool MP4IndexCreator::parse_ftyp(const uint8_t* box, uint64_t box_size,
                                 uint64_t header) {
  Cursor c{box + header, box + box_size};
  c.u32();  // major_brand (the reference only tests compatible_brands)
  c.u32();  // minor_version
  bool supported = false;
  std::string listed;
  while (c.left() >= 4) {
    uint32_t brand = c.u32();
    listed += fourcc_str(brand) + ", ";
    if (brand == fourcc("isom") || brand == fourcc("iso2") ||
        brand == fourcc("avc1")) {
      supported = true;
    }
  }
  if (!supported) return fail("No supported mp4 brands: " + listed);
  parsed_ftyp_ = true;
  return true;
}

bool MP4IndexCreator::parse_moov(const uint8_t* box, uint64_t box_size,
                                 uint64_t header) {
  const uint8_t* moov = box + header;
  const uint64_t moov_size = box_size - header;

  // First 'trak' whose mdia/hdlr handler is 'vide'.
  bool saw_trak = false, have_video = false;
  Box video_mdia;
  for_each_child(moov, moov_size, [&](const Box& trak) {
    if (trak.type != fourcc("trak")) return true;
    saw_trak = true;
    Box mdia, hdlr;
    if (!find_child(trak.body(), trak.body_size(), fourcc("mdia"), mdia))
      return true;
    if (!find_child(mdia.body(), mdia.body_size(), fourcc("hdlr"), hdlr))
      return true;
    Cursor c{hdlr.body(), hdlr.body() + hdlr.body_size()};
    read_full(c);
    c.u32();  // pre_defined
    uint32_t handler = c.u32();
    if (c.bad || handler != fourcc("vide")) return true;
    video_mdia = mdia;
    have_video = true;
    return false;
  });
  if (!saw_trak) return fail("Could not find a trak box");
  if (!have_video) return fail("Could not find a video trak file");

  Box mdhd;
  if (!find_child(video_mdia.body(), video_mdia.body_size(), fourcc("mdhd"),
                  mdhd)) {
    return fail("Could not find 'mdhd'");
  }
  {
    Cursor c{mdhd.body(), mdhd.body() + mdhd.body_size()};
    FullHeader fh = read_full(c);
    const int w = fh.version == 1 ? 8 : 4;
    c.be(w);  // creation_time
    c.be(w);  // modification_time
    timescale_ = c.u32();
    duration_ = c.be(w);
    if (c.bad) return fail("Truncated 'mdhd'");
  }

  Box minf, stbl;
  if (!find_child(video_mdia.body(), video_mdia.body_size(), fourcc("minf"),
                  minf) ||
      !find_child(minf.body(), minf.body_size(), fourcc("stbl"), stbl)) {
    return fail("Could not parse 'stbl' correctly");
  }
  if (!parse_stbl(stbl.body(), stbl.body_size())) {
    return fail("Could not parse 'stbl' correctly");
  }

  // 'mvex' announces movie fragments; remember every trex's defaults.
  Box mvex;
  if (find_child(moov, moov_size, fourcc("mvex"), mvex)) {
    fragments_present_ = true;
    bool leva = false;
    for_each_child(mvex.body(), mvex.body_size(), [&](const Box& b) {
      if (b.type == fourcc("leva")) leva = true;
      if (b.type != fourcc("trex")) return true;
      Cursor c{b.body(), b.body() + b.body_size()};
      read_full(c);
      TrackDefaults t;
      t.track_id = c.u32();
      c.u32();  // default_sample_description_index
      c.u32();  // default_sample_duration
      t.sample_size = c.u32();
      t.sample_flags = c.u32();
      if (!c.bad) trex_.push_back(t);
      return true;
    });
    if (leva) return fail("leva not supported");
  }
  parsed_moov_ = true;
  return true;
}

bool MP4IndexCreator::parse_stbl(const uint8_t* body, uint64_t body_size) {
  Box b;

  // --- sample sizes: stsz, overridden by stz2 when both exist --------------
  std::vector<uint64_t> sizes;
  bool have_sizes = false;
  if (find_child(body, body_size, fourcc("stsz"), b)) {
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    uint32_t fixed = c.u32();
    uint32_t count = c.u32();
    if (c.bad || (fixed == 0 && c.left() / 4 < count)) return fail("Bad 'stsz'");
    // every sample occupies file bytes: a fixed-size table cannot describe more than the file holds
    // (an unchecked count of 2^32-1 was a 32 GB allocation)
    if (fixed != 0 && static_cast<uint64_t>(count) > file_size_ / fixed) return fail("Bad 'stsz'");
    sizes.resize(count, fixed);
    if (fixed == 0)
      for (uint32_t i = 0; i < count; ++i) sizes[i] = c.u32();
    have_sizes = true;
  }
  if (find_child(body, body_size, fourcc("stz2"), b)) {
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    c.skip(3);
    uint32_t field = c.u8();
    uint32_t count = c.u32();
    if (c.bad || (field != 4 && field != 8 && field != 16) ||
        c.left() * 8 / field < count) {
      return fail("Bad 'stz2'");
    }
    sizes.assign(count, 0);
    for (uint32_t i = 0; i < count; ++i) {
      if (field == 4) {
        uint32_t byte = c.p[i / 2];
        sizes[i] = (i & 1) ? (byte & 0xf) : (byte >> 4);
      } else {
        sizes[i] = c.be(field / 8);
      }
    }
    have_sizes = true;
  }
  if (!have_sizes) return fail("Could not find 'stsz' or 'stz2'");
  const size_t n = sizes.size();

  // --- sample -> chunk: stsc run-length table ------------------------------
  // samples_in_chunk[k] for chunk k (0-based); the last run repeats until
  // every sample is placed.
  if (!find_child(body, body_size, fourcc("stsc"), b))
    return fail("Could not find 'stsc'");
  std::vector<uint32_t> samples_in_chunk;
  {
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    uint32_t entries = c.u32();
    if (c.bad || c.left() / 12 < entries) return fail("Bad 'stsc'");
    uint64_t placed = 0;
    uint32_t run_first = 0, run_len = 0;
    for (uint32_t i = 0; i < entries; ++i) {
      uint32_t first_chunk = c.u32();
      uint32_t per_chunk = c.u32();
      c.u32();  // sample_description_index
      if (run_first != 0) {
        if (first_chunk < run_first) return fail("Bad 'stsc'");
        // a run of empty chunks places nothing: it would spin through up to 2^32 chunks
        if (run_len == 0 && first_chunk > run_first && placed < n) return fail("Bad 'stsc'");
        for (uint32_t k = run_first; k < first_chunk && placed < n; ++k) {
          samples_in_chunk.push_back(run_len);
          placed += run_len;
        }
      }
      run_first = first_chunk;
      run_len = per_chunk;
    }
    if (placed < n && run_len == 0) return fail("Bad 'stsc'");
    while (placed < n) {
      samples_in_chunk.push_back(run_len);
      placed += run_len;
    }
  }

  // --- chunk offsets: stco, overridden by co64 -----------------------------
  std::vector<uint64_t> chunk_offsets;
  bool have_chunks = false;
  for (int pass = 0; pass < 2; ++pass) {
    const int w = pass == 0 ? 4 : 8;
    if (!find_child(body, body_size, pass == 0 ? fourcc("stco") : fourcc("co64"),
                    b)) {
      continue;
    }
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    uint32_t count = c.u32();
    if (c.bad || c.left() / w < count) return fail("Bad chunk offset box");
    chunk_offsets.resize(count);
    for (uint32_t i = 0; i < count; ++i) chunk_offsets[i] = c.be(w);
    have_chunks = true;
  }
  if (!have_chunks) return fail("Could not find 'stco' or 'co64'");

  std::vector<uint64_t> offsets;
  offsets.reserve(n);
  {
    size_t s = 0;
    for (size_t k = 0; k < samples_in_chunk.size() && s < n; ++k) {
      if (k >= chunk_offsets.size()) return fail("'stsc' names a chunk 'stco' lacks");
      uint64_t at = chunk_offsets[k];
      for (uint32_t j = 0; j < samples_in_chunk[k] && s < n; ++j, ++s) {
        offsets.push_back(at);
        at += sizes[s];
      }
    }
  }

  // --- sync samples: stss (1-based); absent => every sample ----------------
  std::vector<uint64_t> keyframes;
  if (find_child(body, body_size, fourcc("stss"), b)) {
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    uint32_t count = c.u32();
    if (c.bad || c.left() / 4 < count) return fail("Bad 'stss'");
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t number = c.u32();  // 1-based, strictly increasing (ISO/IEC 14496-12 8.6.2)
      if (number == 0 || number > n || (!keyframes.empty() && number - 1 <= keyframes.back()))
        return fail("Bad 'stss'");
      keyframes.push_back(number - 1);
    }
  } else {
    for (size_t i = 0; i < n; ++i) keyframes.push_back(sample_offsets_.size() + i);
  }

  // --- stsd: VisualSampleEntry -> display size, format, codec config -------
  if (!find_child(body, body_size, fourcc("stsd"), b))
    return fail("Could not find 'stsd'");
  {
    Cursor c{b.body(), b.body() + b.body_size()};
    read_full(c);
    uint32_t entries = c.u32();
    const uint8_t* p = c.p;
    const uint8_t* end = b.body() + b.body_size();
    for (uint32_t i = 0; i < entries && end - p >= 8; ++i) {
      Box e;
      if (!read_box_header(p, end, e) || e.size > uint64_t(end - p)) break;
      // SampleEntry(6 reserved + dref idx) + VisualSampleEntry fixed part.
      Cursor v{e.body(), e.body() + e.body_size()};
      v.skip(6 + 2 + 2 + 2 + 12);
      uint32_t w = v.u16();
      uint32_t h = v.u16();
      v.skip(4 + 4 + 4 + 2 + 32 + 2 + 2);
      if (v.bad) return fail("Truncated sample entry");
      width_ = static_cast<uint32_t>(static_cast<int16_t>(w));
      height_ = static_cast<uint32_t>(static_cast<int16_t>(h));
      format_ = fourcc_str(e.type);
      uint32_t config = 0;
      if (e.type == fourcc("avc1") || e.type == fourcc("avc3"))
        config = fourcc("avcC");
      if (e.type == fourcc("hev1") || e.type == fourcc("hvc1"))
        config = fourcc("hvcC");
      Box cfg;
      if (config && find_child(v.p, v.left(), config, cfg)) {
        extradata_.assign(cfg.body(), cfg.body() + cfg.body_size());
      }
      p += e.size;
    }
  }

  sample_offsets_.insert(sample_offsets_.end(), offsets.begin(), offsets.end());
  sample_sizes_.insert(sample_sizes_.end(), sizes.begin(),
                       sizes.begin() + offsets.size());
  keyframe_indices_.insert(keyframe_indices_.end(), keyframes.begin(),
                           keyframes.end());
  return true;
}

bool MP4IndexCreator::parse_moof(const uint8_t* box, uint64_t box_size,
                                 uint64_t header, uint64_t moof_file_offset) {
  bool first_traf = true;
  uint64_t prev_traf_end = 0;
  bool ok = true;

  for_each_child(box + header, box_size - header, [&](const Box& traf) {
    if (traf.type != fourcc("traf")) return true;
    Box tfhd;
    if (!find_child(traf.body(), traf.body_size(), fourcc("tfhd"), tfhd)) {
      ok = fail("Could not find 'tfhd'");
      return false;
    }
    Cursor c{tfhd.body(), tfhd.body() + tfhd.body_size()};
    FullHeader fh = read_full(c);
    const uint32_t track_id = c.u32();
    uint64_t base = 0;
    uint32_t def_size = 0, def_flags = 0;
    const bool base_given = fh.flags & 0x000001;
    const bool size_given = fh.flags & 0x000010;
    const bool flags_given = fh.flags & 0x000020;
    if (base_given) base = c.u64();
    if (fh.flags & 0x000002) c.u32();  // sample_description_index
    if (fh.flags & 0x000008) c.u32();  // default_sample_duration
    if (size_given) def_size = c.u32();
    if (flags_given) def_flags = c.u32();
    if (c.bad) {
      ok = fail("Truncated 'tfhd'");
      return false;
    }
    if (!base_given) {
      const bool base_is_moof = fh.flags & 0x020000;
      base = (base_is_moof || first_traf) ? moof_file_offset : prev_traf_end;
    }

    const TrackDefaults* trex = nullptr;
    for (const auto& t : trex_)
      if (t.track_id == track_id) trex = &t;
    if (!trex) {
      ok = fail("Could not find 'trex' for track id in 'tfhd'");
      return false;
    }

    uint64_t run_end = base;
    for_each_child(traf.body(), traf.body_size(), [&](const Box& trun) {
      if (trun.type != fourcc("trun")) return true;
      Cursor r{trun.body(), trun.body() + trun.body_size()};
      FullHeader th = read_full(r);
      uint32_t count = r.u32();
      if (r.bad || count > r.left()) {  // even flag-less runs are bounded by the file through the sizes below
        ok = fail("Truncated 'trun'");
        return false;
      }
      uint64_t at = run_end;
      if (th.flags & 0x000001)
        at = base + static_cast<int64_t>(static_cast<int32_t>(r.u32()));
      uint32_t first_flags = 0;
      const bool has_first = th.flags & 0x000004;
      if (has_first) first_flags = r.u32();
      const bool s_dur = th.flags & 0x000100, s_size = th.flags & 0x000200,
                 s_flags = th.flags & 0x000400, s_cto = th.flags & 0x000800;
      for (uint32_t i = 0; i < count && !r.bad; ++i) {
        if (s_dur) r.u32();
        uint64_t sz = s_size ? r.u32() : size_given ? def_size : trex->sample_size;
        uint32_t fl = s_flags                ? r.u32()
                      : (i == 0 && has_first) ? first_flags
                      : flags_given           ? def_flags
                                              : trex->sample_flags;
        if (s_cto) r.u32();
        if (r.bad) break;
        // sample_is_non_sync_sample is bit 16 of the flags word.
        if ((fl & 0x00010000) == 0) keyframe_indices_.push_back(sample_sizes_.size());
        sample_offsets_.push_back(at);
        sample_sizes_.push_back(sz);
        at += sz;
      }
      if (r.bad) {
        ok = fail("Truncated 'trun'");
        return false;
      }
      run_end = at;
      return true;
    });
    prev_traf_end = run_end;
    first_traf = false;
    return ok;
  });
  return ok;
}

VideoIndex MP4IndexCreator::get_video_index() {
  return VideoIndex(timescale_, duration_, width_, height_, format_,
                    sample_offsets_, sample_sizes_, keyframe_indices_,
                    extradata_);
}

}  // namespace hwang
