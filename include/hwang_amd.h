/* hwang_amd.h -- C ABI of libhwang_amd.so, the MI355X-native sparse H.264
 * decode path behind scanner-research/hwang's plugin boundary.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, and is
 * what a foreign-function binding for this path would bind.  Each group names
 * the reference interface it replaces (paths relative to the reference
 * tree).  Functions returning `int` return 0 on success and non-zero on
 * failure; the failure text is available from the object's *_last_error /
 * *_error_message accessor (never thrown, never aborted).
 *
 * Threading: a decoder handle follows the reference's two-thread contract
 * (feed/flush on one thread, get/discard/buffered on another).  All other
 * handles are single-threaded objects.
 */
#ifndef HWANG_AMD_H_
#define HWANG_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* -- enums (hwang/common.h:20-23, hwang/video_decoder_factory.h:23-27) ---- */
enum { HWANG_DEVICE_CPU = 0, HWANG_DEVICE_GPU = 1 };
enum {
  HWANG_DECODER_SOFTWARE = 0, /* FFmpeg backend: not built here            */
  HWANG_DECODER_NVIDIA = 1,   /* NVDEC backend: not built here             */
  HWANG_DECODER_INTEL = 2,    /* dead in the reference                     */
  HWANG_DECODER_AMD = 3,      /* this library's MI355X backend             */
  HWANG_DECODER_FIRST_REGISTERED = 100 /* >= this: backends plugged in at run time, below */
};

/* -- Backend plug-in boundary ------------------------------------------------
 * The seven entry points of VideoDecoderInterface (hwang/video_decoder_interface.h:28-49)
 * as a table of C functions, so that a backend living outside this library can be
 * driven by its DecoderAutomata and ParallelRetriever: the mirror image of
 * INTEGRATION.md, where hwang's factory gains a case for this library.  The
 * reference's factory is fixed at compile time (video_decoder_factory.cpp:31-98).
 * Every function returns 0 on success; last_error(self) describes the last failure;
 * set_wanted (the sparse hint, hwang_decoder_set_wanted) may be NULL.  The test
 * suite's STAMP fixture (tests/fixtures/stamp_backend.c) is such a backend. */
typedef struct hwang_backend_vtable {
  void* (*create)(int device_type, int32_t device_id, uint32_t num_devices); /* NULL: cannot open */
  void (*destroy)(void* self);
  int (*configure)(void* self, uint32_t width, uint32_t height, const char* format,
                   const uint8_t* extradata, size_t extradata_size);
  int (*feed)(void* self, const uint8_t* encoded, size_t encoded_size, int keyframe);
  int (*flush)(void* self);
  int (*discard_frame)(void* self);
  int (*get_frame)(void* self, uint8_t* decoded, size_t decoded_size);
  int (*decoded_frames_buffered)(void* self);
  int (*wait_until_frames_copied)(void* self);
  const char* (*last_error)(void* self);
  void (*set_wanted)(void* self, const uint64_t* ordinals, size_t n);
} hwang_backend_vtable;
/* decoder_type >= HWANG_DECODER_FIRST_REGISTERED; the table is copied.  Non-zero: refused. */
int hwang_register_decoder_backend(int decoder_type, const hwang_backend_vtable* vtable);

const char* hwang_amd_version(void);

/* -- VideoIndex (hwang/video_index.h:22-77) -------------------------------- */
typedef struct hwang_video_index hwang_video_index;

hwang_video_index* hwang_video_index_create(
    uint32_t timescale, uint64_t duration, uint32_t width, uint32_t height,
    const char* format, const uint64_t* sample_offsets,
    const uint64_t* sample_sizes, size_t num_samples,
    const uint64_t* keyframe_indices, size_t num_keyframes,
    const uint8_t* metadata, size_t metadata_size);
void hwang_video_index_destroy(hwang_video_index* index);
/* VideoIndex::deserialize / serialize (video_index.cpp:24-60): proto3 wire
 * format of hwang/hwang_descriptors.proto:5-15.  deserialize returns NULL on
 * malformed input.  serialize: call with out == NULL to get the size. */
hwang_video_index* hwang_video_index_deserialize(const uint8_t* data,
                                                 size_t size);
size_t hwang_video_index_serialize(const hwang_video_index* index,
                                   uint8_t* out, size_t out_capacity);
uint32_t hwang_video_index_timescale(const hwang_video_index* index);
uint64_t hwang_video_index_duration(const hwang_video_index* index);
double hwang_video_index_fps(const hwang_video_index* index);
uint32_t hwang_video_index_frame_width(const hwang_video_index* index);
uint32_t hwang_video_index_frame_height(const hwang_video_index* index);
const char* hwang_video_index_format(const hwang_video_index* index);
uint64_t hwang_video_index_frames(const hwang_video_index* index);
const uint64_t* hwang_video_index_sample_offsets(const hwang_video_index* index);
const uint64_t* hwang_video_index_sample_sizes(const hwang_video_index* index);
size_t hwang_video_index_num_keyframes(const hwang_video_index* index);
const uint64_t* hwang_video_index_keyframe_indices(const hwang_video_index* index);
size_t hwang_video_index_metadata_size(const hwang_video_index* index);
const uint8_t* hwang_video_index_metadata_bytes(const hwang_video_index* index);
/* "" when the index satisfies the invariants the planner and the decoders rely
 * on (the reference used vector::at and assert, video_index.cpp:70-95):
 * keyframes strictly increasing and < frames, at least one keyframe.  Otherwise
 * a description; such an index can be held and serialized but not decoded from. */
const char* hwang_video_index_problem(const hwang_video_index* index);

/* -- MP4IndexCreator (hwang/mp4_index_creator.h:23-45) --------------------- */
typedef struct hwang_mp4_indexer hwang_mp4_indexer;

hwang_mp4_indexer* hwang_mp4_indexer_create(uint64_t file_size);
void hwang_mp4_indexer_destroy(hwang_mp4_indexer* indexer);
/* MP4IndexCreator::feed: returns 1 while more data is wanted (then
 * *next_offset / *next_size say what to read), 0 when done or failed. */
int hwang_mp4_indexer_feed(hwang_mp4_indexer* indexer, const uint8_t* data,
                           size_t size, uint64_t* next_offset,
                           uint64_t* next_size);
int hwang_mp4_indexer_is_done(hwang_mp4_indexer* indexer);
int hwang_mp4_indexer_is_error(hwang_mp4_indexer* indexer);
const char* hwang_mp4_indexer_error_message(hwang_mp4_indexer* indexer);
/* Caller owns the returned index. */
hwang_video_index* hwang_mp4_indexer_get_video_index(hwang_mp4_indexer* indexer);

/* -- slice_into_video_intervals (hwang/video_index.h:79-85) ---------------- */
typedef struct hwang_video_intervals hwang_video_intervals;

/* Returns NULL when a row is >= frames() (the reference asserts). */
hwang_video_intervals* hwang_slice_into_video_intervals(
    const hwang_video_index* index, const uint64_t* rows, size_t num_rows);
void hwang_video_intervals_destroy(hwang_video_intervals* intervals);
size_t hwang_video_intervals_count(const hwang_video_intervals* intervals);
void hwang_video_intervals_range(const hwang_video_intervals* intervals,
                                 size_t i, uint64_t* start_sample,
                                 uint64_t* end_sample);
size_t hwang_video_intervals_num_valid(const hwang_video_intervals* intervals,
                                       size_t i);
const uint64_t* hwang_video_intervals_valid(const hwang_video_intervals* intervals,
                                            size_t i);

/* -- VideoDecoderInterface + VideoDecoderFactory ---------------------------
 * One C entry point per virtual of hwang/video_decoder_interface.h:35-49;
 * hwang_decoder_create is VideoDecoderFactory::make_from_config
 * (hwang/video_decoder_factory.h:35-37). */
typedef struct hwang_decoder hwang_decoder;

int hwang_decoder_type_supported(int decoder_type);
/* NULL when the type is unsupported or the device cannot be opened.
 * num_devices: what VideoDecoderFactory::make_from_config hands to the backend
 * (video_decoder_factory.cpp:90-91).  The AMD backend reads it as the reference's
 * software backend does (software_video_decoder.cpp:148, libavcodec's
 * thread_count): with n > 1, n threads entropy decode consecutive pictures of
 * this ONE decoder at the same time.  1 - what the reference's Python passes
 * (python/hwang/decoder.py:28) - leaves the count to the backend (the usable CPUs,
 * at most 16; HWANG_ENTROPY_WORKERS overrides, 0 = the thread that calls feed()).
 * 0: the caller brings the parallelism (one decoder per thread of its own, as
 * hwang_retriever_* does): no workers, pictures batched for throughput. */
hwang_decoder* hwang_decoder_create(int device_type, int32_t device_id,
                                    uint32_t num_devices, int decoder_type);
void hwang_decoder_destroy(hwang_decoder* dec);
int hwang_decoder_configure(hwang_decoder* dec, uint32_t width, uint32_t height,
                            const char* format, const uint8_t* extradata,
                            size_t extradata_size);
int hwang_decoder_feed(hwang_decoder* dec, const uint8_t* encoded,
                       size_t encoded_size, int keyframe);
int hwang_decoder_flush(hwang_decoder* dec);
int hwang_decoder_discard_frame(hwang_decoder* dec);
/* `decoded` may be host or device (HBM) memory; RGB24, stride width*3.  Device memory is written by the conversion
 * kernel itself, page-locked host memory (hipHostMalloc / hipHostRegister) by the D2H copy, pageable host memory through
 * a pinned bounce buffer and a memcpy - as the NVIDIA backend's cudaMemcpyDefault serves all three
 * (nvidia_video_decoder.cpp:445-446). */
int hwang_decoder_get_frame(hwang_decoder* dec, uint8_t* decoded,
                            size_t decoded_size);
int hwang_decoder_decoded_frames_buffered(hwang_decoder* dec);
int hwang_decoder_wait_until_frames_copied(hwang_decoder* dec);
const char* hwang_decoder_last_error(hwang_decoder* dec);
/* Extensions (not in the reference interface):
 *  - set_wanted: SparseDecodeHints; ordinals are 0-based output positions
 *    from the interval start, ascending, valid until the next flush.
 *  - set_interval_keyframes (with set_wanted): which samples of the interval, same
 *    numbering, are sync samples (DecoderAutomata::EncodedData::keyframes minus
 *    the interval's first frame).  The planner merges byte-adjacent GOPs into one
 *    interval; with this every GOP of it ends behind its own last wanted frame.
 *  - get_frame_yuv: like get_frame but copies the cropped planar I420 frame
 *    (Y w*h, then U, then V (w/2)*(h/2)); the normative decoder output used
 *    by the bit-exact parity tests.
 *  - set_output_yuv: 1 => get_frame is not expected and RGB conversion is
 *    skipped for this handle. */
int hwang_decoder_set_wanted(hwang_decoder* dec, const uint64_t* ordinals,
                             size_t n);
int hwang_decoder_set_interval_keyframes(hwang_decoder* dec, const uint64_t* ordinals,
                                         size_t n);
int hwang_decoder_get_frame_yuv(hwang_decoder* dec, uint8_t* decoded,
                                size_t decoded_size);

/* -- DecoderAutomata (hwang/decoder_automata.h:38-70) ---------------------- */
typedef struct hwang_automata hwang_automata;

/* DecoderAutomata::EncodedData (decoder_automata.h:43-66) as a plain struct;
 * all pointers are borrowed for the duration of hwang_automata_initialize. */
typedef struct hwang_encoded_data {
  const uint8_t* encoded_video;
  size_t encoded_video_size;
  uint32_t width;
  uint32_t height;
  uint64_t start_keyframe;
  uint64_t end_keyframe;
  const char* format;
  const uint64_t* sample_offsets;
  const uint64_t* sample_sizes;
  size_t num_samples;
  const uint64_t* keyframes;
  size_t num_keyframes;
  const uint64_t* valid_frames;
  size_t num_valid_frames;
} hwang_encoded_data;

/* DecoderAutomata::make_instance; NULL if the backend cannot be made. */
hwang_automata* hwang_automata_create(int device_type, int32_t device_id,
                                      int32_t num_devices, int decoder_type);
void hwang_automata_destroy(hwang_automata* a);
int hwang_automata_initialize(hwang_automata* a, const hwang_encoded_data* data,
                              size_t num_data, const uint8_t* extradata,
                              size_t extradata_size);
/* Fills num_frames * width*height*3 bytes of packed RGB24. */
int hwang_automata_get_frames(hwang_automata* a, uint8_t* buffer,
                              int32_t num_frames);
const char* hwang_automata_last_error(hwang_automata* a);
/* stats[4] = samples_fed, frames_discarded, frames_returned, flushes */
void hwang_automata_stats(hwang_automata* a, int64_t stats[4]);

/* -- Parallel sparse retrieval (the perf path) ------------------------------
 * What python/hwang/decoder.py:30-69 (Decoder.retrieve) does one interval at
 * a time through one DecoderAutomata, done for all intervals at once: the
 * plan's intervals are trimmed to their last wanted frame and fanned out
 * over `num_lanes` independent backend instances (one host entropy thread +
 * one DPB in HBM each) on one GPU or on several.  No data-path collective.
 *
 * Multi-GPU (BASELINE.json north_star: "segments fan out embarrassingly-parallel
 * across the 8 GPUs of one node"): the GOP work items of a call
 * share one queue; lane i decodes on device_ids[i % num_devices]; frames land at
 * the row's position of the request's output.  One process per GPU instead:
 * hwang_retriever_set_shard(rank, world) makes this process decode only its
 * share of every call's items (a deterministic longest-first split every rank
 * computes identically); rows of other shares are not written. */
typedef struct hwang_retriever hwang_retriever;

hwang_retriever* hwang_retriever_create(int device_type, int32_t device_id,
                                        int decoder_type, int32_t num_lanes);
/* num_lanes <= 0: 3 lanes per CPU the process may use (cgroup quota honoured),
 * however many devices; never fewer than one per device. */
hwang_retriever* hwang_retriever_create_multi(int device_type, const int32_t* device_ids,
                                              size_t num_devices, int decoder_type,
                                              int32_t num_lanes);
int hwang_retriever_set_shard(hwang_retriever* r, int32_t rank, int32_t world);
int32_t hwang_retriever_num_lanes(hwang_retriever* r);
/* One clip of a batched call (BASELINE config C3: 128 clips, 1 % of each). */
typedef struct hwang_retrieve_request {
  const hwang_video_index* index;
  const uint8_t* file;      /* whole MP4 in host memory, borrowed for the call */
  size_t file_size;
  const uint64_t* rows;     /* sorted ascending */
  size_t num_rows;
  uint8_t* out;             /* num_rows frames: host memory or HBM of any device of this process */
  size_t out_capacity;
} hwang_retrieve_request;
/* All requests' GOP items are pooled into one queue (python/hwang/decoder.py:30-69
 * is one clip, one interval at a time).  Clips may differ in size and codec
 * configuration; a lane re-configures its backend only when they do. */
int hwang_retriever_retrieve_batch(hwang_retriever* r, const hwang_retrieve_request* requests,
                                   size_t num_requests, int out_is_yuv);
/* per device slot (position in device_ids) after a call:
 * [0] lanes bound to it, [1] work items it decoded, [2] frames returned, [3] samples fed */
void hwang_retriever_device_stats(hwang_retriever* r, int32_t device_slot, int64_t stats[4]);
void hwang_retriever_destroy(hwang_retriever* r);
/* `file` is the whole MP4 mapped or read into memory (borrowed for the
 * call).  rows must be sorted ascending.  `out` receives num_rows frames in
 * row order; out_is_yuv selects cropped I420 (w*h*3/2 per frame) instead of
 * RGB24 (w*h*3).  `out` may be host or device memory. */
int hwang_retriever_retrieve(hwang_retriever* r, const hwang_video_index* index,
                             const uint8_t* file, size_t file_size,
                             const uint64_t* rows, size_t num_rows,
                             uint8_t* out, size_t out_capacity, int out_is_yuv);
const char* hwang_retriever_last_error(hwang_retriever* r);
/* stats of the last retrieve():
 *  [0] intervals, [1] samples fed, [2] pictures reconstructed on the GPU,
 *  [3] pictures skipped (unwanted non-reference), [4] frames returned,
 *  [5] bytes uploaded (macroblock records + coefficients),
 *  [6] host entropy-decode nanoseconds summed over lanes (GPU stalls excluded),
 *  [7] wall nanoseconds, [8] nanoseconds lanes were blocked on the GPU (staging
 *  ring full), summed over lanes, [9] the threads that drove the lanes of the last call */
void hwang_retriever_stats(hwang_retriever* r, int64_t stats[10]);
/* Measurement aid (bench.py's `host_lane_threads` block): what the lane threads of the last retrieve() executed, from
 * the CPU's own counters where the kernel lets a process count its events (HWANG_HOST_COUNTERS=1 in the environment
 * when the call starts; all zeros otherwise): [0] user-mode instructions, [1] cycles, [2] mispredicted branches,
 * [3] CPU nanoseconds of those threads (CLOCK_THREAD_CPUTIME_ID, always filled). */
void hwang_retriever_host_counters(hwang_retriever* r, int64_t out[4]);

/* -- Kernel-level entry points (parity tests + per-kernel roofline) ---------
 * Each runs one HIP kernel of the reconstruct pipeline on device buffers the
 * caller provides (plain device pointers),
 * on the given hipStream_t (NULL = default stream), asynchronously. */
/* K5: cropped planar I420 -> packed RGB24 (replaces sws_scale in
 * impls/software/software_video_decoder.cpp:281-339 and NV12_to_RGB in
 * impls/nvidia/convert.cu:109-219). */
int hwang_k_yuv420_to_rgb24(const uint8_t* y, const uint8_t* u,
                            const uint8_t* v, int y_stride, int c_stride,
                            int width, int height, uint8_t* rgb, void* stream);

/* -- Host-only parser probe (CPU tests; no GPU) -----------------------------
 * Runs the backend's host half (entropy decode, motion derivation, DPB,
 * display-order output, sparse skipping) over samples laid out in `data`.
 * out[0]=slices, [1]=slices that ended exactly on rbsp_trailing_bits,
 * [2]=pictures parsed, [3]=pictures skipped (unwanted non-reference),
 * [4]=frames output, [5]=of which skipped, [6]=int16 coefficients emitted,
 * [7]=FNV-1a digest of every record the kernels would receive. */
int hwang_h264_probe(const uint8_t* avcc, size_t avcc_size, const uint8_t* data,
                     const uint64_t* sample_offsets, const uint64_t* sample_sizes,
                     size_t num_samples, const uint64_t* wanted, size_t num_wanted,
                     int64_t out[8], char* error, size_t error_capacity);

/* The same with the slice data decoded by `workers` entropy threads (0: on the calling thread, as above; the
 * backend's `num_devices` > 1, software_video_decoder.cpp:148's thread count).  out[0..7] as above except that
 * surfaces are numbered round robin (which surface a picture gets may then depend on when a worker let go of one, so
 * out[7] is not comparable between runs); out[8] = the digest with the surface numbers blanked, which is the same
 * for every worker count including 0.  `keyframes` (optional, with `wanted`): ordinals of the sync samples among the
 * samples given - each GOP then ends behind its own last wanted position (SparseDecodeHints::set_interval_keyframes). */
int hwang_h264_probe_workers(const uint8_t* avcc, size_t avcc_size, const uint8_t* data,
                             const uint64_t* sample_offsets, const uint64_t* sample_sizes,
                             size_t num_samples, const uint64_t* wanted, size_t num_wanted,
                             const uint64_t* keyframes, size_t num_keyframes,
                             int workers, int64_t out[9], char* error, size_t error_capacity);

/* Host only: what the hvcC payload of an hev1 / hvc1 track (MP4IndexCreator's metadata for it,
 * mp4_index_creator.cpp:454-469) says about the track - profile, tier, level, picture size, chroma format, bit depth,
 * CTB size, parameter set counts.  The AMD backend's configure() answers an HEVC format name
 * (software_video_decoder.cpp:128-132) with this text inside its "not supported" Result.  Returns 0, or 1 with what
 * is wrong with the record in `out`. */
int hwang_hevc_describe_config(const uint8_t* hvcc, size_t size, char* out, size_t capacity);

/* -- HEVC kernel test hook -----------------------------------------------------
 * One picture's worth of hand-made records (hwang_amd/csrc/amd/hevc_frame_job.h: the seam between the host parser and
 * k_hevc.hip) through the HEVC kernels on the current device: residual -> inter -> intra -> deblocking -> SAO, as the
 * backend launches them for a decoded picture.  The test suite drives it with records built in numpy and compares
 * the result with numpy statements of H.265 8.4 - 8.7 (tests/test_gpu_hevc_kernels.py): the HEVC counterpart of the
 * reference having no kernel of its own to test (its HEVC arithmetic is libavcodec's,
 * impls/software/software_video_decoder.cpp:128-132).  Record arrays are in the layouts of hevc_frame_job.h. */
typedef struct hwang_hevc_test_picture {
  int32_t width, height, log2_ctb;  /* coded size, multiples of 8 */
  int32_t cb_qp_offset, cr_qp_offset, sao_enabled, deblock_enabled;
  int32_t intra_rows;               /* 1: k_hevc_intra_rows (records grouped by coding tree block), 0: k_hevc_intra_flow */
  int32_t reserved;
  const uint8_t* recon_in;          /* I420 (w*h*3/2): what the reconstruction surface holds before the kernels run */
  const uint8_t* refs;              /* num_refs I420 pictures back to back: reference surfaces 0 .. num_refs - 1 */
  int32_t num_refs;
  int32_t num_res, num_coefs, num_pus, num_weights, num_intra[3], num_slices;
  const void* res;                  /* ResTu[num_res] */
  const uint32_t* coefs;            /* level stream */
  const void* pus;                  /* InterPu[num_pus] */
  const void* weights;              /* WeightEntry[num_weights] */
  const void* intra[3];             /* IntraTu[num_intra[c]] per plane, in decoding order */
  const void* edges;                /* EdgeInfo[(w/4)*(h/4)] */
  const void* sao;                  /* SaoCtb[coding tree blocks] */
  const void* slices;               /* SliceLoop[num_slices] */
  const uint8_t* scaling;           /* 4064 bytes or NULL */
  uint8_t* out;                     /* I420 result */
  int16_t* resid_out;               /* optional: the residual planes (w*h + 2*(w/2)*(h/2) int16) */
} hwang_hevc_test_picture;
int hwang_hevc_run_test_picture(const hwang_hevc_test_picture* t, char* error, size_t error_capacity);

/* -- Profiling hooks ---------------------------------------------------------
 * When enabled, every kernel launch of the backend is bracketed with HIP
 * events on its own stream; hwang_prof_read drains the accumulated
 * per-kernel totals.  kernel ids: see hwang_prof_kernel_name(). */
void hwang_prof_enable(int on);
int hwang_prof_num_kernels(void);
const char* hwang_prof_kernel_name(int kernel_id);
/* out[0]=launches, out[1]=total microseconds, out[2]=algorithmic bytes */
int hwang_prof_read(int kernel_id, double out[3]);
void hwang_prof_reset(void);

/* -- Replay: the GPU with the host out of the way ------------------------------
 * Not in the reference; a measurement hook (bench.py's `replay` block).  The path is bound by the host's entropy
 * decoders, so the launches it produces say little about what the kernels can do.  hwang_replay_keep makes the
 * device's executor remember the descriptors of the last `pictures` pictures submitted; after a decode, with every
 * decoder idle but alive, hwang_replay_run launches the reconstruction kernels for `width` of those pictures per
 * launch, `launches` times back to back (chain != 0: as one k_chain launch each instead of the wide kernels).  The
 * pictures are decoded again in place - a measurement, not a decode.  out[0] = distinct pictures available, out[1] =
 * pictures per launch used, out[2] = launches, out[3] = seconds (HIP events around all of them); the per-kernel
 * spans come from hwang_prof_*. */
int hwang_replay_keep(int device, int pictures);
int hwang_replay_run(int device, int width, int launches, int chain, double out[4], char* error, size_t error_capacity);

/* -- Orderly shutdown ----------------------------------------------------------
 * Stops the per-device executor threads after draining their streams.  The
 * library also does this from atexit; call it earlier (the ctypes package does,
 * from Python's atexit) when something else - a profiler, an embedding
 * runtime - tears the HIP runtime down before C atexit handlers run.  All
 * decoders and retrievers must have been destroyed or be idle; afterwards new
 * ones cannot be created in this process. */
void hwang_shutdown(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* HWANG_AMD_H_ */
