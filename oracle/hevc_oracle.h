/* TEST INFRASTRUCTURE -- scalar CPU restatement of the HEVC (ITU-T H.265, Main profile: 8 bit, 4:2:0) decode that the
 * reference obtains from FFmpeg when a track's format is "h265" / "hev1" / "hevc"
 * (impls/software/software_video_decoder.cpp:128-132 selects AV_CODEC_ID_HEVC; :349-457 -> avcodec_send_packet /
 * avcodec_receive_frame; the indexer hands the hvcC payload over as extradata, mp4_index_creator.cpp:454-469).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.  The product library
 * never links or calls this code.
 *
 * PARITY UNPINNED: FFmpeg is not in /root/reference nor in this image, no HEVC stream from any other encoder exists
 * on either machine (the reference's test clip is fetched with wget, tests/videos.h:26-43), and the reference's tests
 * hold no golden pixels.  The decode follows H.265 (04/2013) clauses 7.3 / 7.4 (syntax, semantics), 8.3 (POC,
 * reference picture sets, lists), 8.4 (intra), 8.5 (merge / AMVP / temporal candidates, fractional interpolation,
 * weighted prediction), 8.6 (QP derivation, scaling, transforms), 8.7 (deblocking, SAO), 9.3 (CABAC) and C.5.2
 * (output order), restated from memory.  What stands against a mirrored misunderstanding:
 *   - tests/test_hevc_independent.py: numpy statements of the transforms, interpolation, intra prediction, deblocking
 *     and SAO written from the clauses and compared with this code on known-answer pictures;
 *   - tests/test_hevc_tables.py: structural checks of the constant tables;
 *   - every corpus stream carries the decoded picture hash SEI (MD5, D.3.19); hevco_decode() checks it, so the first
 *     foreign stream that carries one pins everything at once (hevco_counters: md5 checked / mismatched).
 * Not supported (refused by name): bit depths above 8, chroma formats other than 4:2:0, tiles, dependent slice
 * segments.  Supported beyond what the corpus encoder emits: wavefront entry points (entropy_coding_sync), long-term
 * reference pictures, inter-predicted reference picture sets, list modification.
 */
#ifndef ORACLE_HEVC_ORACLE_H_
#define ORACLE_HEVC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HevcOracle HevcOracle;

HevcOracle* hevco_open(void);
void hevco_close(HevcOracle* d);
/* hvcC payload (what VideoDecoderInterface::configure receives for an HEVC track). */
int hevco_configure(HevcOracle* d, const uint8_t* hvcc, size_t size);
/* One MP4 sample = one access unit, NAL units length-prefixed.  0 on success. */
int hevco_decode(HevcOracle* d, const uint8_t* sample, size_t size);
/* End of interval: every buffered picture out in display order, all references forgotten. */
int hevco_flush(HevcOracle* d);
int hevco_frames_ready(HevcOracle* d);
int hevco_frame_info(HevcOracle* d, int* width, int* height, int* poc);
/* Copies the oldest ready frame as cropped planar I420 and drops it.  dst may be NULL to just drop. */
int hevco_pop_frame(HevcOracle* d, uint8_t* dst, size_t capacity);
const char* hevco_error(HevcOracle* d);
/* counters[0] slices, [1] slices whose data ended exactly at the stop bit, [2] coding units, [3] pictures,
 * [4] I / [5] P / [6] B slices, [7] intra CUs, [8] skipped CUs, [9] pictures whose MD5 SEI was checked,
 * [10] ... and did not match, [11] transform units */
void hevco_counters(HevcOracle* d, int64_t out[12]);

/* Stand-alone pieces for the independent tests (tests/test_hevc_independent.py). */
/* residual of one transform block: levels[n*n] raster -> res[n*n]; flat scaling lists */
void hevco_residual_block(const int16_t* levels, int log2_size, int c_idx, int qp, int intra, int transform_skip,
                          int16_t* res);
/* intra prediction of one block from explicit neighbour arrays: left[2n+1] (index 0 = top-left corner, then downwards),
 * top[2n] (left to right), after substitution; filtering as 8.4.4.2.3 decides */
void hevco_intra_block(const uint8_t* corner_and_left, const uint8_t* top, int log2_size, int c_idx, int mode,
                       int strong_smoothing_enabled, uint8_t* pred);
/* fractional interpolation + default weighting of one block from a reference plane (w x h, stride w) */
void hevco_inter_block(const uint8_t* ref, int ref_w, int ref_h, int c_idx, int x, int y, int w, int h, int mvx, int mvy,
                       uint8_t* pred);

/* in-loop filter arithmetic on explicit samples: a 4-line luma edge segment (8 samples across per line), one chroma
 * edge row (p1 p0 q0 q1, strength 2), SAO of a block treated as a whole picture */
void hevco_deblock_luma(uint8_t* blk, int bs, int qp_p, int qp_q, int beta_offset_div2, int tc_offset_div2, int no_p, int no_q);
void hevco_deblock_chroma(uint8_t* row, int qp_p, int qp_q, int c_qp_pic_offset, int tc_offset_div2);
void hevco_sao_block(const uint8_t* in, int w, int h, int type, int band_pos, int eo_class, const int8_t* offsets, uint8_t* out);
/* a constant table of hevc_tables.h by name ("trans_matrix", "cabac_init", "scan20" ...): pointer and size in bytes */
const void* hevco_table(const char* name, int* bytes);
void hevco_set_parse_only(HevcOracle* d, int on);

#ifdef __cplusplus
}
#endif
#endif
