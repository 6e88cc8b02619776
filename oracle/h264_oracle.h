/* TEST INFRASTRUCTURE -- scalar CPU restatement of the H.264 decode that the
 * reference obtains from FFmpeg (impls/software/software_video_decoder.cpp:
 * 349-457 -> avcodec_send_packet / avcodec_receive_frame, and :281-339 ->
 * sws_scale).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything in oracle/.  The product library never
 * links or calls this code.
 *
 * PARITY UNPINNED (pixel path): FFmpeg n3.3.1 is not in /root/reference nor
 * in this image, and the reference's tests hold no golden pixels
 * (decoder_automata_test.cpp:262-342 only checks sparse == sequential).  The
 * decode follows ITU-T H.264 clauses 7.3/7.4, 8.2-8.7, 9.1-9.3 restated from
 * memory; H.264 decoding is normative/bit-exact, so any conforming decoder
 * (FFmpeg included) yields these samples if the restatement is right.  What
 * pins it is listed in h264_tables.h and DESIGN.md; since round 3 the
 * reconstruction arithmetic (interpolation, weighted prediction, in-loop
 * filter, transforms, intra and motion-vector prediction) is also checked
 * against numpy / Python restatements written from the clauses
 * (tests/test_*_independent.py).  The entropy tables and the slice-data syntax
 * are checked by two foreign encoders' streams ending every slice where they
 * must: realshort.mp4 (decoded) and, since round 4, cockatoo.mp4 - B slices,
 * weighted prediction, 8x8 transform - in the parse-only mode below
 * (tests/test_oracle_realstream.py).  No decoded SAMPLE has ever been compared
 * with FFmpeg's: that is what "unpinned" means here.
 */
#ifndef ORACLE_H264_ORACLE_H_
#define ORACLE_H264_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct H264Oracle H264Oracle;

H264Oracle* h264o_open(void);
void h264o_close(H264Oracle* d);
/* avcC payload (what VideoDecoderInterface::configure receives). */
int h264o_configure(H264Oracle* d, const uint8_t* avcc, size_t size);
/* One MP4 sample = one access unit in AVCC framing.  0 on success. */
int h264o_decode(H264Oracle* d, const uint8_t* sample, size_t size);
/* End of interval: push every buffered picture out in display order and
 * forget all references (next sample must be an IDR). */
int h264o_flush(H264Oracle* d);
int h264o_frames_ready(H264Oracle* d);
/* Cropped size + POC of the oldest ready frame. */
int h264o_frame_info(H264Oracle* d, int* width, int* height, int* poc);
/* Copies the oldest ready frame as cropped planar I420 (Y w*h, U, V
 * ceil(w/2)*ceil(h/2)) and drops it.  dst may be NULL to just drop. */
int h264o_pop_frame(H264Oracle* d, uint8_t* dst, size_t capacity);
const char* h264o_error(H264Oracle* d);
/* Parse-only mode (before h264o_configure): every syntax element is entropy decoded, reference lists / marking / POC
 * / output order run, counters[] count - but nothing is reconstructed (the frames that come out are blank).  In this
 * mode the oracle also accepts High 4:4:4 Predictive streams (chroma_format_idc 3 without separate colour planes,
 * 8 bit, no transform bypass; CABAC), which it cannot reconstruct: their slice data is parsed with Cb and Cr coded
 * like luma (ctxBlockCat 6 - 13, 7.3.5.3 / 9.3.3.1.3).  Purpose: the desync check - "every slice ends on its
 * rbsp_stop_one_bit after the right number of macroblocks" - on a second real encoder's stream (imageio's
 * cockatoo.mp4: B slices, explicit and implicit weighted prediction, 8x8 transform, four references), which pins
 * the B-slice syntax, the weight tables and the entropy contexts it touches without a second decoder
 * (tests/test_oracle_realstream.py). */
void h264o_set_parse_only(H264Oracle* d, int on);

/* counters[0]=slices decoded, [1]=slices whose data ended exactly at
 * rbsp_trailing_bits (the desync check), [2]=macroblocks, [3]=pictures,
 * [4]=I/[5]=P/[6]=B slices, [7]=intra MBs, [8]=skip MBs, [9]=8x8-transform MBs */
void h264o_counters(H264Oracle* d, int64_t out[10]);

/* Coverage histogram of the syntax and arithmetic cases a stream exercised
 * (tests/test_corpus_coverage.py asserts that the corpus reaches every
 * reachable cell).  Off until enabled; counts are per decoder instance.
 * Section layout (cell index = section base + the index given):
 *   H264O_COV_I4        mode(0..8) * 16 + (left | top << 1 | topleft << 2 | topright << 3)
 *   H264O_COV_I8        same for Intra 8x8
 *   H264O_COV_I16       mode(0..3) * 4 + (left | top << 1)
 *   H264O_COV_IC        chroma mode(0..3) * 4 + (left | top << 1)
 *   H264O_COV_LUMA_MC   (yFrac * 4 + xFrac) * 3 + block class (0: 16x16, 1: 8x8, 2: 4x4 regions of equal motion)
 *   H264O_COV_CHROMA_MC yFrac(0..7) * 8 + xFrac
 *   H264O_COV_WP        weight mode(0 default, 1 explicit, 2 implicit) * 3 + (0 list0 only, 1 list1 only, 2 both)
 *   H264O_COV_BS        ((dir * 2 + macroblock_edge) * 3 + plane) * 5 + bS       (per 4-sample luma / 2-sample chroma run)
 *   H264O_COV_FILT      0-3 luma bS<4: (ap<beta) * 2 + (aq<beta); 4-7 luma bS=4: strong_p * 2 + strong_q;
 *                       8 chroma bS<4; 9 chroma bS=4; 10 luma / 11 chroma sample left alone by the alpha / beta tests
 *   H264O_COV_MBTYPE_I  mb_type 0..25
 *   H264O_COV_MBTYPE_P  mb_type 0..30, 31 = P_Skip
 *   H264O_COV_MBTYPE_B  mb_type 0..48, 49 = B_Skip
 *   H264O_COV_SUB_P     sub_mb_type 0..3        H264O_COV_SUB_B  sub_mb_type 0..12
 *   H264O_COV_COEFF_TOKEN  table(0..3 by nC, 4 chroma DC) * 68 + total_coeff * 4 + trailing_ones
 *   H264O_COV_TOTAL_ZEROS  (total_coeff - 1) * 16 + total_zeros;  _DC: (total_coeff - 1) * 4 + total_zeros
 *   H264O_COV_RUN_BEFORE   (min(zeros_left, 7) - 1) * 15 + run_before
 *   H264O_COV_ESCAPE    0 CAVLC level_prefix 14 (suffixLength 0), 1 level_prefix 15, 2 level_prefix >= 16,
 *                       3 CABAC coeff_abs_level suffix (>= 15), 4 CABAC mvd suffix (>= 9)
 *   H264O_COV_DEQUANT   0/1 4x4 qP < 24 / >= 24, 2/3 8x8 qP < 36 / >= 36, 4/5 Intra16x16 DC qP < 36 / >= 36, 6 chroma DC
 *   H264O_COV_DIRECT    (temporal * 2 + sub_macroblock_only) * 2 + direct_8x8_inference
 *   H264O_COV_CTX       ((table * 460) + ctxIdx) * 2 + bin value; table: 0 I, 1..3 P with cabac_init_idc 0..2,
 *                       4..6 B with cabac_init_idc 0..2 */
enum {
  H264O_COV_I4 = 0,
  H264O_COV_I8 = H264O_COV_I4 + 144,
  H264O_COV_I16 = H264O_COV_I8 + 144,
  H264O_COV_IC = H264O_COV_I16 + 16,
  H264O_COV_LUMA_MC = H264O_COV_IC + 16,
  H264O_COV_CHROMA_MC = H264O_COV_LUMA_MC + 48,
  H264O_COV_WP = H264O_COV_CHROMA_MC + 64,
  H264O_COV_BS = H264O_COV_WP + 9,
  H264O_COV_FILT = H264O_COV_BS + 60,
  H264O_COV_MBTYPE_I = H264O_COV_FILT + 12,
  H264O_COV_MBTYPE_P = H264O_COV_MBTYPE_I + 26,
  H264O_COV_MBTYPE_B = H264O_COV_MBTYPE_P + 32,
  H264O_COV_SUB_P = H264O_COV_MBTYPE_B + 50,
  H264O_COV_SUB_B = H264O_COV_SUB_P + 4,
  H264O_COV_COEFF_TOKEN = H264O_COV_SUB_B + 13,
  H264O_COV_TOTAL_ZEROS = H264O_COV_COEFF_TOKEN + 340,
  H264O_COV_TOTAL_ZEROS_DC = H264O_COV_TOTAL_ZEROS + 240,
  H264O_COV_RUN_BEFORE = H264O_COV_TOTAL_ZEROS_DC + 12,
  H264O_COV_ESCAPE = H264O_COV_RUN_BEFORE + 105,
  H264O_COV_DEQUANT = H264O_COV_ESCAPE + 5,
  H264O_COV_DIRECT = H264O_COV_DEQUANT + 7,
  H264O_COV_CTX = H264O_COV_DIRECT + 8,
  H264O_COV_CELLS = H264O_COV_CTX + 7 * 460 * 2
};
void h264o_coverage_enable(H264Oracle* d);
/* copies min(n, H264O_COV_CELLS) cells; returns H264O_COV_CELLS */
int h264o_coverage(H264Oracle* d, int64_t* out, int n);

/* I420 -> RGB24 exactly as hwang_amd/csrc/amd/k_rgb.hip documents it. */
void h264o_yuv420_to_rgb24(const uint8_t* i420, int width, int height,
                           uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
