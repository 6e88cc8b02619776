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
 * pins it is listed in h264_tables.h and DESIGN.md.
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

/* counters[0]=slices decoded, [1]=slices whose data ended exactly at
 * rbsp_trailing_bits (the desync check), [2]=macroblocks, [3]=pictures,
 * [4]=I/[5]=P/[6]=B slices, [7]=intra MBs, [8]=skip MBs, [9]=8x8-transform MBs */
void h264o_counters(H264Oracle* d, int64_t out[10]);

/* I420 -> RGB24 exactly as hwang_amd/csrc/amd/k_rgb.hip documents it. */
void h264o_yuv420_to_rgb24(const uint8_t* i420, int width, int height,
                           uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
