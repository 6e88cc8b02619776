"""Helpers around the synthetic-corpus generator (corpus/h264_encoder.c)."""
import os
import subprocess

import numpy as np

from tests.util import ROOT

ENCODER = os.path.join(ROOT, "corpus", "_build", "h264_encoder")


def make_clip(path, width=320, height=240, yuv=True, **kw):
    """Encodes a synthetic clip; returns the encoder's own reconstruction
    (display order, cropped I420, one row per frame) when yuv=True."""
    if not os.path.exists(ENCODER):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "corpus")])
    args = [ENCODER, "--width", str(width), "--height", str(height), "--out", path]
    ypath = path + ".yuv"
    if yuv:
        args += ["--yuv-out", ypath]
    for k, v in kw.items():
        args += ["--" + k.replace("_", "-"), str(v)]
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("corpus encoder failed: " + r.stderr[-400:])
    if not yuv:
        return None
    fs = width * height * 3 // 2
    gold = np.fromfile(ypath, dtype=np.uint8).reshape(-1, fs)
    os.unlink(ypath)
    return gold


# (name, kwargs): small clips that together touch every syntax path the
# generator can emit
MATRIX = [
    ("ip_cabac_t8", dict(frames=10, gop=10, bframes=0, cabac=1, t8x8=1, refs=2)),
    ("ibp_cabac", dict(frames=13, gop=13, bframes=2, cabac=1, t8x8=0, refs=2)),
    ("ibp_cabac_t8", dict(frames=13, gop=13, bframes=2, cabac=1, t8x8=1, refs=2)),
    ("ip_cavlc_baseline", dict(frames=8, gop=8, bframes=0, cabac=0, t8x8=0, refs=1)),
    ("ibp_cavlc_t8", dict(frames=13, gop=7, bframes=2, cabac=0, t8x8=1, refs=3)),
    ("slices3", dict(frames=10, gop=5, bframes=1, cabac=1, t8x8=1, refs=2, slices=3)),
    ("temporal_implicit", dict(frames=10, gop=10, bframes=2, cabac=1, t8x8=1, temporal_direct=1, implicit_wp=1)),
    ("deblock_idc2", dict(frames=7, gop=7, bframes=1, cabac=1, deblock=2, slices=2, alpha=2, beta=-1,
                          chroma_qp_offset=-2)),
    ("qp40", dict(frames=7, gop=7, bframes=2, qp=40)),
    ("qp12", dict(frames=5, gop=5, bframes=2, qp=12)),
    ("pcm", dict(frames=4, gop=4, bframes=1, pcm_period=37)),
    # round 2: syntax x264 / other encoders use by default (VERDICT r1 missing #6)
    ("wp_explicit_d5", dict(frames=13, gop=13, bframes=2, refs=3, explicit_wp=1, wp_denom=5, slices=2)),
    ("wp_explicit_d0", dict(frames=9, gop=9, bframes=1, refs=2, explicit_wp=1, wp_denom=0)),
    ("wp_explicit_d7_p", dict(frames=8, gop=8, bframes=0, refs=2, explicit_wp=1, wp_denom=7, cabac=0)),
    ("rplm_b", dict(frames=16, gop=16, bframes=2, refs=4, rplm=1, slices=2)),
    ("rplm_p_cavlc", dict(frames=12, gop=12, bframes=0, refs=4, rplm=1, cabac=0, t8x8=0)),
    ("mmco_p", dict(frames=28, gop=14, bframes=0, refs=4, mmco=1)),
    ("mmco_b_rplm", dict(frames=32, gop=16, bframes=2, refs=4, mmco=1, rplm=1)),
    ("mmco5_rplm", dict(frames=30, gop=15, bframes=0, refs=3, mmco=2, rplm=1)),
    ("mmco_temporal", dict(frames=20, gop=10, bframes=1, refs=3, mmco=1, temporal_direct=1, implicit_wp=1)),
    ("scaling_sps", dict(frames=7, gop=7, bframes=2, scaling=1)),
    ("scaling_pps", dict(frames=7, gop=7, bframes=2, scaling=2)),
    ("scaling_both_4x4only", dict(frames=7, gop=7, bframes=1, scaling=3, t8x8=0, cabac=0)),
    ("everything", dict(frames=26, gop=13, bframes=2, refs=4, scaling=3, explicit_wp=1, wp_denom=4, rplm=1, mmco=1,
                        slices=3, deblock=2, alpha=-1, beta=1, chroma_qp_offset=3)),
    # rate control: the QP walks per macroblock row towards a bit budget (long monotonic runs of mb_qp_delta, every
    # deblocked edge between different QPs), starved and generous
    ("rate_cabac_low", dict(frames=13, gop=13, bframes=2, cabac=1, t8x8=1, refs=2, rate_kbit=12)),
    ("rate_cavlc_high", dict(frames=10, gop=10, bframes=1, cabac=0, t8x8=1, refs=2, slices=3, rate_kbit=90,
                             chroma_qp_offset=-3)),
    # picture order count types 1 (frame_num cycle + per-slice delta, with and without B pictures) and 2, intra
    # prediction that may not lean on inter neighbours, direct prediction at 4x4 granularity (spatial and temporal)
    ("poc1_b", dict(frames=20, gop=10, bframes=2, refs=3, poc_type=1, slices=2)),
    ("poc1_p_cavlc", dict(frames=12, gop=6, bframes=0, refs=2, poc_type=1, cabac=0)),
    ("poc2_p", dict(frames=12, gop=6, bframes=0, refs=3, poc_type=2, mmco=1)),
    ("constrained_intra", dict(frames=13, gop=13, bframes=2, refs=2, constrained_intra=1, pcm_period=53)),
    ("direct4x4_spatial", dict(frames=13, gop=13, bframes=2, refs=2, direct8x8=0)),
    ("direct4x4_temporal", dict(frames=13, gop=13, bframes=2, refs=2, direct8x8=0, temporal_direct=1, implicit_wp=1, cabac=0)),
    # B pictures as references (x264's default b-pyramid): reordering depth 2, a B picture as the co-located one of
    # temporal direct, sparse retrieval that may skip only the non-reference ones; slice_qp_delta != 0
    ("bpyramid3", dict(frames=26, gop=13, bframes=3, refs=4, bpyramid=1, slices=2)),
    ("bpyramid2_temporal", dict(frames=17, gop=17, bframes=2, refs=4, bpyramid=1, temporal_direct=1, implicit_wp=1)),
    ("bpyramid3_cavlc_wp", dict(frames=17, gop=17, bframes=3, refs=4, bpyramid=1, cabac=0, explicit_wp=1, wp_denom=5,
                                slice_qp_delta=5)),
    ("slice_qp_delta", dict(frames=9, gop=9, bframes=1, refs=2, slice_qp_delta=-7, slices=3)),
    # frame_num and pic_order_cnt_lsb four or five bits wide: both wrap every few pictures (FrameNumWrap in list
    # initialisation, sliding window and the marking operations; PicOrderCntMsb), GOPs longer than either period
    ("wrap_b", dict(frames=60, gop=30, bframes=1, refs=3, fn_bits=4, poc_bits=4)),
    ("wrap_mmco_rplm", dict(frames=60, gop=30, bframes=2, refs=4, fn_bits=4, poc_bits=5, mmco=1, rplm=1)),
    ("wrap_mmco5_p", dict(frames=60, gop=30, bframes=0, refs=4, fn_bits=4, poc_bits=5, mmco=2, rplm=1, cabac=0)),
    ("wrap_poc1", dict(frames=48, gop=24, bframes=2, refs=3, fn_bits=5, poc_type=1)),
    ("wrap_pyramid", dict(frames=52, gop=26, bframes=3, refs=4, fn_bits=4, poc_bits=5, bpyramid=1, temporal_direct=1, implicit_wp=1)),
    # what muxers and other encoders put into samples: access unit delimiters, parameter sets in band and an SEI at
    # every IDR, filler data behind the slices; two-byte NAL lengths in avcC
    ("extra_nals", dict(frames=24, gop=8, bframes=2, refs=2, slices=2, extra_nals=1)),
    ("nal_len2_extra_cavlc", dict(frames=16, gop=8, bframes=1, refs=2, cabac=0, extra_nals=1, nal_length_size=2)),
    # two picture parameter sets (other pic_init_qp, other chroma QP offsets), alternating picture by picture
    ("multi_pps", dict(frames=17, gop=17, bframes=2, refs=2, multi_pps=1)),
    ("multi_pps_cavlc_inband", dict(frames=16, gop=8, bframes=1, refs=2, cabac=0, multi_pps=1, extra_nals=1,
                                    chroma_qp_offset=-3, slice_qp_delta=2, slices=2)),
    # ... and the second set switches the entropy coder: CABAC and CAVLC pictures alternate in one stream
    ("mixed_entropy", dict(frames=26, gop=13, bframes=2, refs=3, multi_pps=2, slices=2)),
    ("mixed_entropy_p", dict(frames=16, gop=8, bframes=0, refs=2, cabac=0, multi_pps=2, t8x8=1)),
    # long reference lists: up to sixteen frames in the DPB (ref_idx beyond the first few contexts / te() ranges,
    # sixteen decode surfaces per lane)
    ("refs16_b_rplm", dict(frames=40, gop=40, bframes=2, refs=16, rplm=1)),
    ("refs12_mmco_cavlc", dict(frames=30, gop=30, bframes=1, refs=12, mmco=1, cabac=0)),
    ("refs8_p", dict(frames=24, gop=24, bframes=0, refs=8, explicit_wp=1, wp_denom=3)),
    # a cropping rectangle that starts inside the coded picture (left / top offsets, not only right / bottom padding)
    ("crop_left_top", dict(width=322, height=242, frames=9, gop=9, bframes=2, crop_left=3, crop_top=2)),
    ("crop_left_top_odd", dict(width=306, height=226, frames=7, gop=7, bframes=1, crop_left=5, crop_top=7, cabac=0)),
    # the loop filter switched off (disable_deblocking_filter_idc = 1): no k_bs / k_deblock work at all
    ("deblock_off", dict(frames=10, gop=10, bframes=2, refs=2, deblock=1, slices=2)),
    # slices of different types inside one picture: I slices in P pictures, P and I slices in B pictures
    ("mixed_slices_b", dict(frames=13, gop=13, bframes=2, refs=3, slices=3, mixed_slices=1)),
    ("mixed_slices_p_cavlc_wp", dict(frames=12, gop=12, bframes=0, refs=2, slices=4, mixed_slices=1, cabac=0, explicit_wp=1,
                                     wp_denom=4)),
    ("mixed_slices_temporal", dict(frames=13, gop=13, bframes=2, refs=3, slices=3, mixed_slices=1, temporal_direct=1,
                                   implicit_wp=1, deblock=2)),
    # slice boundaries in the middle of macroblock rows (a new slice every N macroblocks), many small slices
    ("slice_mbs37", dict(frames=10, gop=10, bframes=2, refs=3, slice_mbs=37)),
    ("slice_mbs7_cavlc_idc2", dict(frames=8, gop=8, bframes=1, refs=2, slice_mbs=7, cabac=0, deblock=2, alpha=1, beta=-2)),
    ("slice_mbs53_mixed", dict(frames=10, gop=10, bframes=2, refs=3, slice_mbs=53, mixed_slices=1, temporal_direct=1,
                               implicit_wp=1)),
    # the QP jumping anywhere in 0..51 every few macroblocks: mb_qp_delta up to the wrap-around, QP 0 levels that need
    # the escape codes, chroma QP table ends
    ("qp_swing_cabac", dict(frames=10, gop=10, bframes=2, refs=2, qp_swing=1)),
    ("qp_swing_cavlc", dict(frames=8, gop=8, bframes=1, refs=2, qp_swing=1, cabac=0, chroma_qp_offset=5, scaling=3)),
    # vectors hundreds of samples long, mostly pointing outside the reference picture (edge clamping only; long mvd codes)
    ("wild_mv_p", dict(frames=10, gop=10, bframes=0, refs=3, wild_mv=1)),
    ("wild_mv_b_cavlc", dict(frames=10, gop=10, bframes=2, refs=2, wild_mv=1, cabac=0, explicit_wp=1, wp_denom=6)),
    # every picture an IDR picture (consecutive IDRs, idr_pic_id alternating), without a sync sample box
    ("all_intra", dict(frames=6, gop=1, bframes=0, no_stss=1, qp_swing=1)),
    ("all_intra_cavlc", dict(frames=5, gop=1, bframes=0, cabac=0, t8x8=0, slices=2, scaling=1)),
    # intra families and modes drawn by hash among the valid ones (not by best match): every mode of every family,
    # every availability pattern - also with slices and with constrained intra prediction taking neighbours away
    ("random_intra", dict(frames=8, gop=2, bframes=0, random_intra=1, qp=22)),
    ("random_intra_cavlc_slices", dict(frames=9, gop=3, bframes=1, random_intra=1, cabac=0, slice_mbs=31, constrained_intra=1)),
    ("random_intra_t4_only", dict(frames=6, gop=3, bframes=0, random_intra=1, t8x8=0, slices=3, qp=34)),
    # round 3: --cover makes the encoder take by hash what its mode decision never picks (intra macroblocks of every family
    # and type in P and B slices, Intra16x16 with chroma but no luma AC or without a DC block, B 16x8 / 8x16 halves from
    # different lists, explicit quadrants with reference 0 beside neighbours above 0, levels that need level_prefix >= 16):
    # with these eight the corpus reaches every cell of the coverage histogram that tests/coverage_model.py calls reachable
    ("cover_cabac", dict(frames=26, gop=13, bframes=2, refs=3, cover=1, random_intra=1, constrained_intra=1, qp_swing=1, slices=2)),
    ("cover_cavlc", dict(frames=26, gop=13, bframes=2, refs=3, cover=1, random_intra=1, constrained_intra=1, qp_swing=1, cabac=0,
                         slice_mbs=41)),
    ("cover_cabac_t4", dict(frames=20, gop=10, bframes=1, refs=2, cover=1, random_intra=1, t8x8=0, qp=20)),
    ("cover_cabac_hiqp", dict(frames=39, gop=13, bframes=2, refs=3, cover=1, random_intra=1, constrained_intra=1, qp=45,
                              pcm_period=211, slice_mbs=67)),
    ("cover_cavlc_midqp", dict(frames=26, gop=13, bframes=2, refs=2, cover=1, random_intra=1, qp=16, cabac=0, seed=5)),
    ("cover_cabac_pyramid", dict(frames=39, gop=13, bframes=3, refs=4, cover=1, random_intra=1, bpyramid=1, temporal_direct=1,
                                 implicit_wp=1, qp=36, seed=9)),
    ("cover_cabac_hiqp2", dict(frames=52, gop=13, bframes=2, refs=4, cover=1, random_intra=1, constrained_intra=1, qp=40, slices=3,
                               seed=9, rplm=1)),
    ("cover_cabac_pyramid2", dict(frames=39, gop=13, bframes=3, refs=4, cover=1, bpyramid=1, temporal_direct=1, implicit_wp=1, qp=30,
                                  seed=4, slices=2)),
]


# The row kernels cut a picture into bands of four macroblock rows, one
# workgroup each: cover every remainder (1, 2, 3 rows in the last band, a single
# band, a single row, a single column), pictures wider than the 256-macroblock
# LDS line buffers (surface hand-off, host-side vectors) and slices that end
# inside a band.
BAND_SIZES = [
    ("1x1mb", 16, 16, dict(frames=6, gop=6, bframes=1)),
    ("1col_5rows", 16, 80, dict(frames=6, gop=6, bframes=1)),
    ("3x5mb", 48, 80, dict(frames=7, gop=7, bframes=2)),
    ("4x6mb", 64, 96, dict(frames=7, gop=7, bframes=2, slices=2)),
    ("qcif_9rows", 176, 144, dict(frames=9, gop=9, bframes=2, slices=3)),
    ("wide_129x3mb", 2064, 48, dict(frames=5, gop=5, bframes=1)),
    ("wide_130x7mb_slices", 2080, 112, dict(frames=5, gop=5, bframes=0, slices=4)),
    ("wide_257x3mb", 4112, 48, dict(frames=5, gop=5, bframes=1)),
    ("wide_258x6mb_slices", 4128, 96, dict(frames=4, gop=4, bframes=0, slices=3)),
    # the same edges with the syntax variety added in round 2: mixed slice types cut mid-row, reference B pictures,
    # two parameter sets, wild vectors, QP swings
    ("wide_257x5mb_mixed", 4112, 80, dict(frames=7, gop=7, bframes=2, refs=4, bpyramid=0, mixed_slices=1, slice_mbs=301,
                                            multi_pps=2, wild_mv=1, qp_swing=1)),
    ("wide_131x9mb_pyramid", 2096, 144, dict(frames=9, gop=9, bframes=3, refs=4, bpyramid=1, slice_mbs=97, mixed_slices=1,
                                              temporal_direct=1, implicit_wp=1, rate_kbit=60)),
    # the largest sides the decoder takes: 512 macroblocks (8192 samples) wide or high
    ("wide_512x1mb", 8192, 16, dict(frames=4, gop=4, bframes=1)),
    ("tall_1x512mb", 16, 8192, dict(frames=4, gop=4, bframes=1, slices=5)),
    ("wide_511x2mb", 8176, 32, dict(frames=4, gop=4, bframes=1, refs=2, mixed_slices=1, slice_mbs=200)),
    ("7x7mb_everything", 112, 112, dict(frames=13, gop=13, bframes=3, refs=4, bpyramid=1, slice_mbs=5, mixed_slices=1,
                                          multi_pps=2, wild_mv=1, qp_swing=1, fn_bits=4, poc_bits=5, extra_nals=1)),
]


def all_cases():
    """(name, width, height, encoder kwargs) of every MATRIX and BAND_SIZES case, as the parity tests encode them."""
    for name, kw in MATRIX:
        kw = dict(kw)
        w, h = kw.pop("width", 322 if name == "qp40" else 320), kw.pop("height", 242 if name == "qp40" else 240)
        yield name, w, h, kw
    for name, w, h, kw in BAND_SIZES:
        yield name, w, h, dict(kw)


def plane_md5s(i420, w, h):
    """MD5 of the Y, U and V plane of one cropped I420 frame (the form tests/golden/corpus_md5.json holds)."""
    import hashlib
    cw, ch = (w + 1) // 2, (h + 1) // 2
    b = np.ascontiguousarray(i420).tobytes()
    y, u, v = b[:w * h], b[w * h:w * h + cw * ch], b[w * h + cw * ch:w * h + 2 * cw * ch]
    return [hashlib.md5(p).hexdigest() for p in (y, u, v)]


_GOLDEN_MD5 = None


def golden_md5(name):
    """The committed per-frame plane digests of a corpus case (tests/golden/corpus_md5.json, written by
    tests/golden/make_corpus_golden.py from the CPU oracle's output): {"mp4_md5", "frames": [[y, u, v], ...]}."""
    global _GOLDEN_MD5
    if _GOLDEN_MD5 is None:
        import json
        with open(os.path.join(ROOT, "tests", "golden", "corpus_md5.json")) as f:
            _GOLDEN_MD5 = json.load(f)
    return _GOLDEN_MD5["cases"][name]


# ---------------------------------------------------------------------------------------------------------- HEVC
HEVC_ENCODER = os.path.join(ROOT, "corpus", "_build", "hevc_encoder")


def make_hevc_clip(path, width=128, height=96, yuv=True, **kw):
    """corpus/hevc_encoder.c: a seeded HEVC Main clip in MP4; returns the encoder's own reconstruction (display order,
    cropped I420, one row per frame) when yuv=True."""
    if not os.path.exists(HEVC_ENCODER):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "corpus")])
    args = [HEVC_ENCODER, "--width", str(width), "--height", str(height), "--out", path]
    ypath = path + ".yuv"
    if yuv:
        args += ["--yuv-out", ypath]
    for k, v in kw.items():
        args += ["--" + k.replace("_", "-"), str(v)]
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("HEVC corpus encoder failed: " + r.stderr[-400:])
    if not yuv:
        return None
    fs = width * height + 2 * ((width + 1) // 2) * ((height + 1) // 2)
    gold = np.fromfile(ypath, dtype=np.uint8).reshape(-1, fs)
    os.unlink(ypath)
    return gold


# (name, kwargs): small clips that together touch every tool the HEVC generator can emit
HEVC_MATRIX = [
    ("i_ctb16", dict(frames=3, intra_only=1, ctb=16)),
    ("i_ctb32_sao_pcm", dict(frames=2, intra_only=1, ctb=32, sao=3, pcm=1)),
    ("i_ctb64_qp22", dict(frames=2, intra_only=1, ctb=64, qp=22)),
    ("i_pcm2_tskip_bypass", dict(frames=2, intra_only=1, pcm=2, tskip=1, bypass=1)),
    ("i_sdh_qp20", dict(frames=2, intra_only=1, sdh=1, qp=20)),
    ("i_qpswing", dict(frames=2, intra_only=1, qp_swing=2, ctb=32, cb_off=3, cr_off=-4)),
    ("i_scaling_random_modes", dict(frames=2, intra_only=1, scaling=3, random_intra=1)),
    ("i_slices_cip_nolf", dict(frames=2, intra_only=1, slices=3, cip=1, lf_across=0, sao=1)),
    ("i_wpp", dict(frames=2, intra_only=1, wpp=1, ctb=16, qp_swing=1)),
    ("i_deep_tree_cover", dict(frames=2, intra_only=1, cover=1, th_intra=3, max_tb=16, strong=0)),
    ("i_odd_size", dict(width=134, height=90, frames=2, intra_only=1, ctb=32, sao=1)),
    ("p_refs2", dict(frames=6, gop=6, refs=2)),
    ("ibp", dict(frames=9, gop=9, bframes=2, refs=2)),
    ("ibp_amp_wp_ctb32", dict(frames=9, gop=9, bframes=2, refs=3, amp=1, ctb=32, wp=1)),
    ("pyramid_sao_sdh_cover", dict(frames=13, gop=13, bframes=3, bpyramid=1, refs=4, sao=1, sdh=1, cover=1)),
    ("merge2_pml3_notmvp", dict(frames=12, gop=6, bframes=1, refs=2, merge_cand=2, pml=3, tmvp=0, mvd_l1_zero=1)),
    ("rps_sps_pocwrap_ctb64", dict(frames=20, gop=20, bframes=2, rps_sps=1, poc_bits=4, cabac_init=2, qp_swing=2, ctb=64)),
    ("everything", dict(frames=8, gop=8, bframes=1, slices=2, wpp=1, ctb=16, cip=1, pcm=1, tskip=1, bypass=1, scaling=1, sao=1,
                        amp=1, sdh=1, deblock_off=2, beta=2, tc=-1, slice_cb_off=2, extra_nals=1, nal_length_size=2, hvc1=1)),
    ("deblock_off_th0", dict(frames=5, gop=5, bframes=1, deblock_off=1, th_inter=0, th_intra=0, max_tb=16)),
    ("merge1_p", dict(frames=5, gop=5, refs=1, merge_cand=1, qp=36)),
    # long-term reference pictures (8.3.2): the GOP's first picture stays in the DPB after the short-term sets have let go
    # of it - listed in the slice headers (1) or as the SPS's candidate (2); vectors into it are never scaled, a temporal
    # candidate across the long-term / short-term divide is no candidate (8.5.3.2.7-8), POC most significant bits signalled
    ("lt_p", dict(frames=10, gop=10, refs=2, long_term=1)),
    ("lt_sps_b_tmvp", dict(frames=13, gop=13, bframes=2, refs=2, long_term=2, tmvp=1, ctb=32)),
    ("lt_pyramid_pocwrap", dict(frames=40, gop=40, bframes=3, bpyramid=1, refs=3, long_term=1, poc_bits=4, sao=1)),
    # reference picture sets coded by prediction from an earlier one (7.3.7 / 7.4.8: what real encoders do) - in the SPS,
    # and a slice header's own set from one of the SPS's; ref_pic_lists_modification(): list entries from anywhere in the set
    ("inter_rps_sps", dict(frames=20, gop=20, bframes=2, refs=3, rps_sps=1, inter_rps=1)),
    ("inter_rps_slices_pyramid", dict(frames=20, gop=20, bframes=3, bpyramid=1, refs=4, rps_sps=2, inter_rps=1, ctb=32)),
    ("rplm_b", dict(frames=12, gop=12, bframes=2, refs=3, rplm=1)),
    ("rplm_lt_inter_rps_tmvp", dict(frames=16, gop=16, bframes=1, refs=4, rplm=1, long_term=1, rps_sps=2, inter_rps=1, tmvp=1)),
    # pictures kept in the DPB without being used by the current one (used_by_curr_pic_flag = 0: RefPicSetStFoll)
    ("keep_unused_p", dict(frames=16, gop=16, refs=3, keep_unused=1)),
    ("keep_unused_b_rplm", dict(frames=20, gop=20, bframes=2, refs=4, keep_unused=1, rplm=1, tmvp=1)),
    ("keep_unused_pyramid_lt", dict(frames=20, gop=20, bframes=3, bpyramid=1, refs=4, keep_unused=1, rps_sps=2, inter_rps=1, long_term=2)),
    # two temporal sub-layers: profile_tier_level with a sub-layer level, ordering info per sub-layer, TSA_N pictures with
    # nuh_temporal_id_plus1 = 2
    ("sub_layers_b", dict(frames=13, gop=13, bframes=2, refs=2, sub_layers=2, tmvp=1)),
    # the optional bits of a slice segment header (pic_output_flag, extra header bits, the header extension) in front of
    # several slices and wavefront rows; the SPS's long-term candidates chosen by lt_idx_sps
    ("sh_extras_slices_wpp", dict(frames=12, gop=6, bframes=2, refs=2, sh_extras=1, slices=3, wpp=1, ctb=16)),
    ("lt_idx_sps_sh_extras", dict(frames=14, gop=14, bframes=1, refs=2, long_term=3, sh_extras=1)),
    # two picture parameter sets taken in turn (another init_qp, other default list lengths), both in hvcC and repeated in band
    ("multi_pps", dict(frames=12, gop=6, bframes=2, refs=3, multi_pps=1, extra_nals=1)),
    ("multi_pps_qpswing", dict(frames=10, gop=10, refs=2, multi_pps=1, qp_swing=2, cabac_init=2)),
    # the smallest pictures: one coding tree block that the picture does not fill, a single row / column of blocks
    ("tiny_8x8", dict(width=8, height=8, frames=5, gop=5, bframes=1, refs=2, ctb=16, sao=1)),
    ("tiny_row_264x8", dict(width=264, height=8, frames=5, gop=5, bframes=1, refs=2, ctb=64, sao=1, amp=1)),
    ("tiny_column_16x136", dict(width=16, height=136, frames=5, gop=5, bframes=1, refs=2, ctb=16, sao=1, wpp=1)),
    # open GOPs, x265's default: every GOP but the first starts with a CRA picture (21) whose leading pictures (RASL_N)
    # predict across it from the GOP before; one IDR picture, the picture order count runs through (and wraps)
    ("open_gop_bb", dict(frames=36, gop=12, bframes=2, refs=2, open_gop=1, poc_bits=5)),
    ("open_gop_tools", dict(frames=27, gop=9, bframes=2, refs=3, open_gop=1, tmvp=1, sao=1, amp=1, ctb=32, wp=1, slices=2, extra_nals=1)),
    # closed GOPs with leading pictures: IDR_W_RADL pictures whose RADL_N pictures are displayed in front of them and
    # predict from them alone - negative picture order counts, and a work item whose first outputs are not its first sample
    ("radl_bb_tmvp", dict(frames=36, gop=12, bframes=2, refs=2, open_gop=2, tmvp=1, sao=1)),
]
