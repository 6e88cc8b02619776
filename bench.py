#!/usr/bin/env python3
"""Headline benchmark: sparse H.264 frame retrieval on MI355X.

Workload (BASELINE.json configs[1], "C1"): one synthetic 1080p30 High-profile
CABAC clip of 108 000 frames (1 h; 1 200 unique frames x 90 repeats of closed
GOPs; GOP 60, 2 B-frames between references).  A request = rows =
sorted(default_rng(20240601 + k).choice(N, round(0.01 N))) -> 1080 frames,
returned as RGB24 in HOST memory (SURVEY 8(d)'s timing rules: D2H included - what a
caller of the reference gets).  One "step" = one batched call through the C ABI
(hwang_retriever_retrieve_batch) of `--gpus` such requests (k = 0 .. gpus-1; at
N = 1 exactly C1): the GOP segments of all requests are pooled and fanned out
over the GPUs, so per-GPU work is fixed as N grows ("weak").  The encoded file is
resident in host memory (it is what the host-side entropy decoder reads); every
intermediate picture stays in HBM.  `value_hbm_resident` is the same request
with the frames left in HBM (rounds 1-3 printed that as `value`).

    python bench.py --gpus N --steps K --warmup W

N > 1 under torch.distributed.run (the driver's launch) = one rank per GPU:
every rank plans the whole batch and decodes its share of the work items
(hwang_retriever_set_shard: a deterministic longest-first split), writing the
rows of its share into its own output; ranks only meet in the timing barrier
and the MAX / SUM of times and counts - no collective on the data path.
`python bench.py --gpus N` without a launcher drives all N devices from one
process instead (shared work queue, lanes bound round-robin to the devices).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events
on the executor's streams (hwang_prof_*) in an extra pass after the timed
region (`end_to_end_frac`: SURVEY 8(d)'s whole-pipeline accounting; `frac_replayed_wide`:
the same kernel in the widest run of `replay` - the step's last pictures relaunched
straight from HBM, the host out of the way - beside `frac`, which stays the timed
configuration's);
`value_reference_api*` are the same request through the reference's Python API
(one DecoderAutomata): with the reference's own unmodified caller loop
(num_devices = 1, one interval per initialize: python/hwang/decoder.py:28,60-67)
and with this build's grouped caller; `parity_check` compares MD5s of rows the
TIMED configuration returned with the CPU oracle's decode of the same file;
`cpu_baseline` times the CPU oracle (a port, test
infrastructure) on a bounded sample of the same workload under the
reference's semantics (every sample of every touched GOP is decoded, wanted
frames converted to RGB), and carries the run-time FFmpeg probe's verdict.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
UNIQUE_FRAMES = 1200          # encoded once ...
GOP, BFRAMES = 60, 2
QP, FPS = 26, 30
SEED = 20240601
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CONFIG = "C1"

# BASELINE.json's other single-GPU configurations, selectable with --config for profiles/ (the default line, the one
# the driver records, is always C1): per configuration the clip and which rows a request asks for.
CONFIGS = {
    "C1": dict(width=1920, height=1080, unique=1200, frames=108000, qp=26, fps=30, rows="1pct",
               what="single 1080p30 High-profile CABAC mp4, 1 % uniform-random sample"),
    "C2": dict(width=3840, height=2160, unique=240, frames=3600, qp=26, fps=30, rows="stride10",
               what="single 4K30 High-profile CABAC mp4, every 10th frame"),
    "C4": dict(width=3840, height=2160, unique=120, frames=1200, qp=18, fps=60, rows="dense",
               what="one of the 4K60 CABAC high-bitrate (QP 18) streams, every frame"),
}


def select_config(name, frames=None):
    global WIDTH, HEIGHT, UNIQUE_FRAMES, QP, FPS, CONFIG, _S, SPAN_KERNELS
    c = CONFIGS[name]
    CONFIG = name
    WIDTH, HEIGHT, UNIQUE_FRAMES, QP, FPS = c["width"], c["height"], c["unique"], c["qp"], c["fps"]
    mbs = ((WIDTH + 15) // 16) * ((HEIGHT + 15) // 16)
    _S = mbs * 384
    # (the records-only kernels - k_mv, k_bs, k_residual - run as a stage of their own since round 3, so k_bs and
    # k_residual are spans of their own and no longer inside the deblock / intra spans)
    SPAN_KERNELS = {"mv_derivation": (["k_mv"], mbs * 176), "inter_pred": (["k_inter"], 2 * _S),
                    "intra_pred": (["k_intra"], _S), "deblock": (["k_deblock"], 2 * _S),
                    "boundary_strength": (["k_bs"], mbs * (176 + 96)), "residual_idct": (["k_residual"], mbs * 768),
                    "yuv420_to_rgb24": (["convert_batch"], 9 * WIDTH * HEIGHT // 2)}
    return frames if frames else c["frames"]


def request_rows(np, n, k):
    """Rows of request k over a clip of n frames, as the selected configuration samples them."""
    mode = CONFIGS[CONFIG]["rows"]
    if mode == "1pct":
        return sorted(np.random.default_rng(SEED + k).choice(n, max(1, round(0.01 * n)), replace=False).tolist())
    if mode == "stride10":
        return list(range(k % 10, n, 10))
    return list(range(n))


def corpus_path(total_frames):
    tag = "" if UNIQUE_FRAMES == CONFIGS[CONFIG]["unique"] else "_u%d" % UNIQUE_FRAMES
    return "/tmp/hwang_bench_%s_%dx%d_%d_gop%d_b%d_qp%d%s.mp4" % (CONFIG.lower(), WIDTH, HEIGHT, total_frames, GOP, BFRAMES, QP, tag)


def make_corpus(total_frames):
    """1080p30 clip: UNIQUE_FRAMES frames encoded by the in-tree encoder and
    repeated (each repeat is an independent run of closed GOPs)."""
    path = corpus_path(total_frames)
    if os.path.exists(path):
        return path
    enc = os.path.join(ROOT, "corpus", "_build", "h264_encoder")
    if not os.path.exists(enc):
        raise SystemExit("bench.py: corpus encoder missing; run __graft_entry__.build() first")
    unique = min(UNIQUE_FRAMES, total_frames)
    repeat = max(1, total_frames // unique)
    tmp = path + ".tmp.%d" % os.getpid()
    subprocess.check_call([enc, "--width", str(WIDTH), "--height", str(HEIGHT), "--frames", str(unique),
                           "--gop", str(GOP), "--bframes", str(BFRAMES), "--fps", str(FPS), "--qp", str(QP),
                           "--threads", str(min(64, os.cpu_count() or 8)), "--repeat", str(repeat), "--out", tmp],
                          stdout=subprocess.DEVNULL)
    os.replace(tmp, path)
    return path


# --------------------------------------------------------------- CPU baseline
def cpu_gop_worker(spec_path):
    """Child process (`bench.py --cpu-worker spec.json`): decodes whole GOPs
    with the oracle the way the reference's software backend does
    (software_video_decoder.cpp:103-457 driven by DecoderAutomata: every sample
    of the interval is fed, unwanted frames are discarded, wanted ones are
    converted to RGB24)."""
    import hashlib
    from oracle import oracle_py  # cpu_baseline leg only
    with open(spec_path) as f:
        spec = json.load(f)
    avcc = bytes.fromhex(spec["avcc"])
    returned = fed = 0
    md5s = {}
    t0 = time.time()
    with open(spec["path"], "rb") as f:
        for gop in spec["gops"]:
            dec = oracle_py.OracleDecoder(avcc)
            frames = []
            for off, size in gop["samples"]:
                f.seek(off)
                frames += dec.decode(f.read(size))
                fed += 1
            frames += dec.flush()
            for k, row in zip(gop["wanted"], gop["rows"]):
                buf, w, h, _ = frames[k]
                rgb = oracle_py.i420_to_rgb(buf, w, h)
                returned += 1
                if spec.get("md5"):   # (the untimed extra run that feeds bench.py's parity_check)
                    md5s[str(row)] = hashlib.md5(rgb.tobytes()).hexdigest()
            dec.close()
    print(json.dumps({"returned": returned, "fed": fed, "seconds": time.time() - t0, "md5": md5s}))


def cpu_baseline(path, index, rows, workers, max_gops=64, repeats=3):
    """The oracle under the reference's semantics on the first `max_gops`
    touched GOPs of the request, `workers` single-threaded decoders pulling GOPs
    round-robin, median of `repeats` runs."""
    import tempfile
    offs, sizes = index.sample_offsets(), index.sample_sizes()
    kfs = index.keyframe_indices() + [index.frames()]
    gops, ki = [], 0
    for r in rows:
        while kfs[ki + 1] <= r:
            ki += 1
        if gops and gops[-1]["gop"] == ki:
            gops[-1]["wanted"].append(r - kfs[ki])
            gops[-1]["rows"].append(r)
        elif len(gops) < max_gops:
            gops.append({"gop": ki, "samples": [(offs[s], sizes[s]) for s in range(kfs[ki], kfs[ki + 1])],
                         "wanted": [r - kfs[ki]], "rows": [r]})
        else:
            break
    workers = max(1, min(workers, len(gops)))
    avcc = bytes(index.metadata_bytes()).hex()
    runs = []
    t_all = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(workers):
            with open(os.path.join(tmp, "w%d.json" % i), "w") as f:
                json.dump({"path": path, "avcc": avcc, "gops": gops[i::workers]}, f)
        row_md5 = {}
        for rep_i in range(repeats + 1):
            # one more, untimed, run that also hashes what it returns: the oracle's answer for these rows
            hashing = rep_i == repeats
            if hashing:
                for i in range(workers):
                    with open(os.path.join(tmp, "w%d.json" % i), "w") as f:
                        json.dump({"path": path, "avcc": avcc, "gops": gops[i::workers][:2], "md5": True}, f)
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                                       os.path.join(tmp, "w%d.json" % i)], stdout=subprocess.PIPE)
                     for i in range(workers)]
            res = []
            for p in procs:
                out, _ = p.communicate()
                if p.returncode:
                    raise SystemExit("bench.py: cpu_baseline worker failed")
                res.append(json.loads(out.decode().strip().splitlines()[-1]))
            if hashing:
                for r in res:
                    row_md5.update(r["md5"])
                break
            busy = max(r["seconds"] for r in res)  # slowest worker's decode time (interpreter start-up excluded)
            runs.append((sum(r["returned"] for r in res) / busy, sum(r["fed"] for r in res) / busy,
                         sum(r["returned"] for r in res), sum(r["fed"] for r in res)))
    runs.sort()
    med = runs[len(runs) // 2]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import ffmpeg_probe
        probe = ffmpeg_probe.probe(deep=False)
        probe = {"available": probe["available"], "summary": probe["summary"]}
    except Exception as e:  # noqa: BLE001
        probe = {"available": False, "summary": "probe failed: %r" % (e,)}
    return {
        "value": round(med[0], 3),
        "unit": "frames/s",
        "cores": workers,
        "kind": "port",
        "sample": "%d touched GOPs of the same request (%d samples fed, %d wanted frames returned as RGB24), %d "
                  "single-threaded oracle decoders, reference semantics (whole GOP decoded); median of %d runs, "
                  "%.1f s wall in total" % (len(gops), med[3], med[2], workers, repeats, time.time() - t_all),
        "fed_samples_per_s": round(med[1], 2),
        "runs_frames_per_s": [round(r[0], 3) for r in runs],
        "ffmpeg_probe": probe,
        "_row_md5": row_md5,   # (popped by main(): the parity check's expected values, not part of the printed block)
    }


# Kernels inside each HIP-event span, and the span's algorithmic bytes per
# 1080p picture (S = 1.5 * 1920 * 1088).
_S = 1920 * 1088 * 3 // 2
SPAN_KERNELS = {}
select_config("C1")


def pmc_traffic_per_launch(span, alg_bytes_per_launch):
    """HBM bytes per launch of `span` from the committed rocprofv3 --pmc passes
    (profiles/pmc_summary.json: FETCH_SIZE x 2 + WRITE_SIZE per picture, see
    tools/pmc_summary.py), scaled to this run's pictures per launch.  None if
    no PMC summary has been committed."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if span not in SPAN_KERNELS or not os.path.exists(path) or CONFIG != "C1":
        return None   # (the committed PMC passes are of the C1 run: bytes per 1080p picture)
    pmc = json.load(open(path))
    kernels, alg_per_picture = SPAN_KERNELS[span]
    per_picture = 0
    for k in kernels:
        if k not in pmc or pmc[k].get("fetch_bytes_per_picture") is None:
            return None
        per_picture += pmc[k]["fetch_bytes_per_picture"] + pmc[k]["write_bytes_per_picture"]
    return round(per_picture * alg_bytes_per_launch / alg_per_picture)


def effective_cpus():
    """Hardware threads, capped by the cgroup CPU quota (cpu.max) if any."""
    n = float(len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        pass
    return max(1, int(n + 0.5))


# ----------------------------------------------------------------------- main
def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return cpu_gop_worker(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C1",
                    help="BASELINE.json configuration; C1 is the headline (and the default); C2 / C4 are for profiles/")
    ap.add_argument("--frames", type=int, default=int(os.environ.get("HWANG_BENCH_FRAMES", 0)),
                    help="clip length; default: the configuration's (C1: 108000 = 1 h of 1080p30)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("HWANG_BENCH_LANES", 0)))
    ap.add_argument("--cpu-workers", type=int, default=int(os.environ.get("HWANG_BENCH_CPU_WORKERS", 16)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-output", action="store_true", help="skip the extra pass that leaves the frames in HBM (N = 1)")
    ap.add_argument("--hbm-output", action="store_true",
                    help="time the frames-stay-in-HBM variant as `value` (what rounds 1-3 printed); the line says so")
    ap.add_argument("--no-reference-api", action="store_true",
                    help="skip the extra pass through hwang.Decoder(...).retrieve(rows), the reference's own API (N = 1)")
    ap.add_argument("--reference-api-threads", type=int, default=int(os.environ.get("HWANG_BENCH_REF_THREADS", 0)),
                    help="num_devices handed to the reference API's DecoderAutomata (default: the usable CPUs)")
    ap.add_argument("--host-counters", action="store_true",
                    help="HWANG_HOST_COUNTERS=1: the lane threads count their instructions / cycles / mispredicted branches "
                         "(perf_event_open per thread and call; off by default - the line is measured as a user runs the library)")
    ap.add_argument("--no-hevc", action="store_true", help="skip the HEVC leg (tools/hevc_bench.py: SURVEY section 8 row f4, N = 1)")
    ap.add_argument("--unique", type=int, default=0,
                    help="unique frames encoded before the clip repeats (default: the configuration's; C1: 1200 x 90)")
    args = ap.parse_args()
    if args.host_counters:
        os.environ["HWANG_HOST_COUNTERS"] = "1"
    args.frames = select_config(args.config, args.frames)
    if args.unique:
        global UNIQUE_FRAMES
        UNIQUE_FRAMES = args.unique

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # Two ways to use N GPUs: torch.distributed.run with N ranks (the driver's launch; every rank decodes
    # its share of the batch), or one process driving all N devices (`python bench.py --gpus N`).
    in_process_devices = 1
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            in_process_devices = args.gpus
        else:
            args.gpus = world

    import numpy as np
    import hwang_amd as hw
    from hwang_amd._capi import lib
    from hwang_amd.decoder import BatchRetriever

    # corpus + index (rank 0 generates; the others wait for the file)
    path = corpus_path(args.frames)
    if rank == 0:
        make_corpus(args.frames)
    else:
        while not os.path.exists(path):
            time.sleep(0.5)
    index = hw.index_video(path)
    n = index.frames()
    # The job: ONE batched call of `gpus` requests over the clip, request k = the 1 % sample with seed
    # SEED + k (N = 1: exactly configuration C1).  All requests' GOP segments are pooled and fanned out
    # over the GPUs; per-GPU work is fixed as N grows ("weak").
    nreq = args.gpus
    row_sets = [request_rows(np, n, k) for k in range(nreq)]

    # CPU baseline first (rank 0, N = 1 only): spawns processes, so it runs
    # before this process touches the GPU.
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and args.config == "C1":
        cpu = cpu_baseline(path, index, row_sets[0], max(1, min(args.cpu_workers, effective_cpus())))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible (the decoder has no CPU fallback)")
    # development aid: HWANG_BENCH_DEVICE_IDS=0,0 runs the one-process / several-device-slots path on a single GPU
    forced_ids = [int(x) for x in os.environ.get("HWANG_BENCH_DEVICE_IDS", "").split(",") if x.strip()]
    if in_process_devices > 1 and not forced_ids and in_process_devices > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d but only %d visible" % (in_process_devices, torch.cuda.device_count()))
    # One rank per GPU.  If the launcher started more ranks than there are GPUs (a development box with one
    # GPU: the only way to exercise this path there) the ranks share devices and meet over gloo instead of
    # RCCL, which refuses two ranks on one device; the line then says so in config.launch.
    oversubscribed = world > torch.cuda.device_count()
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    red_dev = "cpu" if oversubscribed else "cuda"
    if world > 1:
        if oversubscribed:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    # (round 5: the retriever drives these lanes with ONE thread per usable CPU when the queue is long - see
    # config.lane_threads - the lane count below is how many decode streams are in flight, i.e. the launch width)
    # 3 lanes per usable CPU (a lane waits for the GPU about as long as it parses;
    # the CPUs are the bottleneck: measured 32 / 40 / 48 / 64 lanes on 16 CPUs ->
    # 660 / 676 / 698 / 651 frames/s); with several ranks on one host the CPUs are shared
    # (4K: two per CPU - a lane's working set grows with the picture and beyond that the parsers evict each other,
    # retriever.cpp; measured 16 / 24 / 32 / 48 lanes -> 485 / 528 / 562 / 477 frames/s)
    per_cpu = 3 if WIDTH * HEIGHT <= 2 * 1920 * 1088 else 2
    lanes = args.lanes or max(2, min(128, per_cpu * effective_cpus() // world))
    devices = (forced_ids or list(range(in_process_devices))) if in_process_devices > 1 else [device_index]
    if world > 1 and "HWANG_RETRIEVER_THREADS" not in os.environ:
        # a rank per GPU on one host: the ranks share the host's CPUs, each drives its lanes with its share of them
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        os.environ["HWANG_RETRIEVER_THREADS"] = str(max(1, effective_cpus() // max(1, local_world)))
    br = BatchRetriever(device_ids=devices, num_lanes=lanes, shard=(rank, world) if world > 1 else None)
    # request k's frames land in HOST memory (D2H is part of the job; N ranks: each rank writes the rows
    # of its share into its own copy of the batch's output - pageable memory, so only the pages of its share ever
    # exist; nothing is gathered - no data-path collective).  --hbm-output: a (rows, H, W, 3) tensor in HBM instead.
    fobj = open(path, "rb")

    def make_outs(in_hbm):
        if in_hbm:
            return [torch.empty((len(r), HEIGHT, WIDTH, 3), dtype=torch.uint8,
                                device="cuda:%d" % (devices[k % len(devices)] if in_process_devices > 1 else device_index))
                    for k, r in enumerate(row_sets)]
        return [torch.empty((len(r), HEIGHT, WIDTH, 3), dtype=torch.uint8) for r in row_sets]
    outs = make_outs(args.hbm_output)
    requests = [(index, fobj, row_sets[k], outs[k].data_ptr(), outs[k].numel()) for k in range(nreq)]

    def step():
        br.retrieve_into(requests)

    def fence():
        for d in devices:
            torch.cuda.synchronize(d)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    c0 = os.times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    c1 = os.times()
    # host side of the story (the path is host bound): CPU seconds this
    # rank's process used per wall second of the timed region, all threads
    host_cpus_busy = ((c1.user - c0.user) + (c1.system - c0.system)) / elapsed
    stats = br.stats()
    dev_stats = br.device_stats()
    # what the lane threads - the entropy decoders - executed in the last step, from the CPU's own counters (rank 0's)
    hc = br.host_counters()
    host_lane_threads = None
    if hc["cpu_ns"] and stats["pictures_reconstructed"]:
        npics = stats["pictures_reconstructed"]   # (not `n`: that is the clip's frame count, printed in config.workload)
        host_lane_threads = {"cpu_ms_per_picture": round(hc["cpu_ns"] / 1e6 / npics, 3)}
        if hc["cycles"]:
            host_lane_threads.update({"instructions_per_picture": int(hc["instructions"] / npics), "cycles_per_picture": int(hc["cycles"] / npics),
                                      "branch_misses_per_picture": int(hc["branch_misses"] / npics),
                                      "ipc": round(hc["instructions"] / hc["cycles"], 2),
                                      "ghz_while_running": round(hc["cycles"] / hc["cpu_ns"], 2),
                                      "what": "user-mode counters of the retriever's lane threads over the last timed step (perf_event_open, "
                                              "per thread); the single-thread figures of the parser alone are in profiles/r05_host_perf_counters.txt"})
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([stats["frames_returned"], stats["pictures_reconstructed"], stats["intervals"], host_cpus_busy],
                           dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(gathered, cnt)
        total_rows = int(sum(g[0].item() for g in gathered))
        total_pics = int(sum(g[1].item() for g in gathered))
        per_gpu = [{"frames_returned": int(g[0].item()), "pictures_reconstructed": int(g[1].item()),
                    "items": int(g[2].item()), "host_cpus_busy": round(g[3].item(), 2)} for g in gathered]
    else:
        total_rows, total_pics = stats["frames_returned"], stats["pictures_reconstructed"]
        per_gpu = [dict(d, host_cpus_busy=round(host_cpus_busy, 2)) for d in dev_stats] if len(devices) > 1 else None
    assert total_rows == sum(len(r) for r in row_sets), "every row of every request must have been returned"
    host_cpus_total = round(sum(g["host_cpus_busy"] for g in per_gpu), 2) if (per_gpu and world > 1) else round(host_cpus_busy, 2)

    # Strong scaling beside the weak line (N > 1): ONE request - request 0, north_star's C1 - whose GOP work items
    # all N devices share (one process per GPU: the deterministic split; one process: the common queue).
    strong = None
    if args.gpus > 1:
        one = [requests[0]]
        br.retrieve_into(one)
        fence()
        ts = time.perf_counter()
        for _ in range(args.steps):
            br.retrieve_into(one)
        fence()
        s_elapsed = time.perf_counter() - ts
        if world > 1:
            t = torch.tensor([s_elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            s_elapsed = float(t.item())
        strong = {"scaling": "strong", "value": round(len(row_sets[0]) * args.steps / s_elapsed, 2), "unit": "frames/s",
                  "ms_per_step": round(1000.0 * s_elapsed / args.steps, 2),
                  "work": "ONE request of %d rows, its GOP work items divided over %d GPUs" % (len(row_sets[0]), args.gpus)}

    # What the timed configuration returned, checked against the CPU oracle's decode of the same file (untimed): MD5 of
    # the rows of request 0 that the cpu_baseline workers hashed.
    parity = None
    if rank == 0 and cpu and cpu.get("_row_md5"):
        import hashlib
        want = cpu.pop("_row_md5")
        pos = {r: k for k, r in enumerate(row_sets[0])}
        host0 = outs[0].cpu() if args.hbm_output else outs[0]
        bad = [int(r) for r, h in want.items() if hashlib.md5(host0[pos[int(r)]].numpy().tobytes()).hexdigest() != h]
        parity = {"rows_checked": len(want), "mismatches": len(bad), "mismatched_rows": bad[:8],
                  "against": "oracle/ (CPU restatement) decoding the same file: MD5 of the RGB24 frame per row"}
        del host0
    if cpu:
        cpu.pop("_row_md5", None)

    # roofline pass: one more step with the per-kernel HIP-event spans enabled
    roofline = None
    kernels = {}
    value_other = None
    value_pinned = None
    reference_api = None
    reference_api_unmodified = None
    replay = None
    if rank == 0:
        if len(devices) == 1:
            lib.hwang_replay_keep(devices[0], 4 * lanes + 64)   # the executor remembers this step's last pictures (replay, below)
        lib.hwang_prof_reset()
        lib.hwang_prof_enable(1)
        step()
        for d in devices:
            torch.cuda.synchronize(d)
        lib.hwang_prof_enable(0)
        prof_stats = br.stats()
        best = None
        alg = {}
        for k in range(lib.hwang_prof_num_kernels()):
            o = (C.c_double * 3)()
            lib.hwang_prof_read(k, o)
            if not o[0]:
                continue
            name = lib.hwang_prof_kernel_name(k).decode()
            alg[name] = o[2]
            kernels[name] = {"launches": int(o[0]), "avg_us": round(o[1] / o[0], 1),
                             "alg_GBps": round(o[2] / (o[1] * 1e-6) / 1e9, 1)}
            if best is None or o[1] > best[1]:
                best = (name, o[1], o[2], o[0])
        if best:
            achieved = best[2] / (best[1] * 1e-6) / 1e9
            # SURVEY 8(d) end-to-end accounting for this rank's step: S (1 + r) per reconstructed picture
            # (= the inter span's algorithmic bytes, + S per intra-only picture, one per work item) + B_in
            # (bytes actually uploaded) + 4.5 W H per returned frame, over wall time x 8 TB/s.
            e2e_bytes = alg.get("inter_pred", 0.0) + _S * prof_stats["intervals"] + prof_stats["bytes_uploaded"] + \
                4.5 * WIDTH * HEIGHT * prof_stats["frames_returned"]
            roofline = {"bound": "hbm", "kernel": best[0], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc_traffic_per_launch(best[0], best[2] / best[3]),
                        "traffic_source": "profiles/pmc_summary.json (separate rocprofv3 --pmc passes of this bench command on the "
                                          "tree that last committed that file): HBM bytes per picture of the kernel, scaled to this "
                                          "run's pictures per launch - a ratio carried over, not counters of this run",
                        "algorithmic_bytes_per_launch": round(best[2] / best[3]),
                        "launches": int(best[3]), "avg_launch_us": round(best[1] / best[3], 1),
                        "end_to_end_frac": round(e2e_bytes / (prof_stats["wall_ns"] * 1e-9) / (HBM_PEAK_GBS * 1e9), 5),
                        "end_to_end_bytes_per_step": round(e2e_bytes)}
        # The GPU with the host out of the way (N = 1): the last pictures of that step - their records and surfaces are
        # still in HBM, the lanes idle - launched again through the same kernels at widths the host's entropy decoders
        # never produce (`roofline` above is what the host-bound run shows; this is what the kernels can do).
        if len(devices) == 1 and world == 1:
            replay = {"what": "the sample-stage kernels (k_inter, k_intra, k_deblock) and the records-only ones relaunched for the "
                              "last pictures of a step straight from HBM, back to back, `width` pictures per launch", "runs": []}
            for w in (48, 1 << 20):
                o = (C.c_double * 4)()
                err = C.create_string_buffer(256)
                if lib.hwang_replay_run(devices[0], w, 3, 0, o, err, 256):
                    replay["runs"].append({"error": err.value.decode()})
                    break
                lib.hwang_prof_reset()
                lib.hwang_prof_enable(1)
                rc = lib.hwang_replay_run(devices[0], w, 30, 0, o, err, 256)
                lib.hwang_prof_enable(0)
                if rc:
                    replay["runs"].append({"error": err.value.decode()})
                    break
                sp = {}
                for k in range(lib.hwang_prof_num_kernels()):
                    q = (C.c_double * 3)()
                    lib.hwang_prof_read(k, q)
                    if q[0]:
                        sp[lib.hwang_prof_kernel_name(k).decode()] = {"avg_us": round(q[1] / q[0], 1), "alg_GBps": round(q[2] / (q[1] * 1e-6) / 1e9, 1),
                                                                      "hbm_frac": round(q[2] / (q[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                stage_us = sum(sp[n]["avg_us"] for n in ("inter_pred", "intra_pred", "deblock") if n in sp)
                replay["runs"].append({"width": int(o[1]), "pictures_available": int(o[0]), "launches": int(o[2]),
                                       "pictures_per_s_all_kernels_one_stream": round(o[1] * o[2] / o[3], 1),
                                       "pictures_per_s_sample_stage": round(o[1] / (stage_us * 1e-6), 1) if stage_us else None,
                                       "kernels": sp})
            lib.hwang_replay_keep(devices[0], 0)
            lib.hwang_prof_reset()
            # the same kernel as `roofline`, with the host out of the way: the widest replay run's fraction, beside the
            # host-bound one (`frac` stays what the contract defines: the timed region's launches)
            if roofline:
                fr = [r["kernels"][roofline["kernel"]]["hbm_frac"] for r in replay["runs"] if "kernels" in r and roofline["kernel"] in r["kernels"]]
                if fr:
                    roofline["frac_replayed_wide"] = max(fr)
        # The other output placement (N = 1): frames left in HBM when `value` is the host-memory rate, and vice versa.
        if args.gpus == 1 and not args.no_hbm_output and args.config == "C1":
            other = (torch.empty((len(row_sets[0]), HEIGHT, WIDTH, 3), dtype=torch.uint8) if args.hbm_output else
                     torch.empty((len(row_sets[0]), HEIGHT, WIDTH, 3), dtype=torch.uint8, device="cuda:%d" % device_index))
            oreq = [(index, fobj, row_sets[0], other.data_ptr(), other.numel())]
            br.retrieve_into(oreq)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(2):
                br.retrieve_into(oreq)
            torch.cuda.synchronize()
            value_other = round(2 * len(row_sets[0]) / (time.perf_counter() - th), 2)
            del other
            # ... and the caller's host buffer PAGE-LOCKED (torch pin_memory = hipHostMalloc): the D2H copy lands in it,
            # no 6 MB memcpy per frame on a lane thread (`value` keeps the pageable buffer the reference's callers have)
            if not args.hbm_output:
                try:
                    pinned = torch.empty((len(row_sets[0]), HEIGHT, WIDTH, 3), dtype=torch.uint8, pin_memory=True)
                except RuntimeError as e:  # not enough lockable memory on this machine
                    pinned = None
                    sys.stderr.write("page-locked output skipped: %s\n" % str(e).splitlines()[0])
                if pinned is not None:
                    preq = [(index, fobj, row_sets[0], pinned.data_ptr(), pinned.numel())]
                    br.retrieve_into(preq)
                    torch.cuda.synchronize()
                    th = time.perf_counter()
                    for _ in range(2):
                        br.retrieve_into(preq)
                    torch.cuda.synchronize()
                    value_pinned = round(2 * len(row_sets[0]) / (time.perf_counter() - th), 2)
                    del pinned
        # The drop-in path itself: the reference's Python API (python/hwang/decoder.py) over the pybind11 module - ONE
        # DecoderAutomata driving ONE AMD backend; frames come back as numpy arrays in host memory, as the reference
        # returns them.  Two callers:
        #  * `value_reference_api_unmodified`: the reference's own loop, unchanged - DecoderAutomata(handle, 1, type)
        #    (decoder.py:28) and per planner interval one initialize([interval]) + one get_frames (decoder.py:60-67);
        #    hwang.Decoder(..., num_devices=1, interval_group_bytes=0) is that loop.  Also with num_devices = the usable
        #    CPUs (a one-character change in decoder.py:28).
        #  * `value_reference_api`: this build's caller - num_devices = the usable CPUs, the seek plan's intervals
        #    handed to initialize() in groups (the C++ API takes a list), the next group read ahead.
        if args.gpus == 1 and not args.no_reference_api and args.config == "C1":
            import hwang
            t_ref = args.reference_api_threads or effective_cpus()
            ridx = hwang.VideoIndex.deserialize(index.serialize())
            rows0 = row_sets[0]
            try:
                avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] * 1024
            except (OSError, IndexError, ValueError):
                avail = 0
            # ONE retrieve(rows) call as the reference's API is used - its answer is a list of 1 080 arrays, 6.7 GB of
            # host memory; on a machine that cannot spare that, four calls of a quarter each
            nparts = 1 if avail > 4 * len(rows0) * HEIGHT * WIDTH * 3 else 4
            step_rows = (len(rows0) + nparts - 1) // nparts
            parts = [rows0[i:i + step_rows] for i in range(0, len(rows0), step_rows)]

            def run_api(num_devices, group_bytes, what):
                rdec = hwang.Decoder(path, video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=num_devices,
                                     interval_group_bytes=group_bytes)
                rdec.retrieve(rows0[:40])   # first-use costs (surfaces, staging, worker threads) stay outside
                # timed: the retrieve() calls themselves; letting go of a call's answer (munmap of up to 6.7 GB of
                # touched pages: ~0.6 s) is the caller's business and happens between them
                got, t_api, cpu_api = 0, 0.0, 0.0
                for part in parts:
                    cr0, tr0 = os.times(), time.perf_counter()
                    frames = rdec.retrieve(part)
                    tr1, cr1 = time.perf_counter(), os.times()
                    t_api += tr1 - tr0
                    cpu_api += (cr1.user - cr0.user) + (cr1.system - cr0.system)
                    got += len(frames)
                    assert frames[0].shape == (HEIGHT, WIDTH, 3)
                    del frames
                assert got == len(rows0)
                del rdec
                return {"value": round(got / t_api, 2), "unit": "frames/s", "num_devices": num_devices,
                        "host_cpus_busy": round(cpu_api / t_api, 2), "calls": len(parts), "api": what}
            reference_api_unmodified = {
                # what differs from the reference's caller although its loop is kept line for line
                "deviations": ["hwang/decoder.py passes consume=True to initialize() (the bytes are moved, not copied)",
                               "num_devices=1 selects the backend's shared pool of entropy workers (up to 16 threads) where the "
                               "reference's software backend takes it as libavcodec's thread_count = 1"],
                "num_devices_1": run_api(1, 0, "hwang.Decoder(path, device_type=GPU).retrieve(rows) exactly as python/hwang/"
                                               "decoder.py:28,60-67 drive it: DecoderAutomata(handle, 1, AMD), one initialize + "
                                               "get_frames per planner interval (the backend picks its entropy workers itself)"),
                "num_devices_cpus": run_api(t_ref, 0, "the same per-interval loop with num_devices = %d" % t_ref)}
            reference_api = run_api(t_ref, 128 << 20,
                                    "hwang.Decoder(path, device_type=GPU, num_devices=%d).retrieve(rows): pybind11 "
                                    "hwang._python.DecoderAutomata -> VideoDecoderFactory -> AMD backend; intervals in 128 MB "
                                    "groups per initialize(); numpy frames in host memory" % t_ref)
    # SURVEY section 8 row f4 beside the north-star path: the same retrieve call on an HEVC Main clip of the same shape
    # (tools/hevc_bench.py; its own corpus encoder's output - HEVC parity is unpinned).  Not `value`.
    hevc = None
    if rank == 0 and args.gpus == 1 and not args.no_hevc and args.config == "C1":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hevc_bench
            hevc = hevc_bench.run(frames=36000, lanes=32, steps=3, cpu=not args.no_cpu_baseline, thread_per_lane=True)
        except Exception as e:  # the leg is an extra: the contract's line does not depend on it
            hevc = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        value = total_rows * args.steps / elapsed
        unique = min(UNIQUE_FRAMES, n)
        assert n == index.frames() == args.frames, "config.workload must name the clip that was decoded"
        print(json.dumps({
            "metric": "decoded frames/sec, 1080p H.264 1%-sparse sample, 1/2/4/8 MI355X vs FFmpeg CPU",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %s; %dx%d, %d frames (%d unique x %d repeats of closed GOPs), GOP %d, %d B-frames, QP %d; "
                                   "one batched call of %d request(s) (%d rows each%s), GOP segments pooled and fanned "
                                   "out over %d GPU(s); frames returned as RGB24 in %s"
                                   % (CONFIG, CONFIGS[CONFIG]["what"], WIDTH, HEIGHT, n, unique, max(1, n // unique), GOP,
                                      BFRAMES, QP, nreq, len(row_sets[0]),
                                      ", seeds %d..%d" % (SEED, SEED + nreq - 1) if CONFIGS[CONFIG]["rows"] == "1pct" else "",
                                      args.gpus, "HBM (--hbm-output)" if args.hbm_output else "host memory (D2H inside the timed region)"),
                       "frames": n, "unique_frames": unique, "repeats": max(1, n // unique), "rows_per_request": len(row_sets[0]),
                       "launch": ("one process, %d devices" % len(devices) if in_process_devices > 1 else
                                  "%d rank(s), one GPU each" % world if not oversubscribed else
                                  "%d ranks sharing %d GPU(s) (development run: gloo, not a scaling number)"
                                  % (world, torch.cuda.device_count())),
                       "lanes_per_gpu": lanes // len(devices), "host_cpus": effective_cpus(),
                       # (retriever.cpp: a thread per usable CPU drives the lanes in turn when there are eight work items
                       # per lane or more, else a thread per lane; HWANG_LANES_PER_THREAD overrides - the number is the
                       # library's own, hwang_retriever_stats[9])
                       "lane_threads": stats["lane_threads"],
                       "sharding": "GOP work items, longest-first split, no collective on the data path"},
            "value_host_output": round(value, 2) if not args.hbm_output else value_other,
            "value_hbm_resident": value_other if not args.hbm_output else round(value, 2),
            "value_host_output_page_locked": value_pinned,
            "value_reference_api_unmodified": reference_api_unmodified,
            "value_reference_api": reference_api,
            "hevc_main_1080p": hevc,
            "parity_check": parity,
            "pictures_reconstructed_per_s": round(total_pics * 1.0 / (elapsed / args.steps), 1),
            "host_lane_threads": host_lane_threads,
            "samples_fed_per_s_rank0": round(stats["samples_fed"] / (elapsed / args.steps), 1),
            "host_cpus_busy_rank0": round(host_cpus_busy, 2),
            "host_cpus_total": host_cpus_total,
            "strong_scaling": strong,
            "host_cpu_ms_per_picture": round(1000.0 * host_cpus_busy * (elapsed / args.steps) /
                                             max(1, stats["pictures_reconstructed"]), 3),
            "upload_MB_per_picture": round(stats["bytes_uploaded"] / 1e6 / max(1, stats["pictures_reconstructed"]), 3),
            "per_gpu": per_gpu,
            "roofline": roofline,
            "kernels": kernels,
            "replay": replay,
            "cpu_baseline": cpu,
        }))
    del br


if __name__ == "__main__":
    main()
