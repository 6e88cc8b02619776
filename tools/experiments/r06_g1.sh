set -x
( find / \( -name 'librocdecode*' -o -name 'libva*.so*' -o -name '*_drv_video.so' -o -name 'libavcodec*' -o -name 'libvdpau*' \) 2>/dev/null; echo "--- /dev/dri"; ls -l /dev/dri 2>&1; echo "--- which"; which ffmpeg ffprobe vainfo gst-launch-1.0 2>&1 ) > gpurun_out/r06_vcn_probe.txt 2>&1
python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
for f in 60 100; do HWANG_LINGER_FILL=$f python bench.py --no-reference-api > gpurun_out/r06_bench_fill$f.json 2> gpurun_out/r06_bench_fill$f.err; done
python bench.py --no-reference-api > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err
