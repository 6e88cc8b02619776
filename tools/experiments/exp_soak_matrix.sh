# Development aid: determinism soak over stream variants that steer the row kernels differently
cd "$GRAFT_REPO_ROOT" || exit 1
for a in "--slices 3 --deblock 2" "--qp 40" "--qp 14" "--cabac 0 --t8x8 0" "--pcm-period 97" "--bframes 0 --refs 1"; do
  echo "=== $a"
  ENC_ARGS="$a" FRAMES=300 REPEAT=20 ROUNDS=4 timeout 400 python3 tools/soak.py 2>&1 | grep -v "^wrote\|rounds done" | tail -4
done
