# Development aid: the documented environment knobs must not change a single frame
cd "$GRAFT_REPO_ROOT" || exit 1
for e in HWANG_COPY_STREAM=1 HWANG_HOST_MV=1 HWANG_CU_PARTITION=0 HWANG_COMPLETION=hostfunc; do
  echo "=== $e"
  env $e FRAMES=300 REPEAT=20 ROUNDS=3 timeout 400 python3 tools/soak.py 2>&1 | grep -v "^wrote\|rounds done" | tail -3
done
