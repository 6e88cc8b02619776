# Development aid: determinism soak against several builds of the library
cd "$GRAFT_REPO_ROOT" || exit 1
for v in prek pre mv db in amd; do
  echo "=== lib $v"
  HWANG_AMD_LIB=$GRAFT_REPO_ROOT/hwang_amd/libhwang_$v.so ROUNDS=5 timeout 400 python3 tools/soak.py 2>&1 | grep -v "^wrote" | tail -6
done
