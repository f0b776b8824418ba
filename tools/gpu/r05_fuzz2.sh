# the differential fuzzers once more on the final build, with the split dot products of the blocked backward slabs FORCED on every level
# (HIPMF_SPLIT_TASKS / HIPMF_SPLIT_MINLEN: by default only the top of large 3D factors qualifies) and with the defaults
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05fuzz2
mkdir -p $OUT
export TMPDIR=/tmp
( echo "HIPMF_SPLIT_TASKS=1000000 HIPMF_SPLIT_MINLEN=64 tools/fuzz.py 300 11000 (split dot products forced):"; HIPMF_SPLIT_TASKS=1000000 HIPMF_SPLIT_MINLEN=64 timeout 900 python tools/fuzz.py 300 11000 2>&1 | tail -2
  echo "HIPMF_SPLIT_TASKS=1000000 HIPMF_SPLIT_MINLEN=64 tools/fuzz_big.py 40 1300:"; HIPMF_SPLIT_TASKS=1000000 HIPMF_SPLIT_MINLEN=64 timeout 900 python tools/fuzz_big.py 40 1300 2>&1 | tail -2
  echo "tools/fuzz.py 300 12000 (defaults):"; timeout 900 python tools/fuzz.py 300 12000 2>&1 | tail -2
  echo "tools/fuzz_big.py 40 1400 (defaults):"; timeout 900 python tools/fuzz_big.py 40 1400 2>&1 | tail -2
  echo "tools/soak.py (two host threads, one handle each):"; timeout 600 python tools/soak.py 2>&1 | tail -3
) > $OUT/fuzz.txt 2>&1
cat $OUT/fuzz.txt
