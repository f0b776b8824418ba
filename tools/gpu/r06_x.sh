cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06x
mkdir -p $OUT
timeout 1500 python tools/config4_knobs.py 200 "" "HIPMF_BLOCKED_SLABS=0" > $OUT/config4_knobs.txt 2>&1
cat $OUT/config4_knobs.txt
timeout 600 python tools/block_groups.py 3dl 144 64 1 4 2>&1 | cut -c1-200
timeout 600 python tools/block_groups.py 3d 100 64 4 2>&1 | cut -c1-200
HIPMF_BLOCKED_SLABS=1 timeout 600 python tools/fuzz_big.py 20 1300 2>&1 | tail -1
HIPMF_BLOCKED_SLABS=1 timeout 600 python tools/fuzz.py 100 7400 2>&1 | tail -1
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round5_gpu.py tests/test_round6_gpu.py -m gpu -q 2>&1 | tail -3
