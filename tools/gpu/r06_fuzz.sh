# round 6: the differential fuzzers and the soak test on the final build (block groups, plain band, replaced pivots + Krylov rescue, narrower gate)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06fuzz
mkdir -p $OUT
export TMPDIR=/tmp
( echo "tools/fuzz.py 400 5000 (random matrices against dense LAPACK; many right-hand sides among the kinds):"; timeout 900 python tools/fuzz.py 400 5000 2>&1 | tail -2
  echo "tools/fuzz_big.py 60 600 (tiled path, random schedule knobs, against SuperLU):"; timeout 900 python tools/fuzz_big.py 60 600 2>&1 | tail -2
  echo "tools/fuzz_host.py 150 (host mirror, two factorisations per case):"; timeout 600 python tools/fuzz_host.py 150 2>&1 | tail -1
  echo "tools/fuzz_complex_det.py 200 9000 (complex twin: determinants and solutions against numpy):"; timeout 600 python tools/fuzz_complex_det.py 200 9000 2>&1 | tail -2
  echo "tools/soak.py (two host threads, one handle each):"; timeout 600 python tools/soak.py 2>&1 | tail -3
  echo "HIPMF_TAG_SOLVE=0 tools/fuzz_big.py 20 900 (completion counters):"; HIPMF_TAG_SOLVE=0 timeout 600 python tools/fuzz_big.py 20 900 2>&1 | tail -1
  echo "HIPMF_LEAF_KERNELS=0 tools/fuzz.py 100 7000:"; HIPMF_LEAF_KERNELS=0 timeout 600 python tools/fuzz.py 100 7000 2>&1 | tail -1
  echo "HIPMF_BLOCK_GROUPS=1 HIPMF_PLAIN_BAND=0 tools/fuzz.py 100 7100 (round-5 shape of the blocked solves):"; HIPMF_BLOCK_GROUPS=1 HIPMF_PLAIN_BAND=0 timeout 600 python tools/fuzz.py 100 7100 2>&1 | tail -1
  echo "HIPMF_KRYLOV=0 tools/fuzz.py 100 7200 (no rescue, no probe solve):"; HIPMF_KRYLOV=0 timeout 600 python tools/fuzz.py 100 7200 2>&1 | tail -1
  echo "tools/soak_big.py (three handles: 1M-DOF real, 250k complex, 60^3 in blocks of 16 columns):"; timeout 900 python tools/soak_big.py 2>&1 | tail -4
) > $OUT/fuzz.txt 2>&1
cat $OUT/fuzz.txt
( time timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_random_patterns_gpu.py tests/test_matrix_zoo_gpu.py -m gpu -q ) > $OUT/pytest_extra.txt 2>&1; tail -3 $OUT/pytest_extra.txt

