# round 4: k_front v3 in the factorisation, by the largest number of off-diagonal rows it takes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
for mm in 0 80 128; do
if [ $mm = 0 ]; then export HIPMF_MID_FRONT=0; else export HIPMF_MID_FRONT=1 HIPMF_MID_MMAX=$mm; fi
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_mm$mm.json 2> $OUT/bench_mm$mm.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_mm$mm.json").read().strip().splitlines()[-1])
print("mmax=$mm value", d["value"], "factor", d["phases_ms"]["factor"], "launches", d["factor"]["factor_launches"], "relerr", d["relative_error"])
PY
done
unset HIPMF_MID_FRONT HIPMF_MID_MMAX
./tools/microbench/front_bench 1 2>&1 | head -3
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
grep -n "k_front" $OUT/factor_sequence.txt | head
tail -16 $OUT/factor_sequence.txt | head -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_matrix_zoo_gpu.py tests/test_random_patterns_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
