# round 5, sixth call: E / E' strides on 128-byte lines -- GPU suite, A/B of the solve and of the factorisation, HBM counters
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
SOLVE_VARIANTS_SHORT=1 timeout 300 python tools/solve_variants.py 1000 > $OUT/solve_variants_c2.txt 2>&1
SOLVE_VARIANTS_SHORT=1 timeout 300 python tools/solve_variants.py 1000 HIPMF_ALIGN_PANELS=0 only=command >> $OUT/solve_variants_c2.txt 2>&1
cat $OUT/solve_variants_c2.txt
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'pool_gb', d['factor']['pool_gb'])"; }
( run aligned; HIPMF_ALIGN_PANELS=0 run packed; run aligned; HIPMF_ALIGN_PANELS=0 run packed ) > $OUT/variants.txt 2>&1
cat $OUT/variants.txt
# HBM counters of the solve kernels, separate passes
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_small.json 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > $OUT/pmc_hbm.txt 2>&1
python tools/sptrsv_traffic.py $OUT/pmc_hbm.txt 2 1065545568 715697040 "gpurun_out/r05f/pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes)" $OUT/bench_small.json > $OUT/sptrsv_traffic.json 2>&1
cat $OUT/sptrsv_traffic.json | head -30
