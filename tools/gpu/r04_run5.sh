# round 4: graph replay of the factorisation levels and k_front, A/B in one call (boxes differ by up to 1.6x)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04e
mkdir -p $OUT
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>$OUT/err_$1.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'launches', d['factor']['factor_launches'], 'relerr %.1e' % d['relative_error'])"; }
HIPMF_FACTOR_GRAPH=0 HIPMF_MID_FRONT=0 run eager_nomid
HIPMF_FACTOR_GRAPH=1 HIPMF_MID_FRONT=0 run graph_nomid
HIPMF_FACTOR_GRAPH=0 HIPMF_MID_FRONT=1 run eager_mid80
HIPMF_FACTOR_GRAPH=1 HIPMF_MID_FRONT=1 run graph_mid80
HIPMF_FACTOR_GRAPH=1 HIPMF_MID_FRONT=1 HIPMF_MID_MMAX=128 run graph_mid128
HIPMF_FACTOR_GRAPH=0 HIPMF_MID_FRONT=0 run eager_nomid_again
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
awk '/gap/ {g=$(NF-3); if (g>0) s+=g} END {print "sum of positive gaps (us):", s}' $OUT/factor_sequence.txt
grep span $OUT/factor_sequence.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round2_gpu.py tests/test_reference_api_gpu.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
