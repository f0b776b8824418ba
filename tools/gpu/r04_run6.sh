# round 4: top-level slabs of the triangular solves -- neighbours on one XCD, 16-row slabs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>$OUT/err_$1.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'solve', d['phases_ms']['solve_total_last'], 'relerr %.1e' % d['relative_error'])"; }
HIPMF_UP_PAIR_XCD=0 run rows_in_order
HIPMF_UP_PAIR_XCD=1 run pair_xcd
HIPMF_UP_PAIR_XCD=0 HIPMF_UP_MAX_GROUPS=16 run slabs16
HIPMF_UP_PAIR_XCD=0 run rows_in_order_again
cd /tmp
for v in 0 1; do
rm -rf /tmp/pmc_F$v
HIPMF_UP_PAIR_XCD=$v timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_F$v -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_F$v.log 2>&1
echo "== HIPMF_UP_PAIR_XCD=$v" >> $GRAFT_REPO_ROOT/$OUT/pmc_fetch.txt
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find /tmp/pmc_F$v -name '*.db' | head -1) | grep -i "fused\|wt_" >> $GRAFT_REPO_ROOT/$OUT/pmc_fetch.txt
done
cat $GRAFT_REPO_ROOT/$OUT/pmc_fetch.txt
