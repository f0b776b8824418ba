cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03pmc
timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > gpurun_out/r03pmc/pmc_hbm_c.txt 2>&1
grep -i "wt_\|fused\|small_factor" gpurun_out/r03pmc/pmc_hbm_c.txt
