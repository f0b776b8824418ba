# FP64 matrix-pipe counters of the headline on the final build (new supernode partition)
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmc_mfma -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
python tools/rocpd_pmc.py $(find /tmp/pmc_mfma -name '*.db' | head -1) > gpurun_out/r04m/pmc_mfma_c2.txt 2>&1
head -14 gpurun_out/r04m/pmc_mfma_c2.txt
