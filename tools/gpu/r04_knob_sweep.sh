# one-at-a-time sweep of environment knobs against the headline on the final build (value / factor / SpTRSV pair); default first and last
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
run() {
  env "$@" python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-28s value %.3f ms  factor %.3f  sptrsv_pair %.4f  solve %.3f' % ('$*', d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['phases_ms']['solve_total_last']))"
}
{
run X=default
for v in 16 32 48; do run HIPMF_WT_FRONTS=$v; done
for v in 32 96 128; do run HIPMF_WT_KB=$v; done
for v in 24 56; do run HIPMF_UP_TOP_FRONTS=$v; done
for v in 16 48; do run HIPMF_UP_STAGE=$v; done
for v in 32 64; do run HIPMF_UP_STAGE_BWD=$v; done
for v in 0 16; do run HIPMF_UP_STAGE_MID=$v; done
for v in 20 36 44; do run HIPMF_SMALL_SPLIT=$v; done
for v in 3000 12000; do run HIPMF_SMALL_WIDE=$v; done
for v in 128 160; do run HIPMF_MID_LU_MMAX=$v; done
for v in 192 320 384; do run HIPMF_UPD32_MAXF=$v; done
for v in 1024 4096; do run HIPMF_UPD_G4=$v; done
run X=default
} | tee gpurun_out/r04s/knob_sweep.txt
