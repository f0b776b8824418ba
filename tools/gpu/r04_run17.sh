cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
for i in 1 2 3; do
HIPMF_UPD_XCD=0 run xcd_off
HIPMF_UPD_XCD=1 run xcd_on
done 2>&1 | tee gpurun_out/r04l/xcd_ab.txt
for x in 0 1; do echo "3D 100^3 HIPMF_UPD_XCD=$x"; HIPMF_UPD_XCD=$x timeout 600 python tools/run3d.py 100 lu 2>&1 | grep -E "rep 1|flops/s"; done 2>&1 | tee -a gpurun_out/r04l/xcd_ab.txt
