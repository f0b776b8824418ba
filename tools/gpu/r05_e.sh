# round 5, fifth call: threaded pinned staging at the host-pointer boundary
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_reference_api_gpu.py tests/test_complex_twin_gpu.py tests/test_round2_gpu.py -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
for t in 4 0 1 8; do
HIPMF_STAGE_THREADS=$t timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --nrhs 0 --grid3d 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('stage threads $t: value', d['value'], 'host_api', d['host_api'])"
done
