# the new orderings on the device: tests, then circuit-like / grid comparison
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_ordering_amd_gpu.py -m gpu -q -s ) > gpurun_out/r04o/pytest_amd.txt 2>&1; tail -5 gpurun_out/r04o/pytest_amd.txt
timeout 600 python tools/ordering_compare.py circuit 60000 grid 1000 grid3d 60 2>&1 | grep -v "^solver_hipmf" | tee gpurun_out/r04o/ordering_compare.txt
