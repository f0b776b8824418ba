cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/block_groups.py 2d 1000 256 1 4 > $OUT/block_groups.txt 2>&1
timeout 600 python tools/block_groups.py 3d 100 64 4 >> $OUT/block_groups.txt 2>&1
cat $OUT/block_groups.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "many or round3 or round5 or rccl or parity" 2>&1 | tail -3
