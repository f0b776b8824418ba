# round 6: the all-small band of the blocked solves as one PLAIN launch per level (HIPMF_PLAIN_BAND), off / on in one call
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06f
mkdir -p $OUT
export TMPDIR=/tmp
for pb in 0 1 0 1; do
  echo "== HIPMF_PLAIN_BAND=$pb" >> $OUT/plain_band.txt
  HIPMF_PLAIN_BAND=$pb timeout 600 python tools/block_groups.py 2d 1000 256 4 >> $OUT/plain_band.txt 2>&1
done
for pb in 0 1; do
  echo "== HIPMF_PLAIN_BAND=$pb" >> $OUT/plain_band.txt
  HIPMF_PLAIN_BAND=$pb timeout 600 python tools/block_groups.py 3d 100 64 4 >> $OUT/plain_band.txt 2>&1
  HIPMF_PLAIN_BAND=$pb timeout 600 python tools/block_groups.py 3dl 144 64 4 >> $OUT/plain_band.txt 2>&1
done
cat $OUT/plain_band.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "many or round3 or round5 or rccl or parity or fused or zoo" 2>&1 | tail -3
