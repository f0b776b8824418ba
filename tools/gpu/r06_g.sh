cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06g
mkdir -p $OUT
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) > $OUT/config5_radau5_brusselator_513.txt 2>&1
tail -12 $OUT/config5_radau5_brusselator_513.txt
