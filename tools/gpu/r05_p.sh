# blocked solves: a front's vector block assembled once (tasks of their own) from which front size on?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05p
mkdir -p $OUT
for a in 2048 512 256 128; do
echo "== HIPMF_SF_ASM_FRONT=$a" >> $OUT/asm_front.txt
HIPMF_SF_ASM_FRONT=$a timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/asm_front.txt 2>&1
HIPMF_SF_ASM_FRONT=$a timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/asm_front.txt 2>&1
done
cat $OUT/asm_front.txt
