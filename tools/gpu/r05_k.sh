# round 5: wave-subtree kernels at other occupancies (compile-time variants of the library), one call
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for v in default w1 st256 n10st256 w4; do
if [ $v = default ]; then L=""; else L="lib=$GRAFT_REPO_ROOT/russell_amd/lib/variants/lib_$v.so"; fi
echo "variant $v"
SOLVE_VARIANTS_SHORT=1 timeout 200 python tools/solve_variants.py 1000 $L 2>&1 | grep "defaults"
done
done > $OUT/wt_variants.txt 2>&1
cat $OUT/wt_variants.txt
