cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ak
for v in "" "HIPMF_FACTOR_GRAPH=1" "HIPMF_SMALL_PAIR=1" "HIPMF_UPD_SPLIT=1000" "HIPMF_FACTOR_CHAIN=1" "HIPMF_SMALL_WIDE=3000" "HIPMF_SMALL_WIDE=12000" ""; do
echo "== $v" >> gpurun_out/r05ak/lu.txt
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['factor'].get('factor_launches'))" >> gpurun_out/r05ak/lu.txt 2>&1
done
cat gpurun_out/r05ak/lu.txt
