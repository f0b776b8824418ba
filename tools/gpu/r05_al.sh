cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05al
for rep in 1 2; do
for v in "" "HIPMF_UPD_SPLIT=300" "HIPMF_UPD_SPLIT=600" "HIPMF_UPD_SPLIT=1000" "HIPMF_UPD_SPLIT=1500" "HIPMF_UPD_SPLIT=2200"; do
echo "== $v" >> gpurun_out/r05al/lu.txt
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['factor'].get('factor_launches'), d['relative_error'])" >> gpurun_out/r05al/lu.txt 2>&1
done
done
cat gpurun_out/r05al/lu.txt
