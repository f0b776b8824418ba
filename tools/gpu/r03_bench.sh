cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03final
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r03final/bench.json 2> gpurun_out/r03final/bench.err
tail -c 300 gpurun_out/r03final/bench.json
python -c "import __graft_entry__ as g; g.smoke()"
