cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03mfma
./tools/microbench/mfma_peak > gpurun_out/r03mfma/mfma_peak.txt 2>&1
cat gpurun_out/r03mfma/mfma_peak.txt
