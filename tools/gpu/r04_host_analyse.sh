# host-only: phases of the symbolic analysis on the GPU box's cores (tools/host/analyse_phases.cpp)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h build
g++ -O2 -std=c++17 -pthread tools/host/analyse_phases.cpp russell_amd/csrc/symbolic.cpp -o build/analyse_phases || exit 1
nproc
for t in ${THREADS:-0}; do
echo "HIPMF_ND_THREADS=$t"
HIPMF_ND_THREADS=$t ./build/analyse_phases 2 1000 0
HIPMF_ND_THREADS=$t ./build/analyse_phases 3 200 1
done 2>&1 | tee gpurun_out/r04h/analyse_phases_${TAG:-base}.txt
