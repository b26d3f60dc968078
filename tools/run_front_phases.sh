cd $GRAFT_REPO_ROOT
TGPU_HIPCC_FLAGS="-DTGS_TIMING" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" > gpurun_out/ph_build.log 2>&1
python tools/experiments/front_phases.py > gpurun_out/front_phases.txt 2>&1
cat gpurun_out/front_phases.txt
