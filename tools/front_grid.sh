#!/bin/bash
# k_front_stream's time per waves-per-SIMD build (TG_STREAM_WPE) and grid size (TGPU_FRONT_BLOCKS = workgroups of 4 waves)
# usage: tools/front_grid.sh "<wpe>:<blocks> <blocks> ..." ...
for spec in "$@"; do
  wpe=${spec%%:*}; grids=${spec#*:}
  TGPU_HIPCC_FLAGS="-DTG_STREAM_WPE=$wpe" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" >/dev/null 2>&1
  for b in $grids; do
    TGPU_FRONT_BLOCKS=$b python tools/front_ablate.py "[wpe $wpe blocks $b]" 2>/dev/null
  done
done
