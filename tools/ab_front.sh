#!/bin/bash
# A/B of k_front_stream build variants on one box: rebuilds the library with each flag set and prints the bench's
# step time and the kernel's HIP-event duration.  usage: tools/ab_front.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  TGPU_HIPCC_FLAGS="$flags" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" >/dev/null 2>&1
  for i in 1 2; do
    python bench.py --steps 20 --warmup 8 --depth ${DEPTH:-8} --no-cpu-baseline --no-secondary --no-e2e --no-sustained 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d['breakdown_ms'].values())[-1]
print('[$flags]', round(d['value']/1e9,3), round(d['ms_per_step'],4), 'front', round(k['k_front_stream']*1e3,1), 'fix', round(k['k_front_stream_fix']*1e3,1), 'walk', round(k['k_walk']*1e3,1))"
  done
done
