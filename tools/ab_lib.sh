#!/bin/bash
# A/B of library build variants on one box: rebuilds the library with each flag set and prints the bench's step time and the
# heavy kernels' HIP-event durations.  usage: tools/ab_lib.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  TGPU_HIPCC_FLAGS="$flags" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" >/dev/null 2>&1
  for i in 1 2; do
    python bench.py --steps 20 --warmup 8 --depth ${DEPTH:-8} --no-cpu-baseline --no-secondary --no-e2e --no-sustained 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d['breakdown_ms'].values())[-1]
print('[$flags]', round(d['value']/1e9,3), round(d['ms_per_step'],4), ' '.join('%s %.1f' % (n.replace('k_',''), k[n]*1e3) for n in ('k_front_stream','k_vit<SB1>','k_vit<216>','k_vit<432>')), 'one batch %.3f' % d['timing']['one_batch_at_a_time']['ms_per_batch_median'])"
  done
done
