#!/bin/bash
# round 6: A/B of the lane-per-slot kernels on one box -- library build variants (hipcc flags) x TGPU_OPT_SLOT settings: the bench's
# step time, the serialised HIP-event durations of the front-end stage (k_slot when fused) and of the trellis stage, one batch at a time.
# usage: tools/experiments/ab_slot.sh "<flags A>|<slot modes>" ...      e.g. "|0 1 2" "-DTGS_FUSED_DEPTH=2|2"
for spec in "$@"; do
  flags="${spec%%|*}"; modes="${spec##*|}"
  TGPU_HIPCC_FLAGS="$flags" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" >/dev/null 2>&1 || { echo "[$flags] build failed"; continue; }
  for m in $modes; do
    for i in 1 2; do
      python bench.py --steps 20 --warmup 8 --depth ${DEPTH:-8} --slot-mode $m --no-cpu-baseline --no-secondary --no-e2e --no-sustained 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=list(d['breakdown_ms'].values())[-1]
print('[$flags] slot=$m', round(d['value']/1e9,3), round(d['ms_per_step'],4), ' '.join('%s %.1f' % (n.replace('k_',''), k[n]*1e3) for n in k if k[n] > 0.02), 'one batch %.3f' % d['timing']['one_batch_at_a_time']['ms_per_batch_median'])"
    done
  done
done
