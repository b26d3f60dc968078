#!/bin/bash
# k_front_stream's time with parts of it taken out (TGS_ABLATE bits, see tg_k_front.hip) or other -D flags.
# usage: tools/front_ablate.sh "<flags A>" "<flags B>" ...   ("" = the product build)
for flags in "$@"; do
  TGPU_HIPCC_FLAGS="$flags" python -c "import osmo_tetra_amd as T; T.build_library(force=True)" >/dev/null 2>&1
  python tools/experiments/front_ablate.py "[$flags]" 2>/dev/null
done
