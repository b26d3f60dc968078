#!/bin/bash
# pin_with_libosmocore.sh -- turn "parity: partial" into a pinned oracle on a machine that HAS libosmocore.
#
# The build container of this repository has no libosmocore (headers, library, network), so the reference files that
# include <osmocom/core/*.h> cannot be built there and five rows of the oracle are pinned by restatement + properties
# only: I block (de)interleaver, U puncturers, the convolutional encoders, V osmo_conv_decode() on noisy input (the call
# chain is lower_mac/viterbi_cch.c:58-66 -> libosmocore), S tetra_burst_sync_in(), L tp_sap_udata_ind().  This script
# builds the reference's OWN sources against a real libosmocore and records what they compute:
#
#   1. oracle/_ref_osmo/libtetra_ref_osmo.so   the reference's lower_mac/{tetra_interleave,tetra_conv_enc,tetra_scramb,
#        crc_simple,tetra_rm3014,viterbi,viterbi_cch,viterbi_tch}.c, compiled where they lie, linked with -losmocore
#   2. oracle/_ref_osmo/tetra-rx               the reference's receiver, built by the reference's own Makefile
#  2b. oracle/_ref_osmo/pin_harness            tools/pin_harness.c linked against the reference's own archives: tp_sap_udata_ind() on single
#        blocks of every type (SCH/HU: no downlink burst carries one) and tetra_gsmtap_makemsg()'s bytes (the GSMTAP constants)
#   3. tests/golden/ref_vectors_osmo.json      tests/golden/make_golden_osmo.py: interleaver and puncturer tables, encoder
#        outputs, osmo_conv_decode() on noisy blocks of all six block types (BER 2 / 5 / 8 %, erasures) and on the speech
#        code, tetra-rx's stdout / stderr for the stream list of SURVEY.md 8(c)
#   4. python -m pytest tests/test_oracle_golden.py -k osmo      the oracle against those vectors
#
# Nothing of the reference is copied into the repository: the vectors are inputs and outputs only.  Commit
# tests/golden/ref_vectors_osmo.json; tests/test_oracle_golden.py::test_osmo_* consume it when it is present (they are
# skipped when it is not, which is the state of this repository: THE SCRIPT HAS NOT BEEN RUN -- it cannot be, here).
#
#   usage: tools/pin_with_libosmocore.sh [REFERENCE_CHECKOUT]        (default /root/reference; needs pkg-config libosmocore)
set -euo pipefail
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_ref_osmo
pkg-config --exists libosmocore || { echo "libosmocore not found by pkg-config: this recipe needs it (src/Makefile:1-2 of the reference does too)" >&2; exit 2; }
[ -d "$REF/src/lower_mac" ] || { echo "no reference checkout at $REF" >&2; exit 2; }
mkdir -p "$OUT"
S=$REF/src
echo "libosmocore $(pkg-config --modversion libosmocore)" | tee "$OUT/versions.txt"
(cd "$REF" && git rev-parse HEAD 2>/dev/null || echo unknown) | sed 's/^/osmo-tetra /' | tee -a "$OUT/versions.txt"
# 1. the block-level functions, from the reference's own sources
gcc -O2 -fPIC -shared -I"$S" $(pkg-config --cflags libosmocore) \
    "$S"/lower_mac/tetra_interleave.c "$S"/lower_mac/tetra_conv_enc.c "$S"/lower_mac/tetra_scramb.c "$S"/lower_mac/crc_simple.c \
    "$S"/lower_mac/tetra_rm3014.c "$S"/lower_mac/viterbi.c "$S"/lower_mac/viterbi_cch.c "$S"/lower_mac/viterbi_tch.c \
    $(pkg-config --libs libosmocore) -o "$OUT/libtetra_ref_osmo.so"
# 2. the reference's receiver, by its own build system (in a scratch copy of its src/: nothing is written to the checkout)
TMP=$(mktemp -d)
cp -r "$S" "$TMP/src"
make -C "$TMP/src" tetra-rx
cp "$TMP/src/tetra-rx" "$OUT/tetra-rx"
# 2b. single blocks through the reference's lower MAC (SCH/HU included) and its GSMTAP message builder: tools/pin_harness.c in front
#     of the reference's own archives -- its recording upper_mac_prim_recv() is taken, the reference's stays out
gcc -O2 -I"$TMP/src" $(pkg-config --cflags libosmocore) "$ROOT/tools/pin_harness.c" -Wl,--allow-multiple-definition \
    "$TMP/src/libosmo-tetra-phy.a" "$TMP/src/libosmo-tetra-mac.a" "$TMP/src/libosmo-tetra-crypto.a" "$TMP/src/libosmo-tetra-mac.a" \
    $(pkg-config --libs libosmocore) -o "$OUT/pin_harness"
rm -rf "$TMP"
# 3. the vectors   4. the oracle against them
make -s -C "$ROOT/oracle" all
python3 "$ROOT/tests/golden/make_golden_osmo.py" "$OUT"
python3 -m pytest "$ROOT/tests/test_oracle_golden.py" -q -k osmo
echo "done: commit tests/golden/ref_vectors_osmo.json (and quote $OUT/versions.txt in DESIGN.md section 6)"
