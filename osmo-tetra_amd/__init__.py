"""osmo_tetra_amd -- MI355X-native TETRA lower-MAC receive path.

The product is libtetra_gpu.so (HIP kernels for gfx950 + C host code, C ABI in
include/tetra_gpu.h).  This package is the thin Python plumbing around it: it builds /
loads the library and wraps the C ABI with ctypes; PyTorch is used only for device
memory, streams and torch.distributed.  There is no CPU fallback: every decode entry point
needs the HIP library and a GPU, and fails loudly otherwise.
"""
from .binding import (  # noqa: F401
    LIB_PATH, REC_BYTES, SLOT_BYTES, TRAIN_NORM_1, TRAIN_NORM_2, TRAIN_SYNC,
    T_SB1, T_SB2, T_NDB, T_BBK, T_SCH_HU, T_SCH_F,
    Channel, Comm, comm_unique_id, COMM_ID_BYTES, Engine, Plan, Prof, TgpuError, UnitData, RxState, NSTAGES,
    find_train_seq, traffic_block, gsmtap_makemsg, gsmtap_batch, GSMTAP_STRIDE, TDMA_TIME_DTYPE, ConvDecoder, get_punctured_rate, rcpc_depunct, rm3014_decode, sync_walk, sync_walk_emul, sync_classify, sync_stream, sync_stream_grid, GridSync, MultiSync, MultiSyncDev, multi_chan_table, sync_multi_launch_prof, wire_foreach_noop, DEV_STAGES, sync_front_prof, sync_front_prof_multi, device_host_locality, Stages, grid_indices, acelp_build_map, acelp_set_tables, acelp_type2_to_codec, acelp_codec_to_acelp, Reorder, STREAM_SLACK, lib, parse_records, wire_unpack, wire_pack, WIRE_BYTES, set_option, get_option, OPT_BURST_MAX, OPT_STREAM_EXACT, OPT_WALK_HOST, OPT_WALK_MONO, OPT_FRONT_BLOCKS, OPT_WALK_WIDE, OPT_RING, OPT_SLOT, pack_bits, cwire_bound, wire_compact, cwire_pack, cwire_info, cwire_expand, cwire_count, record_blocks, synth_slots, declared_symbols,
)
from .build import build as build_library  # noqa: F401
