"""ctypes binding of include/tetra_gpu.h (plain C ABI, no torch types in the signatures)."""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libtetra_gpu.so")
HEADER = os.path.join(ROOT, "include", "tetra_gpu.h")

REC_BYTES = 320
SLOT_BYTES = 510
TRAIN_NORM_1, TRAIN_NORM_2, TRAIN_NORM_3, TRAIN_SYNC, TRAIN_EXT = range(5)
T_SB1, T_SB2, T_NDB, T_BBK, T_SCH_HU, T_SCH_F = range(6)

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
u64p = C.POINTER(C.c_uint64)


class TgpuError(RuntimeError):
    pass


class TdmaTime(C.Structure):
    _fields_ = [("hn", C.c_uint16), ("sn", C.c_uint32), ("tn", C.c_uint32), ("fn", C.c_uint32), ("mn", C.c_uint32)]


class RxState(C.Structure):
    """struct tetra_rx_state, phy/tetra_burst_sync.h:12-20 of the reference"""
    _fields_ = [("state", C.c_int), ("bits_in_buf", C.c_uint), ("bitbuf", C.c_uint8 * 4096),
                ("bitbuf_start_bitnum", C.c_uint), ("next_frame_start_bitnum", C.c_uint),
                ("burst_cb_priv", C.c_void_p)]


class Block(C.Structure):
    _fields_ = [("type", C.c_int), ("blk_num", C.c_int), ("crc_ok", C.c_int), ("crc", C.c_uint16),
                ("scrambling_code", C.c_uint32), ("type1_len", C.c_uint16), ("type1", u8p)]


class SyncInfo(C.Structure):
    _fields_ = [("cc", C.c_uint8), ("tn", C.c_uint8), ("fn", C.c_uint8), ("mn", C.c_uint8),
                ("mcc", C.c_uint16), ("mnc", C.c_uint16), ("scramb_init", C.c_uint32)]


class UnitData(C.Structure):
    _fields_ = [("type", C.c_int), ("blk_num", C.c_int), ("lchan", C.c_int), ("crc_ok", C.c_int),
                ("crc", C.c_uint16), ("scrambling_code", C.c_uint32), ("tdma_time", TdmaTime),
                ("burst_seq", C.c_uint32), ("burst_type", C.c_int), ("type1_len", C.c_uint16),
                ("type1", u8p), ("traffic", C.c_int), ("type4", u8p), ("type4_len", C.c_uint16),
                ("time_str", TdmaTime)]


class SyncSlot(C.Structure):
    _fields_ = [("off", C.c_uint64), ("burst_seq", C.c_uint32), ("tn_adds", C.c_uint32), ("type", C.c_uint8)]


class SyncEventRec(C.Structure):
    _fields_ = [("ev", C.c_int32), ("bitnum", C.c_uint32), ("arg", C.c_uint32)]


class SyncResult(C.Structure):
    _fields_ = [("nslots", C.c_uint32), ("slots", C.POINTER(SyncSlot)), ("nevents", C.c_uint32),
                ("events", C.POINTER(SyncEventRec)), ("final_state", C.c_int), ("tail_tn_adds", C.c_uint32),
                ("burst_seq", C.c_uint32), ("anchor", C.c_uint64),
                ("grid_bits", C.POINTER(C.c_uint32)), ("ngrid", C.c_uint32), ("noffgrid", C.c_uint32),
                ("grid_base", C.c_uint32)]


class CwireInfo(C.Structure):
    _fields_ = [("nchan", C.c_uint32), ("ngrid", C.c_uint32), ("ndelivered", C.c_uint32), ("total_bytes", C.c_uint64)]


class MultiChan(C.Structure):
    _fields_ = [("h_stream", u8p), ("d_off", C.c_uint64), ("len", C.c_uint64), ("scramb_init", C.c_uint32)]


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("scramb_init", C.c_uint32), ("mcc", C.c_uint16), ("mnc", C.c_uint16),
                ("cc", C.c_uint8), ("ber", C.c_double), ("null_pdu_header", C.c_int)]


UNITDATA_CB = C.CFUNCTYPE(C.c_int, C.POINTER(UnitData), C.c_uint, C.c_void_p)
EVENT_CB = C.CFUNCTYPE(None, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p)

_lib = None


def declared_symbols():
    """function names declared in include/tetra_gpu.h"""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:struct\s+\w+|enum\s+\w+|int|void|uint32_t|size_t|char)\s*\*?\s*(\w+)\s*\(",
                       txt, flags=re.M)
    return sorted(set(n for n in names if n.startswith(("tgpu_", "tetra_", "tp_sap_", "get_punctured"))))


def lib():
    """load libtetra_gpu.so; raises if it has not been built (no silent fallback)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TgpuError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # PyTorch is the device-memory / stream plumbing, so the library must run on the HIP runtime
    # torch has loaded (same libamdhip64 SONAME): import torch first, never a second runtime.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.tgpu_strerror.restype = C.c_char_p
    L.tgpu_strerror.argtypes = [C.c_int]
    L.tgpu_engine_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.tgpu_stages_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.tgpu_stages_lengths.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
    L.tgpu_stages_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 6
    L.tgpu_stages_destroy.argtypes = [C.c_void_p]
    L.tgpu_stages_destroy.restype = None
    L.tgpu_device_host_locality.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.tgpu_engine_destroy.argtypes = [C.c_void_p]
    L.tgpu_plan_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.tgpu_plan_destroy.argtypes = [C.c_void_p]
    L.tgpu_plan_load.argtypes = [C.c_void_p, C.c_uint32, u64p, u8p, u32p, C.c_uint32, u32p]
    L.tgpu_rm3014_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint)]
    L.tgpu_plan_set_rm_decode.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_plan_set_fastpath.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_channel_set_rm_decode.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_traffic_block.argtypes = [u8p, C.c_uint, C.POINTER(C.c_int16)]
    L.tgpu_traffic_block.restype = None
    L.tgpu_gsmtap_makemsg.argtypes = [C.POINTER(TdmaTime), C.c_int, C.c_uint8, C.c_uint8, C.c_int8, C.c_uint8, u8p, C.c_uint,
                                      u8p, C.c_size_t]
    L.tgpu_conv_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.tgpu_conv_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.tgpu_conv_destroy.argtypes = [C.c_void_p]
    L.tgpu_conv_destroy.restype = None
    L.get_punctured_rate.argtypes = [C.c_int, u8p, C.c_int, u8p]
    L.tetra_rcpc_depunct.argtypes = [C.c_int, u8p, C.c_int, u8p]
    L.tgpu_channel_burst_rx.argtypes = [C.c_void_p, u8p, C.c_uint, C.c_int, C.c_uint32]
    L.tgpu_plan_load_blocks.argtypes = [C.c_void_p, C.c_uint32, u64p, u8p, u32p]
    L.tgpu_plan_load_slots.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SyncSlot), C.c_uint32]
    L.tgpu_plan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tgpu_plan_final_codes.argtypes = [C.c_void_p, C.c_void_p, u32p]
    L.tgpu_plan_execute_soft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tgpu_plan_set_wire_only.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_plan_set_side_stream.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_plan_execute_float.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.tgpu_float_to_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tgpu_float_to_bits_afc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_float, C.c_float,
                                         C.POINTER(C.c_float), C.c_void_p]
    L.tgpu_plan_set_wire.argtypes = [C.c_void_p, C.c_void_p]
    L.tgpu_wire_unpack.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
    L.tgpu_plan_read_packed.argtypes = [C.c_void_p, u32p]
    L.tgpu_prof_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    L.tgpu_prof_destroy.argtypes = [C.c_void_p]
    L.tgpu_plan_execute_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.tgpu_comm_unique_id.argtypes = [u8p]
    L.tgpu_comm_create.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tgpu_comm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
    L.tgpu_comm_destroy.argtypes = [C.c_void_p]
    L.tgpu_comm_destroy.restype = None
    L.tgpu_comm_gatherv.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]
    L.tgpu_comm_gatherv_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_void_p]
    L.tgpu_cwire_bound.restype = C.c_uint64
    L.tgpu_cwire_bound.argtypes = [C.c_uint32, C.c_uint32]
    L.tgpu_plan_set_cwire.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.tgpu_plan_set_traffic.argtypes = [C.c_void_p] * 5
    L.tgpu_plan_traffic.argtypes = [C.c_void_p] * 7
    L.tgpu_wire_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, u32p, u32p, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p]
    L.tgpu_cwire_pack.restype = C.c_int64
    L.tgpu_cwire_pack.argtypes = [u8p, u32p, C.c_uint32, C.c_uint32, u32p, u32p, u8p, C.c_size_t]
    L.tgpu_cwire_info.argtypes = [u8p, C.c_size_t, C.POINTER(CwireInfo)]
    L.tgpu_cwire_chan.argtypes = [u8p, C.c_size_t, C.c_uint32, u32p, u32p, u32p]
    L.tgpu_cwire_foreach.restype = C.c_int64
    L.tgpu_cwire_foreach.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.tgpu_cwire_expand.argtypes = [u8p, C.c_size_t, u8p, u32p]
    L.tgpu_sync_dev_cwire_bytes.restype = C.c_uint64
    L.tgpu_sync_dev_cwire_bytes.argtypes = [C.c_void_p]
    L.tgpu_sync_dev_cwire_needed.restype = C.c_uint64
    L.tgpu_sync_dev_cwire_needed.argtypes = [C.c_void_p]
    L.tgpu_plan_execute_float_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.tgpu_prof_read.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    L.tgpu_stage_name.restype = C.c_char_p
    L.tgpu_stage_name.argtypes = [C.c_int]
    L.tgpu_record_blocks.argtypes = [u8p, C.POINTER(Block)]
    L.tgpu_record_sync_info.argtypes = [u8p, C.POINTER(SyncInfo)]
    L.tgpu_channel_create.argtypes = [C.c_void_p, C.c_uint32, UNITDATA_CB, EVENT_CB, C.c_void_p, C.POINTER(C.c_void_p)]
    L.tgpu_channel_destroy.argtypes = [C.c_void_p]
    L.tgpu_channel_bind_flags.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_bool), C.POINTER(C.c_bool)]
    L.tgpu_channel_set_traffic.argtypes = [C.c_void_p, C.c_int]
    L.tgpu_channel_set_blk2_stolen.argtypes = [C.c_void_p, C.c_bool]
    L.tgpu_channel_flush.argtypes = [C.c_void_p]
    L.tetra_burst_sync_in.argtypes = [C.POINTER(RxState), u8p, C.c_uint]
    L.tetra_find_train_seq.argtypes = [u8p, C.c_uint, C.c_uint32, C.POINTER(C.c_uint)]
    L.tetra_tdma_time_add_tn.argtypes = [C.POINTER(TdmaTime), C.c_uint32]
    L.tetra_scramb_get_init.restype = C.c_uint32
    L.tetra_scramb_get_init.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8]
    L.tgpu_channel_deliver.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SyncSlot), u8p, u8p]
    L.tgpu_channel_scramb_init.argtypes = [C.c_void_p, u32p]
    L.tgpu_sync_walk.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint64, u32p, u16p, C.c_uint32, C.c_uint32, C.POINTER(SyncResult)]
    L.tgpu_sync_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u32p, u16p, C.c_void_p]
    L.tgpu_sync_multi_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(MultiChan), C.c_void_p, C.c_uint32, C.c_void_p,
                                         C.POINTER(C.c_void_p), C.c_void_p]
    L.tgpu_sync_multi_launch_packed.argtypes = L.tgpu_sync_multi_launch.argtypes
    L.tgpu_pack_bits.restype = C.c_int64
    L.tgpu_pack_bits.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint]
    L.tgpu_sync_multi_collect.argtypes = [C.c_void_p, C.POINTER(SyncResult)]
    L.tgpu_sync_multi_launch_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(MultiChan), C.c_void_p, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    L.tgpu_sync_dev_ngrid.argtypes = [C.c_void_p]
    L.tgpu_sync_dev_ngrid.restype = C.c_uint32
    L.tgpu_sync_dev_fellback.argtypes = [C.c_void_p]
    L.tgpu_sync_dev_fused.argtypes = [C.c_void_p]
    L.tgpu_sync_dev_why.argtypes = [C.c_void_p, C.c_uint32]
    L.tgpu_sync_dev_free.argtypes = [C.c_void_p]
    L.tgpu_sync_dev_free.restype = None
    L.tgpu_sync_walk_emul.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint64, u32p, u16p, u32p, C.c_uint32, C.POINTER(SyncResult),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tgpu_sync_stream.argtypes = [C.c_void_p, u8p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SyncResult), C.c_void_p]
    L.tgpu_sync_stream_grid.argtypes = [C.c_void_p, C.c_void_p, u8p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.POINTER(SyncResult), C.c_void_p]
    L.tgpu_sync_stream_grid_begin.argtypes = [C.c_void_p, C.c_void_p, u8p, C.c_void_p, C.c_uint64, C.c_uint32,
                                              C.POINTER(SyncResult), C.c_void_p]
    L.tgpu_sync_stream_grid_finish.argtypes = [C.c_void_p, C.c_void_p, u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.POINTER(SyncResult), C.c_void_p]
    L.tgpu_sync_result_free.argtypes = [C.POINTER(SyncResult)]
    L.tgpu_sync_multi_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(MultiChan), C.c_void_p, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.c_void_p]
    L.tgpu_sync_multi_finish.argtypes = [C.c_void_p, C.c_uint32, C.c_uint, C.POINTER(SyncResult), C.c_void_p]
    L.tgpu_sync_multi_ngrid.argtypes = [C.c_void_p]
    L.tgpu_sync_multi_ngrid.restype = C.c_uint32
    L.tgpu_sync_multi_free.argtypes = [C.c_void_p]
    L.tgpu_sync_front_prof_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(MultiChan), C.c_void_p, C.c_uint32,
                                             C.c_uint32, C.POINTER(C.c_float), C.c_void_p]
    L.tgpu_acelp_build_map.argtypes = [C.POINTER(u8p), C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_int32)]
    L.tgpu_acelp_set_tables.argtypes = [C.POINTER(u8p), C.POINTER(C.c_uint)]
    L.tetra_acelp_type2_to_codec.argtypes = [u8p, u8p]
    L.tetra_acelp_type2_to_codec.restype = None
    L.tetra_acelp_codec_to_acelp.argtypes = [u8p, u8p]
    L.tetra_acelp_codec_to_acelp.restype = None
    L.tgpu_reorder_create.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    L.tgpu_reorder_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.tgpu_reorder_destroy.argtypes = [C.c_void_p]
    L.tgpu_sync_front_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                                       C.POINTER(C.c_float), C.c_void_p]
    L.tgpu_synth_slots.argtypes = [C.POINTER(SynthCfg), u8p, C.c_size_t, u8p, u8p]
    _lib = L
    return L


OPT_BURST_MAX, OPT_STREAM_EXACT, OPT_WALK_HOST, OPT_WALK_MONO, OPT_FRONT_BLOCKS, OPT_WALK_WIDE, OPT_RING, OPT_SLOT = range(1, 9)


def set_option(opt, value):
    """tgpu_engine_set_option: a process-wide switch of the library (enum tgpu_option); nothing reads the environment"""
    lib().tgpu_engine_set_option.argtypes = [C.c_void_p, C.c_int, C.c_long]
    _chk(lib().tgpu_engine_set_option(None, int(opt), int(value)), "tgpu_engine_set_option")


def get_option(opt):
    lib().tgpu_engine_get_option.argtypes = [C.c_void_p, C.c_int]
    lib().tgpu_engine_get_option.restype = C.c_long
    return int(lib().tgpu_engine_get_option(None, int(opt)))


def _chk(rc, what):
    if rc != 0:
        raise TgpuError(f"{what}: {lib().tgpu_strerror(rc).decode()} ({rc})")


def _np_u8(a):
    return np.ascontiguousarray(a, np.uint8)


class Engine:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(lib().tgpu_engine_create(C.byref(self._h), device), "tgpu_engine_create")
        self.device = device

    def float_to_bits(self, d_in_ptr, n, d_bits_ptr, d_soft_ptr=0, hip_stream=0):
        _chk(lib().tgpu_float_to_bits(self._h, C.c_void_p(d_in_ptr), n, C.c_void_p(d_bits_ptr),
                                      C.c_void_p(d_soft_ptr), C.c_void_p(hip_stream)), "tgpu_float_to_bits")

    def float_to_bits_afc(self, d_in_ptr, n, d_bits_ptr, filter_val=0.0001, filter_goal=0.0, state=0.0, hip_stream=0):
        st = C.c_float(state)
        _chk(lib().tgpu_float_to_bits_afc(self._h, C.c_void_p(d_in_ptr), n, C.c_void_p(d_bits_ptr), C.c_float(filter_val),
                                          C.c_float(filter_goal), C.byref(st), C.c_void_p(hip_stream)),
             "tgpu_float_to_bits_afc")
        return st.value

    def close(self):
        if self._h:
            lib().tgpu_engine_destroy(self._h)
            self._h = C.c_void_p()


class Plan:
    """batch of slots -> records, device resident (tgpu_plan_*)"""

    def __init__(self, engine, max_slots, max_chan=1):
        self.engine = engine
        self._h = C.c_void_p()
        _chk(lib().tgpu_plan_create(engine._h, max_slots, max_chan, C.byref(self._h)), "tgpu_plan_create")
        self.nslots = 0
        self.nchan = 0

    def load(self, slot_off, slot_type, slot_chan=None, chan_code=None):
        off = np.ascontiguousarray(slot_off, np.uint64)
        typ = _np_u8(slot_type)
        n = len(typ)
        chan = np.zeros(n, np.uint32) if slot_chan is None else np.ascontiguousarray(slot_chan, np.uint32)
        nchan = int(chan.max()) + 1 if n else 1
        codes = np.zeros(nchan, np.uint32) if chan_code is None else np.ascontiguousarray(chan_code, np.uint32)
        nchan = max(nchan, len(codes))
        assert len(off) == n and len(chan) == n and len(codes) == nchan
        _chk(lib().tgpu_plan_load(self._h, n, off.ctypes.data_as(u64p), typ.ctypes.data_as(u8p),
                                  chan.ctypes.data_as(u32p), nchan, codes.ctypes.data_as(u32p)), "tgpu_plan_load")
        self.nslots, self.nchan = n, nchan

    def set_rm_decode(self, on=True):
        _chk(lib().tgpu_plan_set_rm_decode(self._h, int(bool(on))), "tgpu_plan_set_rm_decode")

    def set_fastpath(self, on=True):
        _chk(lib().tgpu_plan_set_fastpath(self._h, int(bool(on))), "tgpu_plan_set_fastpath")

    def load_blocks(self, blk_off, blk_type, blk_code):
        """tgpu_plan_load_blocks: type-5 blocks on their own (enum tp_sap_data_type per block, code per block)"""
        off = np.ascontiguousarray(blk_off, np.uint64)
        typ = _np_u8(blk_type)
        code = np.ascontiguousarray(blk_code, np.uint32)
        assert len(off) == len(typ) == len(code)
        _chk(lib().tgpu_plan_load_blocks(self._h, len(typ), off.ctypes.data_as(u64p), typ.ctypes.data_as(u8p),
                                         code.ctypes.data_as(u32p)), "tgpu_plan_load_blocks")
        self.nslots, self.nchan = len(typ), 1

    def load_slots(self, outcome, scramb_init=0):
        """tgpu_plan_load_slots: one channel, slot table of sync_stream()/sync_walk() read in place"""
        sa = outcome["slot_arr"]
        assert sa.dtype == SLOT_DTYPE and sa.flags.c_contiguous
        _chk(lib().tgpu_plan_load_slots(self._h, len(sa), sa.ctypes.data_as(C.POINTER(SyncSlot)), scramb_init),
             "tgpu_plan_load_slots")
        self.nslots, self.nchan = len(sa), 1

    def execute(self, d_stream_ptr, d_rec_ptr, hip_stream=0):
        _chk(lib().tgpu_plan_execute(self._h, C.c_void_p(d_stream_ptr), C.c_void_p(d_rec_ptr),
                                     C.c_void_p(hip_stream)), "tgpu_plan_execute")

    def execute_soft(self, d_soft_ptr, d_rec_ptr, hip_stream=0):
        _chk(lib().tgpu_plan_execute_soft(self._h, C.c_void_p(d_soft_ptr), C.c_void_p(d_rec_ptr),
                                          C.c_void_p(hip_stream)), "tgpu_plan_execute_soft")

    def execute_float(self, d_phi_ptr, nfloats, d_rec_ptr, hip_stream=0):
        """decode straight from the float phase stream (slicer + soft gather fused; see tetra_gpu.h)"""
        _chk(lib().tgpu_plan_execute_float(self._h, C.c_void_p(d_phi_ptr), int(nfloats), C.c_void_p(d_rec_ptr),
                                           C.c_void_p(hip_stream)), "tgpu_plan_execute_float")

    def execute_float_prof(self, d_phi_ptr, nfloats, d_rec_ptr, hip_stream, prof, step):
        _chk(lib().tgpu_plan_execute_float_prof(self._h, C.c_void_p(d_phi_ptr), int(nfloats), C.c_void_p(d_rec_ptr),
                                                C.c_void_p(hip_stream), prof._h, step), "tgpu_plan_execute_float_prof")

    def execute_prof(self, d_stream_ptr, d_rec_ptr, hip_stream, prof, step):
        _chk(lib().tgpu_plan_execute_prof(self._h, C.c_void_p(d_stream_ptr), C.c_void_p(d_rec_ptr),
                                          C.c_void_p(hip_stream), prof._h, step), "tgpu_plan_execute_prof")

    def set_side_stream(self, on=True):
        _chk(lib().tgpu_plan_set_side_stream(self._h, int(on)), "tgpu_plan_set_side_stream")

    def set_wire_only(self, on=True):
        _chk(lib().tgpu_plan_set_wire_only(self._h, int(bool(on))), "tgpu_plan_set_wire_only")

    def set_traffic(self, d_traffic_ptr, d_type4_ptr=0, d_blocks_ptr=0, d_lens_ptr=0):
        """every batch of the plan ends with the traffic stage (tgpu_plan_set_traffic); d_traffic_ptr 0 = off"""
        _chk(lib().tgpu_plan_set_traffic(self._h, C.c_void_p(d_traffic_ptr), C.c_void_p(d_type4_ptr), C.c_void_p(d_blocks_ptr),
                                         C.c_void_p(d_lens_ptr)), "tgpu_plan_set_traffic")

    def traffic(self, d_traffic_ptr, d_rec_ptr, d_type4_ptr, d_blocks_ptr, d_lens_ptr, hip_stream=0):
        """the traffic stage on the batch the plan executed last (tgpu_plan_traffic)"""
        _chk(lib().tgpu_plan_traffic(self._h, C.c_void_p(d_traffic_ptr), C.c_void_p(d_rec_ptr), C.c_void_p(d_type4_ptr),
                                     C.c_void_p(d_blocks_ptr), C.c_void_p(d_lens_ptr), C.c_void_p(hip_stream)), "tgpu_plan_traffic")

    def set_cwire(self, d_cwire_ptr, cap_bytes=0):
        """device-walk batches of this plan leave the compact transport form of their wire records (set_wire() too) here"""
        _chk(lib().tgpu_plan_set_cwire(self._h, C.c_void_p(d_cwire_ptr), cap_bytes), "tgpu_plan_set_cwire")

    def set_wire(self, d_wire_ptr):
        _chk(lib().tgpu_plan_set_wire(self._h, C.c_void_p(d_wire_ptr)), "tgpu_plan_set_wire")

    def read_packed(self):
        out = np.zeros((self.nslots, 20), np.uint32)
        _chk(lib().tgpu_plan_read_packed(self._h, out.ctypes.data_as(u32p)), "tgpu_plan_read_packed")
        return out

    def final_codes(self, d_rec_ptr=0):
        out = np.zeros(self.nchan, np.uint32)
        _chk(lib().tgpu_plan_final_codes(self._h, C.c_void_p(d_rec_ptr), out.ctypes.data_as(u32p)),
             "tgpu_plan_final_codes")
        return out

    def close(self):
        if self._h:
            lib().tgpu_plan_destroy(self._h)
            self._h = C.c_void_p()


NSTAGES = 6


COMM_ID_BYTES = 128


def comm_unique_id():
    """tgpu_comm_unique_id: 128 bytes one rank draws and hands to the others"""
    b = np.zeros(COMM_ID_BYTES, np.uint8)
    _chk(lib().tgpu_comm_unique_id(b.ctypes.data_as(u8p)), "tgpu_comm_unique_id")
    return b


class Comm:
    """the C-ABI gather of wire records to the collecting rank over RCCL (tgpu_comm_*)"""

    def __init__(self, eng, uid, rank, world):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        uid = _np_u8(uid)
        _chk(lib().tgpu_comm_create(eng._h, uid.ctypes.data_as(u8p), rank, world, C.byref(self._h)), "tgpu_comm_create")

    def gatherv(self, d_send_ptr, nbytes, d_recv_ptr, offs, root=0, hip_stream=0):
        """a size per rank: rank r's nbytes[r] bytes arrive at the root's d_recv + offs[r]"""
        nb = (C.c_size_t * len(nbytes))(*[int(x) for x in nbytes])
        of = (C.c_size_t * len(nbytes))(*[int(x) for x in offs]) if offs is not None else None
        _chk(lib().tgpu_comm_gatherv(self._h, C.c_void_p(d_send_ptr), nb, C.c_void_p(d_recv_ptr), of, root, C.c_void_p(hip_stream)),
             "tgpu_comm_gatherv")

    def gatherv_batch(self, d_send_ptrs, nbytes, d_recv_ptr, offs, root=0, hip_stream=0):
        """len(d_send_ptrs) messages in one exchange: nbytes / offs are [message][rank]"""
        m, w = len(d_send_ptrs), self.world
        ptrs = (C.c_void_p * m)(*[int(x) for x in d_send_ptrs])
        nb = (C.c_size_t * (m * w))(*[int(x) for row in nbytes for x in row])
        of = (C.c_size_t * (m * w))(*[int(x) for row in offs for x in row]) if offs is not None else None
        _chk(lib().tgpu_comm_gatherv_batch(self._h, m, ptrs, nb, C.c_void_p(d_recv_ptr), of, root, C.c_void_p(hip_stream)),
             "tgpu_comm_gatherv_batch")

    def gather(self, d_send_ptr, nbytes, d_recv_ptr, root=0, hip_stream=0):
        _chk(lib().tgpu_comm_gather(self._h, C.c_void_p(d_send_ptr), nbytes, C.c_void_p(d_recv_ptr or 0), root,
                                    C.c_void_p(hip_stream)), "tgpu_comm_gather")

    def close(self):
        if self._h:
            lib().tgpu_comm_destroy(self._h)
            self._h = C.c_void_p()


class Prof:
    """per-stage HIP-event timing of Plan.execute_prof() (tgpu_prof_*)"""

    def __init__(self, max_steps):
        self._h = C.c_void_p()
        self.max_steps = max_steps
        _chk(lib().tgpu_prof_create(max_steps, C.byref(self._h)), "tgpu_prof_create")

    def read(self, nsteps):
        ms = np.zeros((nsteps, NSTAGES), np.float32)
        _chk(lib().tgpu_prof_read(self._h, nsteps, ms.ctypes.data_as(C.POINTER(C.c_float))), "tgpu_prof_read")
        return ms

    @staticmethod
    def stage_names():
        return [lib().tgpu_stage_name(i).decode() for i in range(NSTAGES)]

    def close(self):
        if self._h:
            lib().tgpu_prof_destroy(self._h)
            self._h = C.c_void_p()


WIRE_BYTES = 40


def wire_unpack(wire, slot_ids=None, codes=None):
    """(n,40) wire records -> (n,320) full records (tgpu_wire_unpack)"""
    wire = np.ascontiguousarray(wire, np.uint8).reshape(-1, WIRE_BYTES)
    n = len(wire)
    out = np.zeros((n, REC_BYTES), np.uint8)
    L = lib()
    for i in range(n):
        _chk(L.tgpu_wire_unpack(wire[i].ctypes.data_as(u8p), int(slot_ids[i]) if slot_ids is not None else i,
                                int(codes[i]) if codes is not None else 0, out[i].ctypes.data_as(u8p)), "tgpu_wire_unpack")
    return out


def wire_pack(rec):
    """(n,320) full records -> (n,40) wire records (tgpu_wire_pack: the host form of what the trellis kernels write)"""
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES)
    out = np.zeros((len(rec), WIRE_BYTES), np.uint8)
    L = lib()
    for i in range(len(rec)):
        _chk(L.tgpu_wire_pack(rec[i].ctypes.data_as(u8p), out[i].ctypes.data_as(u8p)), "tgpu_wire_pack")
    return out


def parse_records(rec):
    """(n,320) uint8 host array -> dict of numpy views (layout: csrc/tg_layout.h)"""
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1, REC_BYTES)
    return dict(
        type=rec[:, 0], flags=rec[:, 1], crc_ok=rec[:, 2:4],
        crc=rec[:, 4:8].copy().view(np.uint16).reshape(-1, 2),
        code=rec[:, 8:12].copy().view(np.uint32).reshape(-1),
        slot=rec[:, 12:16].copy().view(np.uint32).reshape(-1),
        sbf0=rec[:, 16:20].copy().view(np.uint32).reshape(-1),
        sbf1=rec[:, 20:24].copy().view(np.uint32).reshape(-1),
        sbcode=rec[:, 24:28].copy().view(np.uint32).reshape(-1),
        bbk=rec[:, 32:46], bits1=rec[:, 48:316], bits2=rec[:, 176:300],
    )


def record_blocks(rec_row):
    """one 320-byte record -> list of dicts in tetra_burst_rx_cb() call order (tgpu_record_blocks)"""
    r = _np_u8(rec_row)
    blk = (Block * 3)()
    n = lib().tgpu_record_blocks(r.ctypes.data_as(u8p), blk)
    out = []
    for i in range(n):
        b = blk[i]
        out.append(dict(type=b.type, blk_num=b.blk_num, crc_ok=b.crc_ok, crc=b.crc, scramb=b.scrambling_code,
                        type1=bytes(bytearray(b.type1[: b.type1_len]))))
    return out


def find_train_seq(buf, end, mask):
    buf = _np_u8(buf)
    assert len(buf) >= end + 22
    off = C.c_uint(0)
    rc = lib().tetra_find_train_seq(buf.ctypes.data_as(u8p), end, mask, C.byref(off))
    return rc, off.value


def synth_slots(types, seed=1, scramb_init=0, mcc=262, mnc=42, cc=1, ber=0.0, null_pdu_header=True, want_type1=False):
    types = _np_u8(types)
    n = len(types)
    out = np.zeros((n, SLOT_BYTES), np.uint8)
    t1 = np.zeros((n, 288), np.uint8) if want_type1 else None
    cfg = SynthCfg(seed, scramb_init, mcc, mnc, cc, ber, int(null_pdu_header))
    _chk(lib().tgpu_synth_slots(C.byref(cfg), types.ctypes.data_as(u8p), n, out.ctypes.data_as(u8p),
                                t1.ctypes.data_as(u8p) if want_type1 else None), "tgpu_synth_slots")
    return (out, t1) if want_type1 else out


STREAM_SLACK = 192


SLOT_DTYPE = np.dtype([("off", np.uint64), ("burst_seq", np.uint32), ("tn_adds", np.uint32), ("type", np.uint8)], align=True)
EVENT_DTYPE = np.dtype([("ev", np.int32), ("bitnum", np.uint32), ("arg", np.uint32)], align=True)


class _ResultOwner:
    """frees a tgpu_sync_result when the last numpy view onto it is gone"""

    def __init__(self, res):
        self.res = res

    def __del__(self):
        lib().tgpu_sync_result_free(C.byref(self.res))


class SyncOutcome(dict):
    """result of the stream synchroniser; 'slot_arr' / 'event_arr' are numpy structured arrays
    (SLOT_DTYPE / EVENT_DTYPE) viewing the C result in place (freed with the last view), 'slots' / 'events'
    lazily built python tuples (off, type, burst_seq, tn_adds) / (ev, bitnum, arg)"""

    def __missing__(self, key):
        if key == "slots":
            a = self["slot_arr"]
            v = list(zip(a["off"].tolist(), a["type"].tolist(), a["burst_seq"].tolist(), a["tn_adds"].tolist()))
        elif key == "events":
            a = self["event_arr"]
            v = list(zip(a["ev"].tolist(), a["bitnum"].tolist(), a["arg"].tolist()))
        else:
            raise KeyError(key)
        self[key] = v
        return v


def _sync_result_to_py(res):
    assert SLOT_DTYPE.itemsize == C.sizeof(SyncSlot) and EVENT_DTYPE.itemsize == C.sizeof(SyncEventRec)
    owner = _ResultOwner(res)

    def grab(ptr, n, dt):
        if not n:
            return np.zeros(0, dt)
        raw = (C.c_uint8 * (n * dt.itemsize)).from_address(C.addressof(ptr.contents))
        raw._owner = owner                       # the view keeps the C arrays alive
        return np.frombuffer(raw, dt)
    grid = bool(res.grid_bits)
    sa = grab(res.slots, 0 if grid else res.nslots, SLOT_DTYPE)
    ea = grab(res.events, res.nevents, EVENT_DTYPE)
    out = SyncOutcome(slot_arr=sa, event_arr=ea, final_state=res.final_state, tail_tn_adds=res.tail_tn_adds,
                      burst_seq=res.burst_seq, anchor=res.anchor, nslots=res.nslots, ngrid=res.ngrid,
                      noffgrid=res.noffgrid, grid_base=res.grid_base)
    if grid:
        out["grid_bits"] = grab(res.grid_bits, (res.ngrid + 31) // 32, np.dtype(np.uint32))
    return out


def rm3014_decode(rx30):
    """tgpu_rm3014_decode: (14 data bits, corrected bit errors) for a received 30-bit AACH word (bit 29 first)"""
    d, n = C.c_uint16(0), C.c_uint(0)
    _chk(lib().tgpu_rm3014_decode(int(rx30), C.byref(d), C.byref(n)), "tgpu_rm3014_decode")
    return int(d.value), int(n.value)


GSMTAP_STRIDE = 52


def gsmtap_batch(engine, d_rec_ptr, d_times_ptr, nslots, d_msgs_ptr, d_lens_ptr, d_traffic_ptr=0, hip_stream=0):
    """tgpu_gsmtap_batch: GSMTAP messages of every CRC-OK block of a decoded batch, on the device (3 per slot)"""
    lib().tgpu_gsmtap_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(lib().tgpu_gsmtap_batch(engine._h, C.c_void_p(d_rec_ptr), C.c_void_p(d_times_ptr), C.c_void_p(d_traffic_ptr), nslots,
                                 C.c_void_p(d_msgs_ptr), C.c_void_p(d_lens_ptr), C.c_void_p(hip_stream)), "tgpu_gsmtap_batch")


TDMA_TIME_DTYPE = np.dtype([("hn", np.uint16), ("_p", np.uint16), ("sn", np.uint32), ("tn", np.uint32), ("fn", np.uint32), ("mn", np.uint32)])


def traffic_block(type4):
    """tgpu_traffic_block: the reference's 690-word traffic dump block from descrambled type-4 bits"""
    t = _np_u8(type4)
    out = np.zeros(690, np.int16)
    lib().tgpu_traffic_block(t.ctypes.data_as(u8p), len(t), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


class ConvDecoder:
    """tgpu_conv_*: depuncture + Viterbi for one block shape (any of the reference's puncturers, either
    mother code), batches resident in HBM"""

    def __init__(self, engine, punct, mother_rate, type3_len, type2_len):
        self._h = C.c_void_p()
        self.type3_len, self.type2_len = type3_len, type2_len
        _chk(lib().tgpu_conv_create(engine._h, punct, mother_rate, type3_len, type2_len, C.byref(self._h)),
             "tgpu_conv_create")

    def execute(self, d_type3_ptr, nblocks, d_type2_ptr, hip_stream=0):
        _chk(lib().tgpu_conv_execute(self._h, C.c_void_p(d_type3_ptr), nblocks, C.c_void_p(d_type2_ptr),
                                     C.c_void_p(hip_stream)), "tgpu_conv_execute")

    def close(self):
        if self._h:
            lib().tgpu_conv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _acelp_tables(cls):
    arrs = [np.ascontiguousarray(c, np.uint8) for c in cls]
    ptrs = (u8p * 3)(*[a.ctypes.data_as(u8p) for a in arrs])
    ns = (C.c_uint * 3)(*[len(a) for a in arrs])
    return arrs, ptrs, ns


def acelp_build_map(cls, to_codec):
    """tgpu_acelp_build_map: index map (source position per destination, -1 = none) of one direction of the ACELP
    re-ordering for the caller's class position tables cls = (class0, class1, class2)"""
    arrs, ptrs, ns = _acelp_tables(cls)
    m = np.zeros(2 * sum(len(a) for a in arrs), np.int32)
    rc = lib().tgpu_acelp_build_map(ptrs, ns, int(bool(to_codec)), m.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc < 0:
        _chk(rc, "tgpu_acelp_build_map")
    assert rc == len(m)
    return m


def acelp_set_tables(cls):
    arrs, ptrs, ns = _acelp_tables(cls)
    _chk(lib().tgpu_acelp_set_tables(ptrs, ns), "tgpu_acelp_set_tables")


def acelp_type2_to_codec(bits, out=None):
    """the reference's tetra_acelp_type2_to_codec() (host buffers, tables from acelp_set_tables)"""
    b = _np_u8(bits)
    o = np.zeros(len(b), np.uint8) if out is None else out
    lib().tetra_acelp_type2_to_codec(b.ctypes.data_as(u8p), o.ctypes.data_as(u8p))
    return o


def acelp_codec_to_acelp(bits, out=None):
    b = _np_u8(bits)
    o = np.zeros(len(b), np.uint8) if out is None else out
    lib().tetra_acelp_codec_to_acelp(b.ctypes.data_as(u8p), o.ctypes.data_as(u8p))
    return o


class Reorder:
    """tgpu_reorder_*: a fixed index map applied to batches of blocks resident in HBM"""

    def __init__(self, engine, src_of_dst):
        self._h = C.c_void_p()
        m = np.ascontiguousarray(src_of_dst, np.int32)
        self.nbits = len(m)
        _chk(lib().tgpu_reorder_create(engine._h, m.ctypes.data_as(C.POINTER(C.c_int32)), len(m), C.byref(self._h)),
             "tgpu_reorder_create")

    def execute(self, d_in_ptr, nblocks, d_out_ptr, hip_stream=0):
        _chk(lib().tgpu_reorder_execute(self._h, C.c_void_p(d_in_ptr), nblocks, C.c_void_p(d_out_ptr),
                                        C.c_void_p(hip_stream)), "tgpu_reorder_execute")

    def close(self):
        if self._h:
            lib().tgpu_reorder_destroy(self._h)
            self._h = C.c_void_p()


def get_punctured_rate(pu, mother, n):
    """the reference's puncturer under its own name (host buffers); returns (rc, out)"""
    m = _np_u8(mother)
    out = np.zeros(n, np.uint8)
    rc = lib().get_punctured_rate(pu, m.ctypes.data_as(u8p), n, out.ctypes.data_as(u8p))
    return rc, out


def rcpc_depunct(pu, type3, mother_len, fill=0xFF):
    """tetra_rcpc_depunct under its own name (host buffers); returns (rc, out)"""
    t = _np_u8(type3)
    out = np.full(mother_len, fill, np.uint8)
    rc = lib().tetra_rcpc_depunct(pu, t.ctypes.data_as(u8p), len(t), out.ctypes.data_as(u8p))
    return rc, out


def gsmtap_makemsg(tm, lchan, ts, bits, ss=0, signal_dbm=0, snr=0, out_size=None):
    """tgpu_gsmtap_makemsg: the reference's GSMTAP message (header + MSB-first packed bits) as bytes;
    tm = (hn, sn, tn, fn, mn)"""
    b = _np_u8(bits)
    t = TdmaTime(*tm)
    n = 16 + (len(b) + 7) // 8 if out_size is None else out_size
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib().tgpu_gsmtap_makemsg(C.byref(t), lchan, ts, ss, signal_dbm, snr, b.ctypes.data_as(u8p), len(b),
                                   out.ctypes.data_as(u8p), n)
    _chk(rc if rc < 0 else 0, "tgpu_gsmtap_makemsg")
    return out[:rc].tobytes()


def grid_indices(outcome):
    """grid mode: indices of the delivered grid slots (set bits of grid_bits), ascending"""
    bits = np.unpackbits(outcome["grid_bits"].view(np.uint8), bitorder="little")[:outcome["ngrid"]]
    return np.flatnonzero(bits)


def sync_stream_grid(engine, plan, h_stream, d_stream_ptr, chunk=64, hip_stream=0, burst_events=True, scramb_init=0):
    """tgpu_sync_stream_grid: classification + host walk (bitmap) + device-built plan lists; the plan is loaded
    (slot = grid slot) unless outcome['noffgrid'] != 0"""
    h_stream = _np_u8(h_stream)
    res = SyncResult()
    _chk(lib().tgpu_sync_stream_grid(engine._h, plan._h, h_stream.ctypes.data_as(u8p), C.c_void_p(d_stream_ptr),
                                     len(h_stream), chunk, 0 if burst_events else 1, scramb_init, C.byref(res),
                                     C.c_void_p(hip_stream)), "tgpu_sync_stream_grid")
    out = _sync_result_to_py(res)
    if not out["noffgrid"] and out["ngrid"]:
        plan.nslots, plan.nchan = out["ngrid"], 1
    return out


class MultiSync:
    """tgpu_sync_multi_*: several recorded channels (host copies `streams`, device copies at d_base + d_offs[c]) in one
    grid / one plan batch.  finish() returns one outcome per channel; record index of grid slot i of channel c =
    outcome['grid_base'] + i"""

    def __init__(self, engine, plan, streams, d_base_ptr, d_offs, chunk=64, hip_stream=0, codes=None):
        self.engine, self.plan, self.hip_stream = engine, plan, hip_stream
        self.streams = [_np_u8(x) for x in streams]
        n = len(self.streams)
        self._ch = (MultiChan * n)()
        for c, x in enumerate(self.streams):
            self._ch[c].h_stream = x.ctypes.data_as(u8p)
            self._ch[c].d_off = int(d_offs[c])
            self._ch[c].len = len(x)
            self._ch[c].scramb_init = int(codes[c]) if codes is not None else 0
        self._h = C.c_void_p()
        _chk(lib().tgpu_sync_multi_begin(engine._h, plan._h, n, self._ch, C.c_void_p(d_base_ptr), chunk, C.byref(self._h),
                                         C.c_void_p(hip_stream)), "tgpu_sync_multi_begin")
        self.ngrid = lib().tgpu_sync_multi_ngrid(self._h)

    def finish(self, burst_events=False, nthreads=4):
        n = len(self.streams)
        res = (SyncResult * n)()
        try:
            _chk(lib().tgpu_sync_multi_finish(self._h, 0 if burst_events else 1, nthreads, res, C.c_void_p(self.hip_stream)),
                 "tgpu_sync_multi_finish")
        finally:
            lib().tgpu_sync_multi_free(self._h)
            self._h = C.c_void_p()
        out = []
        for c in range(n):
            r = SyncResult()
            C.memmove(C.byref(r), C.byref(res[c]), C.sizeof(SyncResult))
            out.append(_sync_result_to_py(r))
        if self.ngrid:
            self.plan.nslots, self.plan.nchan = self.ngrid, n
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tgpu_sync_multi_free(self._h)


class MultiSyncDev:
    """tgpu_sync_multi_launch / _collect: the same batch with the synchroniser walks on the device (k_walk) and the
    decode into d_rec enqueued behind them -- the constructor returns without waiting, collect() waits and returns one
    outcome per channel (events, delivered bitmap, counts); .fellback tells whether the host walks had to decide"""

    def __init__(self, engine, plan, streams, d_base_ptr, d_offs, d_rec_ptr, chunk=64, hip_stream=0, codes=None, chans=None, packed=False):
        self.engine, self.plan, self.hip_stream = engine, plan, hip_stream
        if chans is not None:       # (a prepared channel table: the bench builds it once)
            self.streams, self._ch = chans
        else:
            self.streams, self._ch = multi_chan_table(streams, d_offs, codes)
        n = len(self.streams)
        self._h = C.c_void_p()
        # packed=True: d_base_ptr = the streams packed one bit per bit (pack_bits), the offsets count bits
        fn = lib().tgpu_sync_multi_launch_packed if packed else lib().tgpu_sync_multi_launch
        _chk(fn(engine._h, plan._h, n, self._ch, C.c_void_p(d_base_ptr), chunk, C.c_void_p(d_rec_ptr),
                C.byref(self._h), C.c_void_p(hip_stream)), "tgpu_sync_multi_launch")
        self.ngrid = lib().tgpu_sync_dev_ngrid(self._h)
        self.fused = int(lib().tgpu_sync_dev_fused(self._h))       # 1: front end + trellises as one launch (k_slot, OPT_SLOT 2); 2: trellises early, beside the walk (k_slot_e, OPT_SLOT 3)
        self.fellback = False

    def collect_begin(self):
        """the C half of collect(): wait for the batch, take its outcome out of the plan's buffers -- after this the plan is
        free for the next launch, before collect_end() turns the outcome into Python objects"""
        n = len(self.streams)
        self._res = (SyncResult * n)()
        try:
            _chk(lib().tgpu_sync_multi_collect(self._h, self._res), "tgpu_sync_multi_collect")
            self.fellback = bool(lib().tgpu_sync_dev_fellback(self._h))
            self.why = [int(lib().tgpu_sync_dev_why(self._h, c)) for c in range(n)] if self.fellback else [0] * n
            self.cwire_bytes = int(lib().tgpu_sync_dev_cwire_bytes(self._h))
            self.cwire_needed = int(lib().tgpu_sync_dev_cwire_needed(self._h))
        except Exception:
            for c in range(n):          # (the C side hands back nothing half-filled with an error; whatever is there goes)
                lib().tgpu_sync_result_free(C.byref(self._res[c]))
            self._res = None
            raise
        finally:
            lib().tgpu_sync_dev_free(self._h)
            self._h = C.c_void_p()
        if self.ngrid:
            self.plan.nslots, self.plan.nchan = self.ngrid, n

    def collect(self, raw=False):
        self.collect_begin()
        return self.collect_end(raw)

    def collect_end(self, raw=False):
        n = len(self.streams)
        res, self._res = self._res, None
        if raw:         # counts only (the bench's timed loop): no numpy views, the C arrays are released here
            out = [dict(nslots=res[c].nslots, nevents=res[c].nevents, noffgrid=res[c].noffgrid, ngrid=res[c].ngrid,
                        final_state=res[c].final_state, burst_seq=res[c].burst_seq) for c in range(n)]
            for c in range(n):
                lib().tgpu_sync_result_free(C.byref(res[c]))
            return out
        out = []
        for c in range(n):
            r = SyncResult()
            C.memmove(C.byref(r), C.byref(res[c]), C.sizeof(SyncResult))
            out.append(_sync_result_to_py(r))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tgpu_sync_dev_free(self._h)


DEV_STAGES = 8


def sync_multi_launch_prof(engine, plan, chans, d_base_ptr, d_rec_ptr, prof, step, chunk=64, hip_stream=0):
    """tgpu_sync_multi_launch_prof: one device-walk batch, synchronously, HIP events between all stages.  Returns
    {stage name: ms} for the stages in front of the decode; the decode's own stages are in prof[step]"""
    xs, ch = chans
    ms = (C.c_float * DEV_STAGES)()
    _chk(lib().tgpu_sync_multi_launch_prof(engine._h, plan._h, len(xs), ch, C.c_void_p(d_base_ptr), chunk, C.c_void_p(d_rec_ptr),
                                           C.c_void_p(hip_stream), prof._h, step, ms), "tgpu_sync_multi_launch_prof")
    lib().tgpu_sync_dev_stage_name.restype = C.c_char_p
    return {lib().tgpu_sync_dev_stage_name(i).decode(): float(ms[i]) for i in range(DEV_STAGES)}


def pack_bits(stream, out=None, nthreads=1):
    """tgpu_pack_bits: a 1-bit-per-byte host array -> one bit per bit (LSB first); returns (packed uint8 array, number of
    pieces that held a byte other than 0 / 1).  out: a preallocated (pinned) uint8 array of (len + 7) // 8 bytes"""
    x = _np_u8(stream)
    if out is None:
        out = np.zeros((len(x) + 7) // 8, np.uint8)
    bad = lib().tgpu_pack_bits(C.c_void_p(x.ctypes.data), len(x), C.c_void_p(out.ctypes.data), nthreads)
    if bad < 0:
        _chk(int(bad), "tgpu_pack_bits")
    return out, int(bad)


def cwire_bound(ngrid, nchan):
    return int(lib().tgpu_cwire_bound(ngrid, nchan))


def wire_compact(engine, d_wire_ptr, d_bits_ptr, ngrid, gbase, ncls, d_cwire_ptr, cap, d_total_ptr=0, hip_stream=0):
    """tgpu_wire_compact: 40-byte wire records + delivered bitmap (device) -> the compact transport form (device)"""
    gb, nc = np.ascontiguousarray(gbase, np.uint32), np.ascontiguousarray(ncls, np.uint32)
    _chk(lib().tgpu_wire_compact(engine._h, C.c_void_p(d_wire_ptr), C.c_void_p(d_bits_ptr), ngrid, len(gb), gb.ctypes.data_as(u32p),
                                 nc.ctypes.data_as(u32p), C.c_void_p(d_cwire_ptr), cap, C.c_void_p(d_total_ptr), C.c_void_p(hip_stream)),
         "tgpu_wire_compact")


def cwire_pack(wire, grid_bits, ngrid, gbase, ncls):
    """tgpu_cwire_pack (host): the compact form of ngrid 40-byte wire records under a delivered bitmap, as a uint8 array"""
    w, b = _np_u8(wire).reshape(-1), np.ascontiguousarray(grid_bits, np.uint32)
    gb, nc = np.ascontiguousarray(gbase, np.uint32), np.ascontiguousarray(ncls, np.uint32)
    out = np.zeros(cwire_bound(ngrid, len(gb)), np.uint8)
    n = lib().tgpu_cwire_pack(w.ctypes.data_as(u8p), b.ctypes.data_as(u32p), ngrid, len(gb), gb.ctypes.data_as(u32p),
                              nc.ctypes.data_as(u32p), out.ctypes.data_as(u8p), len(out))
    if n < 0:
        _chk(int(n), "tgpu_cwire_pack")
    return out[:n].copy()


def cwire_info(cw):
    """header + per-channel (gbase, ncls, delivered) of a compact buffer (host array)"""
    c = _np_u8(cw)
    inf = CwireInfo()
    _chk(lib().tgpu_cwire_info(c.ctypes.data_as(u8p), len(c), C.byref(inf)), "tgpu_cwire_info")
    chans = []
    for k in range(inf.nchan):
        g, n, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _chk(lib().tgpu_cwire_chan(c.ctypes.data_as(u8p), len(c), k, C.byref(g), C.byref(n), C.byref(d)), "tgpu_cwire_chan")
        chans.append((g.value, n.value, d.value))
    return dict(nchan=inf.nchan, ngrid=inf.ngrid, ndelivered=inf.ndelivered, total_bytes=int(inf.total_bytes), chans=chans)


def cwire_expand(cw):
    """(wire records of the whole grid [ngrid, 40] with 0xff rows for undelivered slots, delivered bitmap) of a compact buffer"""
    c = _np_u8(cw)
    inf = cwire_info(c)
    wire = np.empty((inf["ngrid"], WIRE_BYTES), np.uint8)
    bits = np.zeros((inf["ngrid"] + 31) // 32, np.uint32)
    _chk(lib().tgpu_cwire_expand(c.ctypes.data_as(u8p), len(c), wire.ctypes.data_as(u8p), bits.ctypes.data_as(u32p)), "tgpu_cwire_expand")
    return wire, bits


def cwire_count(cw):
    """tgpu_cwire_foreach with no callback: parses every record, returns the number of delivered bursts"""
    c = _np_u8(cw)
    n = lib().tgpu_cwire_foreach(c.ctypes.data_as(u8p), len(c), None, None)
    if n < 0:
        _chk(int(n), "tgpu_cwire_foreach")
    return int(n)


def wire_foreach_noop(wire, grid_bits, ngrid):
    """tgpu_wire_foreach with the library's no-op callback: number of delivered records handed over (host arrays)"""
    w = _np_u8(wire)
    acc = C.c_uint64(0)
    lib().tgpu_wire_noop_cb.restype = C.c_void_p
    lib().tgpu_wire_foreach.restype = C.c_uint64
    lib().tgpu_wire_foreach.argtypes = [u8p, u32p, C.c_uint32, C.c_void_p, C.c_void_p]
    bits = None if grid_bits is None else np.ascontiguousarray(grid_bits, np.uint32)
    return int(lib().tgpu_wire_foreach(w.ctypes.data_as(u8p), bits.ctypes.data_as(u32p) if bits is not None else None, ngrid,
                                       lib().tgpu_wire_noop_cb(), C.cast(C.byref(acc), C.c_void_p)))


def multi_chan_table(streams, d_offs, codes=None):
    """(host arrays kept alive, struct tgpu_multi_chan[n]) for MultiSync / MultiSyncDev"""
    xs = [_np_u8(x) for x in streams]
    ch = (MultiChan * len(xs))()
    for c, x in enumerate(xs):
        ch[c].h_stream = x.ctypes.data_as(u8p)
        ch[c].d_off = int(d_offs[c])
        ch[c].len = len(x)
        ch[c].scramb_init = int(codes[c]) if codes is not None else 0
    return xs, ch


class Stages:
    """tgpu_stages_*: the lower MAC's steps one by one for a batch of blocks of one tp_sap_data_type (device pointers)"""

    def __init__(self, engine, blk_type):
        self._h = C.c_void_p()
        _chk(lib().tgpu_stages_create(engine._h, int(blk_type), C.byref(self._h)), "tgpu_stages_create")
        v = [C.c_uint32() for _ in range(4)]
        _chk(lib().tgpu_stages_lengths(self._h, *[C.byref(x) for x in v]), "tgpu_stages_lengths")
        self.K, self.mother_len, self.type2_len, self.type1_len = [int(x.value) for x in v]

    def execute(self, d_type5, d_codes, nblocks, d_type4, d_type3, d_type3dp, d_type2, d_crc=0, hip_stream=0):
        _chk(lib().tgpu_stages_execute(self._h, C.c_void_p(d_type5), C.c_void_p(d_codes), nblocks, C.c_void_p(d_type4),
                                       C.c_void_p(d_type3), C.c_void_p(d_type3dp), C.c_void_p(d_type2), C.c_void_p(d_crc),
                                       C.c_void_p(hip_stream)), "tgpu_stages_execute")

    def close(self):
        if self._h:
            lib().tgpu_stages_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_host_locality(device=0):
    """tgpu_device_host_locality: (PCI address, NUMA node or -1, sorted list of that node's CPUs or [])"""
    bdf, cl, node = C.create_string_buffer(16), C.create_string_buffer(512), C.c_int(-1)
    _chk(lib().tgpu_device_host_locality(device, bdf, C.byref(node), cl, 512), "tgpu_device_host_locality")
    cpus = []
    for part in cl.value.decode().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return bdf.value.decode(), int(node.value), sorted(set(cpus))


def sync_front_prof_multi(engine, plan, streams, d_base_ptr, d_offs, chunk=64, nrep=10, hip_stream=0):
    """tgpu_sync_front_prof_multi: mean microseconds of (k_front_stream, k_front_stream_fix) on a multi-channel batch"""
    xs = [_np_u8(x) for x in streams]
    ch = (MultiChan * len(xs))()
    for c, x in enumerate(xs):
        ch[c].h_stream = x.ctypes.data_as(u8p)
        ch[c].d_off = int(d_offs[c])
        ch[c].len = len(x)
    us = (C.c_float * 2)()
    _chk(lib().tgpu_sync_front_prof_multi(engine._h, plan._h, len(xs), ch, C.c_void_p(d_base_ptr), chunk, nrep, us,
                                          C.c_void_p(hip_stream)), "tgpu_sync_front_prof_multi")
    return float(us[0]), float(us[1])


def sync_front_prof(engine, plan, d_stream_ptr, length, anchor, chunk=64, nrep=10, hip_stream=0):
    """tgpu_sync_front_prof: mean microseconds of (k_front_stream, k_front_stream_fix) over nrep runs"""
    us = (C.c_float * 2)()
    _chk(lib().tgpu_sync_front_prof(engine._h, plan._h, C.c_void_p(d_stream_ptr), length, chunk, anchor, nrep, us,
                                    C.c_void_p(hip_stream)), "tgpu_sync_front_prof")
    return float(us[0]), float(us[1])


class GridSync:
    """tgpu_sync_stream_grid in two halves: begin() launches the classification asynchronously, finish() waits,
    walks on the host and loads the plan (returns the outcome)"""

    def __init__(self, engine, plan, h_stream, d_stream_ptr, chunk=64, hip_stream=0):
        self.engine, self.plan, self.chunk, self.hip_stream = engine, plan, chunk, hip_stream
        self.h_stream = _np_u8(h_stream)
        self.res = SyncResult()
        _chk(lib().tgpu_sync_stream_grid_begin(engine._h, plan._h, self.h_stream.ctypes.data_as(u8p), C.c_void_p(d_stream_ptr),
                                               len(self.h_stream), chunk, C.byref(self.res), C.c_void_p(hip_stream)),
             "tgpu_sync_stream_grid_begin")

    def finish(self, burst_events=True, scramb_init=0):
        _chk(lib().tgpu_sync_stream_grid_finish(self.engine._h, self.plan._h, self.h_stream.ctypes.data_as(u8p), len(self.h_stream),
                                                self.chunk, 0 if burst_events else 1, scramb_init, C.byref(self.res),
                                                C.c_void_p(self.hip_stream)), "tgpu_sync_stream_grid_finish")
        out = _sync_result_to_py(self.res)
        if not out["noffgrid"] and out["ngrid"]:
            self.plan.nslots, self.plan.nchan = out["ngrid"], 1
        return out


def cls_plain_bits(cls):
    """numpy statement of k_cls_plain: one bit per classification word, set iff the word alone says 'delivered' """
    v = np.asarray(cls, np.uint32) & 0x03FFFFFF
    ok = (v == (3 | 214 << 8)) | (v == (0 | 244 << 8)) | (v == (1 | 244 << 8))
    pad = np.zeros((-len(ok)) % 32, bool)
    return np.packbits(np.concatenate([ok, pad]).reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)


def sync_walk(stream, chunk=64, anchor=0, cls=None, burst_events=True, ysum=None, grid=False, per_call=False, plain=None):
    """host half of the stream synchroniser (tgpu_sync_walk / tgpu_sync_walk_plain); cls=None: every slot settled on
    the bytes, ysum=None: re-lock searches scan the bytes, plain: k_cls_plain's bitmap (True: computed here)"""
    stream = _np_u8(stream)
    res = SyncResult()
    if cls is not None:
        cls = np.ascontiguousarray(cls, np.uint32)
    if ysum is not None:
        ysum = np.ascontiguousarray(ysum, np.uint16)
        assert cls is not None and len(ysum) == len(cls)
    if plain is True:
        plain = cls_plain_bits(cls)
    flags = (0 if burst_events else 1) | (2 if grid else 0) | (4 if per_call else 0)
    if plain is not None:
        plain = np.ascontiguousarray(plain, np.uint32)
        assert cls is not None and len(plain) * 32 >= len(cls)
        _chk(lib().tgpu_sync_walk_plain(stream.ctypes.data_as(u8p), len(stream), chunk, anchor, cls.ctypes.data_as(u32p),
                                        ysum.ctypes.data_as(u16p) if ysum is not None else None, plain.ctypes.data_as(u32p),
                                        len(cls), flags, C.byref(res)), "tgpu_sync_walk_plain")
        return _sync_result_to_py(res)
    _chk(lib().tgpu_sync_walk(stream.ctypes.data_as(u8p), len(stream), chunk, anchor,
                              cls.ctypes.data_as(u32p) if cls is not None else None,
                              ysum.ctypes.data_as(u16p) if ysum is not None else None,
                              len(cls) if cls is not None else 0, flags, C.byref(res)), "tgpu_sync_walk")
    return _sync_result_to_py(res)


def sync_walk_emul(stream, chunk, anchor, cls, ysum, plain=None):
    """tgpu_sync_walk_emul: the device form of the walk (k_walk's phases over csrc/tg_walk_core.h) run on the host;
    returns (outcome, status, why) -- status 1: the device form would hand this channel to the host walk"""
    stream = _np_u8(stream)
    cls = np.ascontiguousarray(cls, np.uint32)
    ysum = np.ascontiguousarray(ysum, np.uint16)
    plain = np.ascontiguousarray(cls_plain_bits(cls) if plain is None else plain, np.uint32)
    res = SyncResult()
    st, why = C.c_int(0), C.c_int(0)
    _chk(lib().tgpu_sync_walk_emul(stream.ctypes.data_as(u8p), len(stream), chunk, anchor, cls.ctypes.data_as(u32p),
                                   ysum.ctypes.data_as(u16p), plain.ctypes.data_as(u32p), len(cls), C.byref(res),
                                   C.byref(st), C.byref(why)), "tgpu_sync_walk_emul")
    return _sync_result_to_py(res), st.value, why.value


def sync_classify(engine, d_stream_ptr, length, chunk, anchor, nslots, hip_stream=0, with_ysum=False):
    out = np.zeros(nslots, np.uint32)
    ys = np.zeros(nslots, np.uint16) if with_ysum else None
    _chk(lib().tgpu_sync_classify(engine._h, C.c_void_p(d_stream_ptr), length, chunk, anchor, nslots,
                                  out.ctypes.data_as(u32p), ys.ctypes.data_as(u16p) if with_ysum else None,
                                  C.c_void_p(hip_stream)), "tgpu_sync_classify")
    return (out, ys) if with_ysum else out


def sync_stream(engine, h_stream, d_stream_ptr, chunk=64, hip_stream=0, burst_events=True):
    """tgpu_sync_stream: host first lock + GPU classification + host walk"""
    h_stream = _np_u8(h_stream)
    res = SyncResult()
    _chk(lib().tgpu_sync_stream(engine._h, h_stream.ctypes.data_as(u8p), C.c_void_p(d_stream_ptr), len(h_stream),
                                chunk, 0 if burst_events else 1, C.byref(res), C.c_void_p(hip_stream)), "tgpu_sync_stream")
    return _sync_result_to_py(res)


class Channel:
    """tgpu_channel + tetra_burst_sync_in(): host bytes in, upper-MAC style callbacks out"""

    def __init__(self, engine, batch_slots=64, on_unitdata=None, on_event=None):
        self.engine = engine
        self.records, self.events = [], []
        self._user_cb = on_unitdata

        def _cb(udp, offset, priv):
            ud = udp.contents
            if offset in (0, 0xFFFFFFFF):
                d = dict(burst_seq=ud.burst_seq, burst_type=ud.burst_type, type=ud.type, blk_num=ud.blk_num,
                         lchan=ud.lchan, crc_ok=ud.crc_ok, traffic=ud.traffic, crc=ud.crc,
                         scramb=ud.scrambling_code, time=(ud.tdma_time.tn, ud.tdma_time.fn, ud.tdma_time.mn),
                         time_str=(ud.time_str.tn, ud.time_str.fn, ud.time_str.mn))
                if ud.traffic:
                    d["type1"] = b""
                    d["type4"] = bytes(bytearray(ud.type4[: ud.type4_len]))
                else:
                    d["type1"] = bytes(bytearray(ud.type1[: ud.type1_len]))
                self.records.append(d)
            else:
                d = self.records[-1]
            if offset == 0xFFFFFFFF:
                return -1
            if self._user_cb is not None:
                return int(self._user_cb(self, d, offset))
            return -1

        def _ev(ev, bitnum, arg, priv):
            self.events.append((ev, bitnum, arg))
            if on_event is not None:
                on_event(ev, bitnum, arg)

        self._cb, self._ev = UNITDATA_CB(_cb), EVENT_CB(_ev)
        self._h = C.c_void_p()
        _chk(lib().tgpu_channel_create(engine._h, batch_slots, self._cb, self._ev, None, C.byref(self._h)),
             "tgpu_channel_create")
        self.trs = RxState()
        self.trs.burst_cb_priv = self._h

    def set_rm_decode(self, on=True):
        _chk(lib().tgpu_channel_set_rm_decode(self._h, int(bool(on))), "tgpu_channel_set_rm_decode")

    def set_traffic(self, v):
        lib().tgpu_channel_set_traffic(self._h, int(v))

    def set_blk2_stolen(self, v):
        lib().tgpu_channel_set_blk2_stolen(self._h, bool(v))

    def feed(self, stream, chunk=64):
        """like tetra-rx.c:82-95: read 64 bytes, tetra_burst_sync_in()"""
        stream = _np_u8(stream)
        L = lib()
        base = stream.ctypes.data
        for o in range(0, len(stream), chunk):
            n = min(chunk, len(stream) - o)
            L.tetra_burst_sync_in(C.byref(self.trs), C.cast(base + o, u8p), n)

    def flush(self):
        _chk(lib().tgpu_channel_flush(self._h), "tgpu_channel_flush")

    def scramb_init(self):
        v = C.c_uint32(0)
        _chk(lib().tgpu_channel_scramb_init(self._h, C.byref(v)), "tgpu_channel_scramb_init")
        return v.value

    def burst_rx(self, burst, train_type, tn_steps=1):
        """the tetra_burst_rx_cb() seam: one 510-bit burst + its training-sequence type"""
        b = _np_u8(burst)
        _chk(lib().tgpu_channel_burst_rx(self._h, b.ctypes.data_as(u8p), len(b), int(train_type), int(tn_steps)),
             "tgpu_channel_burst_rx")

    def deliver(self, slots, h_stream, h_rec):
        """slots: list of (off, type, burst_seq, tn_adds) from sync_stream(); h_rec: (n,320) host records"""
        if isinstance(slots, np.ndarray) and slots.dtype == SLOT_DTYPE:
            a = np.ascontiguousarray(slots)
        else:
            a = np.zeros(len(slots), SLOT_DTYPE)
            for i, (off, t, seq, tn) in enumerate(slots):
                a[i] = (off, seq, tn, t)
        h_stream, h_rec = _np_u8(h_stream), _np_u8(h_rec)
        _chk(lib().tgpu_channel_deliver(self._h, len(a), a.ctypes.data_as(C.POINTER(SyncSlot)),
                                        h_stream.ctypes.data_as(u8p), h_rec.ctypes.data_as(u8p)), "tgpu_channel_deliver")

    def close(self):
        if self._h:
            lib().tgpu_channel_destroy(self._h)
            self._h = C.c_void_p()
