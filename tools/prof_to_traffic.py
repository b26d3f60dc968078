#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 summaries of tools/prof_run.sh:
   python tools/prof_to_traffic.py gpurun_out/prof_<mix tag> [gpurun_out/prof_<config2 tag> [gpurun_out/prof_<config5 tag>]]
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the x2 is the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md); VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"k_vit<0, 1>": "k_vit<SB1>", "k_vit<1, 1>": "k_vit<216>", "k_vit<2, 1>": "k_vit<432>",
         "k_vit<0, 2>": "k_vit_soft<SB1>", "k_vit<1, 2>": "k_vit_soft<216>", "k_vit<2, 2>": "k_vit_soft<432>",
         "k_front_soft<true>": "k_front_soft<float>", "k_front_soft<false>": "k_front_soft<int8>",
         "k_front_stream<false>": "k_front_stream", "k_front_stream_fix<false>": "k_front_stream_fix", "k_front_stream_fix<false, 640>": "k_front_stream_fix",      # (<true>: packed ingest)
         "k_walk_nodes<false>": "k_walk_nodes", "k_slot_t<1>": "k_slot_t", "k_slot_t<0>": "k_slot_t", "k_slot_t": "k_slot_t", "k_slot<false>": "k_slot"}


def read(path):
    vals = {}
    for line in open(os.path.join(path, "summary.md")):
        m = re.match(r"\| (.+?) \| (\w+) \| ([0-9.e+]+) \| (\d+) \|", line)
        if m:
            vals.setdefault(NAMES.get(m.group(1), m.group(1)), {})[m.group(2)] = float(m.group(3))
        t = re.match(r"\| (.+?) \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \| ([0-9.]+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if t:       # kernel trace row: calls, total, avg, min, max, %
            vals.setdefault(NAMES.get(t.group(1), t.group(1)), {})["duration_us"] = float(t.group(4))
    return vals


def insts(vals):
    """wave-level vector instructions per launch (SQ_INSTS_VALU) of every kernel that issues a million or more"""
    return {k: int(v["SQ_INSTS_VALU"]) for k, v in vals.items() if v.get("SQ_INSTS_VALU", 0) >= 1e6 and k.startswith("k_")}  # (the library's kernels: not torch's fills of the bench's set-up)


def derive(vals):
    traffic, busy = {}, {}
    for k, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v and v["FETCH_SIZE"] > 1000:
            traffic[k] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        if "SQ_ACTIVE_INST_VALU" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            busy[k] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
    return traffic, busy


def main():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(path))
    mix_t, mix_b = derive(read(sys.argv[1]))
    tj["mix"] = {k: mix_t[k] for k in mix_t if k.startswith(("k_front_stream", "k_vit", "k_slot"))}
    tj["mix_valu_busy"] = {k: mix_b[k] for k in mix_b if k.startswith(("k_front_stream", "k_vit", "k_slot"))}
    mv = read(sys.argv[1])
    tj["mix_valu_insts_per_step"] = insts(mv)
    fs = mv.get("k_front_stream", {})
    # shader clock while the kernels run: GRBM_GUI_ACTIVE counts every XCD's cycles
    tj["mix_sclk_ghz"] = round(fs["GRBM_GUI_ACTIVE"] / 8 / (fs["duration_us"] * 1e3), 3) if "GRBM_GUI_ACTIVE" in fs and "duration_us" in fs else 2.3
    tj["_mix_provenance"] = ("round 6: rocprofv3 --pmc passes (tools/prof_run.sh %s mix 1) on `python bench.py --steps 12 --warmup 6 "
                             "--windows 2 --depth 1 --no-cpu-baseline --no-secondary --no-e2e --no-sustained`; summary in profiles/r06_mix_rocprofv3.md"
                             % os.path.basename(sys.argv[1]).replace("prof_", ""))
    if len(sys.argv) > 2:
        c2_t, c2_b = derive(read(sys.argv[2]))
        for k in ("k_front", "k_vit<216>", "k_vit<432>", "k_slot_t"):
            if k in c2_t:
                tj[k] = c2_t[k]
            if k in c2_b:
                tj.setdefault("valu_busy", {})[k] = c2_b[k]
        tj["_provenance"] = ("round 6: the same recipe on `python bench.py --workload config2 ...` (tools/prof_run.sh %s config2 1; "
                             "profiles/r06_config2_rocprofv3.md); bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"
                             % os.path.basename(sys.argv[2]).replace("prof_", ""))
    if len(sys.argv) > 3:
        c5_t, c5_b = derive(read(sys.argv[3]))
        keep = ("k_front_soft", "k_vit_soft", "k_float_to_bits")
        tj["config5"] = {k: c5_t[k] for k in c5_t if k.startswith(keep)}
        tj["config5_valu_busy"] = {k: c5_b[k] for k in c5_b if k.startswith(keep)}
        tj["_config5_provenance"] = ("round 6: the same recipe on `python bench.py --workload config5 ...` (tools/prof_run.sh %s config5 1; "
                                     "profiles/r06_config5_rocprofv3.md)" % os.path.basename(sys.argv[3]).replace("prof_", ""))
    if len(sys.argv) > 4:       # the pipelined configuration: PMC passes of `--depth 8` on rotating captures
        d8 = read(sys.argv[4])
        d8_t, d8_b = derive(d8)
        tj["mix_depth8"] = {k: d8_t[k] for k in d8_t if k.startswith(("k_front_stream", "k_vit", "k_slot"))}
        tj["mix_depth8_fetch_x2_bytes"] = {k: int(2 * v["FETCH_SIZE"] * 1024) for k, v in d8.items() if k.startswith(("k_front_stream", "k_vit", "k_slot")) and "FETCH_SIZE" in v}
        tj["mix_depth8_valu_insts_per_step"] = insts(d8)
        tj["_mix_depth8_provenance"] = ("round 6: the same PMC passes on the PIPELINED configuration (tools/prof_run.sh %s mix 8): `python bench.py --steps 12 "
                                        "--warmup 6 --windows 2 --depth 8 --no-cpu-baseline --no-secondary --no-e2e --no-sustained` -- 8 steps in flight on 8 "
                                        "distinct captures (a counter pass serialises the kernels; what it shows is each launch's traffic on input that no "
                                        "other launch has touched for 8 steps); summary in profiles/r06_mix_depth8_rocprofv3.md"
                                        % os.path.basename(sys.argv[4]).replace("prof_", ""))
    json.dump(tj, open(path, "w"), indent=1)
    print(json.dumps({k: tj[k] for k in ("mix", "mix_valu_busy")}, indent=1))


if __name__ == "__main__":
    main()
