"""bench.py itself on the GPU box: the multi-rank path nobody can rehearse on RCCL with one GPU is run with two ranks
over gloo sharing that GPU (same code: single-GPU reference, decode-only phase, gathered phase with every step's wire
records collected on rank 0, the oracle check of rank 0's records AND of what arrived in its sink), and with one rank
through the library's RCCL gather (--force-gather)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    """a rendezvous port nobody listens on right now (fixed ports collide with a run that has not let go of its socket yet)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def _line(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("form", ["compact", "grid", "compact-every-2", "compact-selflaunch"])
def test_bench_two_ranks_over_gloo_on_one_gpu(form):
    """(compact-selflaunch: plain `python bench.py --gpus 2` -- no torch.distributed.run around it: bench.py starts its own ranks)"""
    every = ["--gather-every", "2"] if form.endswith("every-2") else []
    port = _free_port()
    launcher = [] if form.endswith("selflaunch") else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                                       "127.0.0.1", "--master-port", port]
    form = form.split("-")[0]
    d = _line([sys.executable] + launcher + ["bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "4",
               "--warmup", "2", "--windows", "2", "--bursts", "64000", "--wire-form", form] + every)
    assert d["per_gpu_efficiency"]["scaling_claim"] == "gathered" and 0 < d["per_gpu_efficiency"]["gathered_link_bound"] <= 1
    assert d["gathered"]["steps_per_exchange"] == (2 if every and form == "compact" else 1)
    assert d["gathered"]["wire_form"].startswith(form)
    per_rank = d["gathered"]["bursts_delivered_per_step"] / 2
    if form == "compact":       # delivered bursts only, ~33.5 bytes each (+ bitmap and tables), against 40 per grid slot
        assert d["gathered"]["bytes_per_rank_and_step"] < 35.5 * per_rank and len(d["gathered"]["bytes_per_rank_and_step_all_ranks"]) == 2
    else:
        assert d["gathered"]["bytes_per_rank_and_step"] >= 40 * 64000
    assert d["n_gpus"] == 2 and d["metric"] == "decoded bursts/s" and d["scaling"] == "weak" and d["steps"] == 4
    for k in ("decode_only", "gathered", "single_gpu_reference", "per_gpu_efficiency", "roofline", "timing"):
        assert k in d, k
    assert "error" not in d["gathered"] and d["gathered"]["value"] > 0
    # N > 1: the headline is BASELINE configs[3] as written (the gathered rate); the replica number stands beside it, and the line says which is which
    assert d["value_is"] == "gathered" and d["value"] == d["gathered"]["value"] > 0 and d["ms_per_step"] == d["gathered"]["ms_per_step"]
    assert d["decode_only"]["value"] > 0 and "configs[3]" in d["config"]["workload"] and "replicas" in d["config"]["workload"]
    assert set(d["which_figure_answers_which_config"]) == {"gathered", "decode_only"}
    assert d["ranks_seen"]["world_size"] == 2 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert d["gathered"]["bursts_delivered_per_step"] == d["decode_only"]["bursts_delivered_per_step"] > 2 * 0.9 * 64000 * 0.9
    assert "equal the oracle's" in d["config"]["check"] and "collecting rank" in d["config"]["check"]
    assert 0 < d["per_gpu_efficiency"]["decode_only"] < 1.5
    assert len(d["timing"]["windows_ms_per_step"]) == 2


@pytest.mark.parametrize("form", ["compact", "grid", "compact-every-3"])
def test_bench_single_rank_with_the_rccl_gather(form):
    """(compact-every-3: three steps' blocks in ONE exchange -- tgpu_comm_gatherv_batch, a grouped send / receive of three messages)"""
    every = ["--gather-every", "3"] if form.endswith("every-3") else []
    form = form.split("-")[0]
    d = _line([sys.executable, "bench.py", "--force-gather", "--steps", "4", "--warmup", "2", "--windows", "2", "--bursts", "64000",
               "--no-secondary", "--no-e2e", "--no-sustained", "--wire-form", form] + every)
    assert d["gathered"]["steps_per_exchange"] == (3 if every else 1)
    assert 0 < d["gathered"]["link_bound"]["max_per_gpu_efficiency"] <= 1
    assert d["n_gpus"] == 1 and ("tgpu_comm_gatherv" if form == "compact" else "tgpu_comm_gather (") in d["gathered"]["exchange"]
    assert d["gathered"]["wire_form"].startswith(form)
    assert "collecting rank" in d["config"]["check"] and d["cpu_baseline"]["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
