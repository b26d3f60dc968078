"""Build recipe for libtetra_gpu.so (HIP kernels for gfx950 + the C host code).

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the
resulting .so stays in-tree (git-ignored) and travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libtetra_gpu.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

HIP_SRCS = ["tg_k_front.hip", "tg_k_trellis.hip", "tg_k_slot.hip", "tg_k_walk.hip", "tg_k_aux.hip", "tg_cwire.hip", "tg_traffic.hip"]
C_SRCS = ["tg_host.c", "tg_sync.c", "tg_stream.c", "tg_synth.c", "tg_rm.c", "tg_conv.c", "tg_gsmtap.c", "tg_reorder.c", "tg_comm.c", "tg_stages.c", "tg_cwire.c", "tg_pack.c"]
HEADERS = ["tg_layout.h", "vit_core.h", "slot_core.h", "tg_internal.h", "tg_dev.h", "tg_dev_vit.h", "tg_conv.h", "tg_cwire.h", "tg_walk_core.h", os.path.join(ROOT, "include", "tetra_gpu.h")]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    srcs = [os.path.join(CSRC, s) for s in HIP_SRCS + C_SRCS]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _newer(srcs + hdrs + [os.path.abspath(__file__)], LIB):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(ROCM, "include")]
    hip_flags = os.environ.get("TGPU_HIPCC_FLAGS", "").split()     # experiment builds (-DTG_...=n)
    cc_flags = os.environ.get("TGPU_CC_FLAGS", "").split()         # experiment builds of the host code
    # an object is kept when it is newer than its source, every header and this recipe, and was built with the same flags
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(hip_flags) + "|" + " ".join(cc_flags)
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags_now

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    jobs, objs = [], []
    for s in HIP_SRCS + C_SRCS:
        src, o = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        objs.append(o)
        if not force and same_flags and not _newer([src] + hdrs + [os.path.abspath(__file__)], o):
            continue
        if s in HIP_SRCS:
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"] + inc + hip_flags + ["-c", src, "-o", o])
        else:
            jobs.append(["gcc", "-O3", "-std=gnu11", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter"] + inc + cc_flags +
                        ["-c", src, "-o", o])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    open(stamp, "w").write(flags_now)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-lpthread", "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
