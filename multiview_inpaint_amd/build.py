"""Builds the HIP shared library for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m multiview_inpaint_amd.build [--force]

Output: multiview_inpaint_amd/csrc/libmvi_hip.so — a plain C-ABI library (include/*.h), no torch.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libmvi_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# gfx950 has one unified VGPR/AGPR file: keep the MFMA accumulators of the attention kernel in VGPRs so
# the softmax reads them in place (no v_accvgpr_read/write shuffling between the two MFMA products)
# -fno-honor-nans: scores are finite or -inf (masked tail), never NaN; lets fmaxf become v_max3_f32 without the
# canonicalising v_max x,x the IEEE lowering inserts in front of every MFMA output
_ATTN = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"] + os.environ.get("MVI_ATTN_FLAGS", "").split()
# -fno-slp-vectorize: the SLP vectoriser packs the softmax row sums into v_pk_add_f32, which issue slower beside MFMAs
# than the scalar adds they replace (MI355X_MICROARCH.md, cycle constants)
EXTRA = {"attn_flash.hip": _ATTN, "attn_flash8.hip": _ATTN + ["-fno-slp-vectorize"], "attn_flash8m16.hip": _ATTN + ["-fno-slp-vectorize"], "ff_geglu.hip": _ATTN + ["-fno-slp-vectorize"],
         "linear_n320.hip": _ATTN + ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(HERE, "..", "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime())):
        return obj, False
    subprocess.check_call([HIPCC, *FLAGS, *EXTRA.get(os.path.basename(src), []), "-c", src, "-o", obj])
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB])
        if verbose:
            print(f"[mvi] built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
