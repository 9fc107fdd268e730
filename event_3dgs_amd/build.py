"""Builds libe3dgs_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One object per source (compiled in parallel, rebuilt only when the source or a header changed), then one link."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.path.join(HERE, "libe3dgs_hip.so")
SOURCES = ["capi.hip", "forward.hip", "backward.hip", "scan_sort.hip", "aux.hip", "densify.hip"]
# -ffp-contract=off is part of the arithmetic contract (bit-exact forward vs the oracle):
# only explicit FMA() fuses.  (No float atomics anywhere in the library.)
# -fno-slp-vectorize: the SLP pass pairs scalar fp32 ops into v_pk_* instructions (1.7x the issue cost of one op) and pays
# for it with v_mov shuffles -- a net loss in the compositing kernels (measured -3 % on render_bwd_kernel).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "e3dgs_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    os.makedirs(OBJ, exist_ok=True)
    jobs, objs = [], []
    for src in sources:
        sp, op = os.path.join(CSRC, src), os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            jobs.append([hipcc] + FLAGS + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


def status():
    """What build() would do here, without doing it: (objects that would be recompiled, whether the library would be
    relinked, modification time of the library).  smoke() prints it on the GPU box, so that the driver's records say whether
    the box ran the binaries that travelled with the tree or rebuilt them."""
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "e3dgs_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    stale = []
    for src in sources:
        sp, op = os.path.join(CSRC, src), os.path.join(OBJ, src.replace(".hip", ".o"))
        if not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            stale.append(src)
    have = os.path.exists(OUT)
    return stale, (not have) or bool(stale), (os.path.getmtime(OUT) if have else None)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
