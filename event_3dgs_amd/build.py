"""Builds libe3dgs_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libe3dgs_hip.so")
SOURCES = ["capi.hip", "forward.hip", "backward.hip", "scan_sort.hip", "aux.hip"]
# -ffp-contract=off is part of the arithmetic contract (bit-exact forward vs the oracle):
# only explicit FMA() fuses.  -munsafe-fp-atomics selects the hardware global_atomic_add_f32.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared"]


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "e3dgs_hip.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
