"""Builds the compiled torch extension `diff_gaussian_rasterization._C_native` in-tree (g++, no GPU needed).

The module only marshals torch tensors to the C ABI of event_3dgs_amd/libe3dgs_hip.so (which it links against and finds
at run time through an $ORIGIN-relative rpath), so it is host code: one g++ invocation, no hipcc."""
import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "ext.cpp")
NAME = "_C_native"
OUT = os.path.join(HERE, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False, verbose=False):
    import torch
    from torch.utils import cpp_extension as ce
    lib = os.path.join(ROOT, "event_3dgs_amd", "libe3dgs_hip.so")
    if not os.path.exists(lib):
        from event_3dgs_amd import build as hip_build
        hip_build.build()
    deps = [SRC, os.path.join(ROOT, "include", "e3dgs_hip.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths(device_type="cuda") + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT,
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
            f"-L{os.path.dirname(lib)}", "-l:libe3dgs_hip.so", "-Wl,-rpath,$ORIGIN/../event_3dgs_amd", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
