"""Builds instascene_amd/_C_hip.so: the compiled torch extension over libinstascene_hip.so (isr_torch_ext.cpp).  Host code only
(g++ against torch's headers; the kernels live in libinstascene_hip.so, which must be built first).  `python build.py`."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "_C_hip.so")
SRC = os.path.join(HERE, "isr_torch_ext.cpp")


def build(force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce
    deps = [SRC, os.path.join(ROOT, "include", "instascene_rasterizer.h"), os.path.join(PKG, "libinstascene_hip.so")]
    # up to date = newer than its sources AND built against this torch (a stamp beside the object: an extension built against
    # another torch's ABI would fail to import, and the drop-in would quietly serve the Python binding instead)
    stamp = OUT + ".torch_version"
    same_torch = os.path.exists(stamp) and open(stamp).read().strip() == torch.__version__
    if not force and os.path.exists(OUT) and same_torch and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.environ.get("ROCM_PATH", "/opt/rocm") + "/include",
                                os.path.join(ROOT, "include")]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_C_hip", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
    cmd += ["-I" + i for i in inc] + [SRC, "-o", OUT, "-L" + tl, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
                                      "-ltorch_python", "-L" + PKG, "-linstascene_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tl]
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(torch.__version__ + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
