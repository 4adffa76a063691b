"""
Builds tests/emu/build/libpokerrl_emu.so: the package's kernel + C-ABI SOURCES compiled for the host with g++ -DPRL_EMU
against the fiber-based SIMT emulator (prl_emu.h). TEST INFRASTRUCTURE ONLY -- see prl_emu.h. The product never loads it.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pokerrl_amd", "csrc")
OUT_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libpokerrl_emu.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
        [os.path.join(HERE, "prl_emu.h"), os.path.join(HERE, "prl_emu.cpp"), os.path.join(ROOT, "include", "pokerrl_hip.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    if not force and os.path.isfile(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    objs, procs = [], []
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DPRL_EMU", "-I" + CSRC, "-I" + HERE,
             "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]
    for s in srcs + [os.path.join(HERE, "prl_emu.cpp")]:
        o = os.path.join(OUT_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.isfile(o) or os.path.getmtime(o) < newest:
            procs.append(subprocess.Popen(["g++"] + flags + ["-x", "c++", "-c", s, "-o", o]))
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("emu build failed")
    subprocess.check_call(["g++", "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
