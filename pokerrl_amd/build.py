"""
Builds libpokerrl_hip.so IN-TREE (pokerrl_amd/lib/) with hipcc for gfx950. Cross-compiles without a GPU.

    python -m pokerrl_amd.build [--force]

-ffp-contract=off: parity with the reference needs separate multiply / add (no FMA contraction), host and device.
-fhip-fp32-correctly-rounded-divide-sqrt: regret matching divides; NumPy's float32 division is correctly rounded.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpokerrl_hip.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC, "-Wall", "-Wno-unused-function"]
DEVICE = ["--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(INCLUDE, "pokerrl_hip.h")]
    return max(os.path.getmtime(h) for h in hs)


def build_native(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = _newest_header()
    objs, jobs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s + ".o")
        objs.append(obj)
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [HIPCC] + COMMON + DEVICE + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append((s, cmd))
    procs = []
    for s, cmd in jobs:  # compile in parallel
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("---- %s ----\n%s\n" % (s, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    if jobs or not os.path.isfile(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines):
    """Instrumented / experimental copy of the library (never loaded by the package): lib/libpokerrl_hip_<name>.so"""
    out = os.path.join(LIB_DIR, "libpokerrl_hip_%s.so" % name)
    vdir = os.path.join(LIB_DIR, "obj_" + name)
    os.makedirs(vdir, exist_ok=True)
    procs, objs = [], []
    only = os.environ.get("PRL_VARIANT_ONLY", "").split()  # experiments on one kernel file: the other objects are the product's own
    for s in sources():
        if only and s not in only:
            objs.append(os.path.join(OBJ_DIR, s + ".o"))
            continue
        obj = os.path.join(vdir, s + ".o")
        objs.append(obj)
        extra = os.environ.get("PRL_VARIANT_FLAGS", "").split()  # experiments with compiler options, e.g. "-mllvm -amdgpu-sched-strategy=max-ilp"
        cmd = [HIPCC] + COMMON + DEVICE + extra + ["-D" + d for d in defines] + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        procs.append(subprocess.Popen(cmd))
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("hipcc failed")
    subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m pokerrl_amd.build --variant timing PRL_FHP_TIMING
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build_native(force="--force" in sys.argv, verbose="-v" in sys.argv))
