"""Build driver for the native parts (hipcc for gfx950; gcc for the CPU oracle).

Everything is built IN-TREE so that the shared objects travel with a repo snapshot:
    herald_amd/libherald_amd.so        the C-ABI library (include/herald_amd.h)
    oracle/_build/liboracle.so         the CPU restatement (test infrastructure only)
    oracle/_ref/*                      reference-built checkers (only where /root/reference exists)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "herald_amd")
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libherald_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",          # parity: never fuse the reference's separate roundings
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"),
]


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build step failed: %s" % " ".join(cmd[:3]))
    return r.stdout


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libherald_amd.so."""
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "herald_amd.h"))
    objs = []
    procs = []
    for src in hip_sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + out + "\n")
            raise RuntimeError("hipcc failed on %s" % cmd[-3])
        if verbose and out.strip():
            print(out)
    if force or procs or _newer(objs, LIB):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs)
    return LIB


PS_LIB = os.path.join(PKG, "libherald_ps.so")


def build_ps_shim(force=False):
    """libherald_ps.so: the libps.so names (include/herald_ps.h) over libherald_amd.so's ha_ps_* engine."""
    src = os.path.join(CSRC, "libps_shim.cpp")
    deps = [src, os.path.join(ROOT, "include", "herald_ps.h"), os.path.join(ROOT, "include", "herald_amd.h"), LIB]
    if force or _newer(deps, PS_LIB):
        _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", src, "-o", PS_LIB, "-L", PKG, "-lherald_amd",
              "-Wl,-rpath,$ORIGIN"])
    return PS_LIB


def build_plugins(force=False):
    """pybind11 modules `hetu_cache` and `laia_cache` (host C++ over the C-ABI) -> herald_amd/plugins/."""
    import sysconfig
    import pybind11
    out_dir = os.path.join(PKG, "plugins")
    os.makedirs(out_dir, exist_ok=True)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    inc = ["-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-I", "/opt/rocm/include"]
    outs = []
    procs = []
    for name in ("hetu_cache", "laia_cache"):
        src = os.path.join(CSRC, "py_%s.cpp" % name)
        out = os.path.join(out_dir, name + suffix)
        outs.append(out)
        if force or _newer([src, os.path.join(ROOT, "include", "herald_amd.h"), LIB], out):
            cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__"] + inc + \
                  [src, "-o", out, "-L", PKG, "-lherald_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
                   "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + o + "\n")
            raise RuntimeError("plugin build failed")
    return outs


def build_oracle(force=False):
    """gcc build of the CPU restatement (oracle/oracle.c)."""
    src = os.path.join(ROOT, "oracle", "oracle.c")
    out_dir = os.path.join(ROOT, "oracle", "_build")
    out = os.path.join(out_dir, "liboracle.so")
    if not os.path.exists(src):
        return None
    os.makedirs(out_dir, exist_ok=True)
    if force or _newer([src], out):
        _run(["gcc", "-O3", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-std=c11",
              "-Wall", "-o", out, src, "-lm"])
    return out


def build_ref(force=False):
    """Reference-built checkers; only possible where /root/reference is mounted."""
    script = os.path.join(ROOT, "oracle", "build_ref.sh")
    if not (os.path.isdir("/root/reference") and os.path.exists(script)):
        return None
    _run(["bash", script])
    return os.path.join(ROOT, "oracle", "_ref")


def build_all(force=False, verbose=False):
    lib = build_lib(force=force, verbose=verbose)
    build_ps_shim(force=force)
    build_plugins(force=force)
    build_oracle(force=force)
    build_ref(force=force)
    return lib


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
