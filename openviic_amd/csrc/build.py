"""Build ``libovc.so`` (the HIP kernels + C ABI of include/ovc.h) for gfx950 with hipcc.

    python -m openviic_amd.csrc.build [--force]
    python -m openviic_amd.csrc.build --hooks      # the measurement build for tools/ (tools/libovc_hooks.so)

hipcc cross-compiles without a GPU.  The library is written next to the sources (in-tree), which
is where ``openviic_amd.native`` loads it from; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "rowops.hip", "attention.hip", "beam.hip", "engine.hip"]
HEADERS = ["common.h", "gemm_split.h", os.path.join("..", "..", "include", "ovc.h")]
LIBRARY = os.path.join(HERE, "libovc.so")
ARCH = "gfx950"


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    return any(os.path.getmtime(os.path.join(HERE, f)) > built for f in SOURCES + HEADERS)


HOOKS_LIBRARY = os.path.join(HERE, "..", "..", "tools", "libovc_hooks.so")


def build(force: bool = False, verbose: bool = True, hooks: bool = False, defines=()) -> str:
    """``hooks=True`` builds the MEASUREMENT library instead (``-DOVC_MEASUREMENT_HOOKS`` -> tools/libovc_hooks.so): the only
    build that reads OVC_DEBUG_* / OVC_KSPLIT_* / the kernel A/B switches from the environment (csrc/common.h).  It is loaded
    explicitly (``OVC_LIBRARY=tools/libovc_hooks.so``), never by default; ``ovc_build_info()`` names it."""
    library = os.path.abspath(HOOKS_LIBRARY) if hooks else LIBRARY
    if not hooks and not defines and not force and not is_stale():
        return library
    hipcc = find_hipcc()
    objects = []
    procs = []
    suffix = ".hooks.o" if hooks else ".o"
    flags = (["-DOVC_MEASUREMENT_HOOKS"] if hooks else []) + ["-D" + d for d in defines]
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", suffix))
        objects.append(obj)
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-Wno-unused-result"] + flags + [
               "-c", os.path.join(HERE, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on {}:\n{}\n".format(src, out))
        elif verbose and out.strip():
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", library] + objects)
    if verbose:
        print("built", library)
    return library


if __name__ == "__main__":
    build(force="--force" in sys.argv, hooks="--hooks" in sys.argv,
          defines=[a[2:] for a in sys.argv[1:] if a.startswith("-D")])
