"""Builds libmedplib_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and `python -m medplib_amd.build`.

Each .hip/.cpp under csrc/ is compiled to an object (cached by mtime) and linked into medplib_amd/lib/libmedplib_hip.so.
hipcc cross-compiles for gfx950 without a GPU present."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmedplib_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=off"]


def _hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.run([c, "--version"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            return c
        except Exception:
            continue
    raise RuntimeError("hipcc not found")


def _stale(src, obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def _file_flags(src):
    """Extra hipcc flags a source asks for itself with a `// build-flags: ...` line in its header comment (e.g. the MFMA
    register-form switch of the attention backward, which is a per-translation-unit LLVM option)."""
    flags = []
    with open(src) as f:
        for _, line in zip(range(60), f):
            if line.startswith("// build-flags:"):
                flags += line.split(":", 1)[1].split()
    return flags


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        if force or _stale(src, obj, hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + _file_flags(src) + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return src, r.returncode, r.stdout

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(compile_one, jobs):
            if verbose and (rc != 0 or out.strip()):
                print(f"[build] {os.path.basename(src)} rc={rc}\n{out}", file=sys.stderr)
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc compilation failed")
    objs = [os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    if verbose:
        print(f"[build] {LIB} ({len(jobs)} objects rebuilt)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
