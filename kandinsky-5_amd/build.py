"""Build libk5.so (HIP kernels + engine + C ABI) for gfx950, in-tree.

    python kandinsky-5_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Output: kandinsky-5_amd/lib/libk5.so (git-ignored, travels
to the GPU box with the repo snapshot)."""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libk5.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-fno-slp-vectorize", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
RESOURCES = os.path.join(OBJDIR, "resources.json")   # per-kernel VGPR / AGPR / scratch / spill report of the last compile


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "k5.h"), __file__]
    return max(os.path.getmtime(p) for p in deps)


def parse_resources(remarks):
    """`remark: <file>:<line>:<col>:  Key: value [-Rpass-analysis=...]` lines -> {demangled-ish kernel name: {key: int}}"""
    out, cur = {}, None
    for ln in remarks:
        body = ln.split("remark:", 1)[1].split("[-Rpass-analysis")[0].strip()
        if ":" not in body:
            continue
        k, v = (t.strip() for t in body.split(":", 1))
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            try:
                cur[k] = int(v)
            except ValueError:
                cur[k] = v
    return out


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest_dep():
        return LIB
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(ROOT, "include", "k5.h")))

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        remarks = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
        rest = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln]
        if rest:
            print("\n".join(rest), file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(obj + ".resources.json", "w") as f:
            json.dump(parse_resources(remarks), f, indent=1)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    merged = {}
    for o in objs:
        try:
            merged.update(json.load(open(o + ".resources.json")))
        except FileNotFoundError:
            pass
    with open(RESOURCES, "w") as f:
        json.dump(merged, f, indent=1)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
