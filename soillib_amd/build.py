"""Builds soillib_amd/lib/libsoil_hip.so (the C-ABI library) with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so is built in-tree so that it
travels with the repository snapshot to the GPU box.  Run directly
(`python -m soillib_amd.build`) or through __graft_entry__.build().
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libsoil_hip.so")

SOURCES = ["runtime.hip", "erosion_cells.hip", "erosion_particles.hip", "erosion_particles_tiled.hip", "erosion_step.hip", "slab_runner.hip", "graph.hip",
           "stencil.hip", "path.hip", "noise.hip", "io_tiff.hip", "conditioning.hip"]

# -ffp-contract=off / no fast-math: the numerical contract
# needs every fp32 operation evaluated as written.  -munsafe-fp-atomics selects
# the hardware global_atomic_add_f32 instead of a CAS loop.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-munsafe-fp-atomics", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsoil_hip.so cannot be built")


def _deps_digest():
    h = hashlib.sha256()
    paths = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    paths.append(os.path.join(HERE, "..", "include", "soil_hip.h"))
    paths.append(os.path.join(HERE, "..", "include", "soil_slab.h"))
    for p in paths:
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False, variant=None, extra_flags=()):
    """variant/extra_flags: a diagnostics build next to the product one, e.g.
    build(variant="prof", extra_flags=["-DSOIL_PROF"]) -> lib/libsoil_hip_prof.so (load it
    with SOIL_LIB=...; tools/prof_round.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    suffix = "_" + variant if variant else ""
    LIB = os.path.join(LIBDIR, "libsoil_hip%s.so" % suffix)
    stamp = os.path.join(LIBDIR, "libsoil_hip%s.digest" % suffix)
    digest = _deps_digest() + " ".join(extra_flags)
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        if open(stamp).read().strip() == digest:
            return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", suffix + ".o"))
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
