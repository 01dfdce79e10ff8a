"""Build libcareless_hip.so for gfx950 with hipcc (in-tree; the .so travels with the source tree).

    python -m careless_amd.build            # rebuild if any source is newer than the library
    python -m careless_amd.build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcareless_hip.so")
# (source, object stem, extra flags): elbo_mlp.hip is compiled twice -- Dense-only scalers and the per-image-layer variant
UNITS = [("cl_api.hip", "cl_api", []), ("elbo_mlp.hip", "elbo_mlp", ["-DCL_IMGL=0"]), ("elbo_mlp.hip", "elbo_mlp_imgl", ["-DCL_IMGL=1"]),
         ("elbo_mlp.hip", "elbo_mlp_packed", ["-DCL_IMGL=2"]),
         ("elbo_mlp.hip", "elbo_mlp_chain", ["-DCL_CHAIN=1"]),
         ("elbo_mlp.hip", "elbo_mlp_det", ["-DCL_DET=1"]),               # deterministic mode: the epilogue's atomics as stores
         ("elbo_mlp.hip", "elbo_mlp_packed_det", ["-DCL_IMGL=2", "-DCL_DET=1"]),    # ... in the packed layout (single-pass Laue)
         ("elbo_mlp.hip", "elbo_mlp_chain_det", ["-DCL_CHAIN=1", "-DCL_DET=1"]),    # ... for the last block of a layer-block chain
         ("elbo_narrow.hip", "elbo_narrow", ["-fno-slp-vectorize"]),     # (packed fp32 math costs more than it saves beside MFMAs)
         # (4x4x1 results feed vector code: no accumulator-register detour); four parts = four groups of instances, compiled in parallel
         ("elbo_lane.hip", "elbo_lane0", ["-DCL_LANE_PART=0", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
         ("elbo_lane.hip", "elbo_lane1", ["-DCL_LANE_PART=1", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
         ("elbo_lane.hip", "elbo_lane2", ["-DCL_LANE_PART=2", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
         ("elbo_lane.hip", "elbo_lane3", ["-DCL_LANE_PART=3", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
         ("elbo_lane.hip", "elbo_lane4", ["-DCL_LANE_PART=4", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),      # per-image layers (round 5)
         ("elbo_elem.hip", "elbo_elem", []), ("elbo_laue.hip", "elbo_laue", []), ("wide_gemm.hip", "wide_gemm", []),
         ("elbo_peel.hip", "elbo_peel", []),
         # host threads, no device code: the formatter's symmetry bookkeeping (exact products kept apart from their sums)
         ("host_format.cpp", "host_format", ["-ffp-contract=off", "-pthread"])]
SOURCES = sorted({u[0] for u in UNITS})
HEADERS = ["cl_math.h", "cl_kernels.h", os.path.join("..", "..", "include", "careless_hip.h")]
ARCH = "gfx950"


LANE_FLAG = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]      # internal LLVM option (validated on ROCm 7.2.0 / AMD clang 22); probed before use


def _lane_flag_ok(hipcc: str) -> bool:
    """The lane kernel asks LLVM to keep 4x4x1 MFMA results in architectural registers.  The option is internal to the AMDGPU
    backend and may be renamed or dropped by a later ROCm: compile an empty device function with it; without it the kernel still
    builds (its inline-assembly accumulators do not depend on the option), only slower by a few accumulator-register moves."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n__global__ void probe() {}\n")
        r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-O3", "--cuda-device-only"] + LANE_FLAG + ["-c", src, "-o", os.path.join(d, "probe.o")],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return r.returncode == 0


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libcareless_hip.so cannot be built on this machine")
    return exe


def source_hash() -> str:
    """Identity of the kernel sources a library is built from (sha256 over csrc/, include/ and this file, 12 hex digits): profiles
    and `profiles/traffic.json` record it, `bench.py` prints it, so a figure measured on other sources shows as stale."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(HERE, "..", "include", "careless_hip.h")
    for f in sorted(os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith((".hip", ".h", ".cpp"))) + [inc, os.path.abspath(__file__)]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, stamps: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path.
    `stamps=True` builds the diagnostic variant libcareless_hip_stamps.so (in-kernel phase timers, -DCL_STAMPS)."""
    if stamps:
        return _build(os.path.join(LIBDIR, "libcareless_hip_stamps.so"), ["-DCL_STAMPS"], verbose)
    if not force and not needs_build():
        return LIB
    return _build(LIB, [], verbose)


def _build(LIB: str, extra, verbose: bool) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    lane_ok = _lane_flag_ok(hipcc)
    if not lane_ok and verbose:
        print("hipcc rejects " + " ".join(LANE_FLAG) + ": building elbo_lane.hip without it", flush=True)
    for s, stem, flags in UNITS:
        if not lane_ok:
            flags = [f for f in flags if f not in LANE_FLAG]
        o = os.path.join(LIBDIR, stem + ".o" + ("s" if extra else ""))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17"] + list(extra) + flags + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
