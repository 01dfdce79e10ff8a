"""Build libcareless_hip.so for gfx950 with hipcc (in-tree; the .so travels with the source tree).

    python -m careless_amd.build            # rebuild if any source is newer than the library
    python -m careless_amd.build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcareless_hip.so")
# The fused scaler kernels' LeakyReLU is fmaxf(x, leak x); with NaNs honoured hipcc puts a canonicalising v_max_f32 x, x in front of
# every one of them, without it emits the one v_max_f32 -- an instruction it KNOWS, so that its hazard recognizer pads the two wait
# states gfx950 wants in front of an MFMA that reads the result (until round 6 the bare instruction was inline assembly, which it does
# not see: NOTEBOOK R6.1).  Nothing in these units tests for NaN (the non-finite stop is taken on the gradient norm, in elbo_elem.hip).
NNAN = ["-fno-honor-nans"]
# (source, object stem, extra flags): elbo_mlp.hip is compiled twice -- Dense-only scalers and the per-image-layer variant
UNITS = [("cl_api.hip", "cl_api", []), ("elbo_mlp.hip", "elbo_mlp", ["-DCL_IMGL=0"] + NNAN), ("elbo_mlp.hip", "elbo_mlp_imgl", ["-DCL_IMGL=1"] + NNAN),
         ("elbo_mlp.hip", "elbo_mlp_packed", ["-DCL_IMGL=2"] + NNAN),
         ("elbo_mlp.hip", "elbo_mlp_chain", ["-DCL_CHAIN=1"] + NNAN),
         ("elbo_mlp.hip", "elbo_mlp_det", ["-DCL_DET=1"] + NNAN),               # deterministic mode: the epilogue's atomics as stores
         ("elbo_mlp.hip", "elbo_mlp_packed_det", ["-DCL_IMGL=2", "-DCL_DET=1"] + NNAN),    # ... in the packed layout (single-pass Laue)
         ("elbo_mlp.hip", "elbo_mlp_chain_det", ["-DCL_CHAIN=1", "-DCL_DET=1"] + NNAN),    # ... for the last block of a layer-block chain
         ("elbo_narrow.hip", "elbo_narrow", ["-fno-slp-vectorize"] + NNAN),     # (packed fp32 math costs more than it saves beside MFMAs)
         # (4x4x1 results feed vector code: no accumulator-register detour); four parts = four groups of instances, compiled in parallel
         ("elbo_lane.hip", "elbo_lane0", ["-DCL_LANE_PART=0", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),
         ("elbo_lane.hip", "elbo_lane1", ["-DCL_LANE_PART=1", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),
         ("elbo_lane.hip", "elbo_lane2", ["-DCL_LANE_PART=2", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),
         ("elbo_lane.hip", "elbo_lane3", ["-DCL_LANE_PART=3", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),
         ("elbo_lane.hip", "elbo_lane4", ["-DCL_LANE_PART=4", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),      # per-image layers (round 5)
         # three per-image layers on the default depth: WITHOUT the option above (its AGPR-copy rewrite pass crashes on the 23-layer instances)
         ("elbo_lane.hip", "elbo_lane5", ["-DCL_LANE_PART=5"] + NNAN),
         # ... and the widest instances once more per depth below the default (round 6: `--mlp-layers 2 .. 19` at widths 7 .. 10; 16 - 25 s each)
         *[("elbo_lane.hip", f"elbo_lane_d{D}", ["-DCL_LANE_PART=7", f"-DCL_LANE_NL={D}", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN) for D in range(2, 20)],
         # ... and the per-image-layer instances per depth (`--mlp-layers D --image-layers 1|2`; 10 - 40 s each)
         *[("elbo_lane.hip", f"elbo_lane_i{D}", ["-DCL_LANE_PART=9", f"-DCL_LANE_NL={D}", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN) for D in range(2, 20)],
         ("elbo_lane.hip", "elbo_lane_b20", ["-DCL_LANE_PART=8", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + NNAN),      # the layer-block launches (act_out / dH_ext) at the default depth
         ("elbo_elem.hip", "elbo_elem", []), ("elbo_laue.hip", "elbo_laue", []), ("wide_gemm.hip", "wide_gemm", []),
         ("elbo_peel.hip", "elbo_peel", []), ("elbo_frozen.hip", "elbo_frozen", []),
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
    """Objects go to a directory of this process' own and the library is moved into place when it is complete: two builds at once (a
    test session that finds the library stale while `python -m careless_amd.build` runs) do not tread on each other."""
    import tempfile
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    lane_ok = _lane_flag_ok(hipcc)
    if not lane_ok and verbose:
        print("hipcc rejects " + " ".join(LANE_FLAG) + ": building elbo_lane.hip without it", flush=True)
    work = tempfile.mkdtemp(prefix=".build_", dir=LIBDIR)
    try:
        jobs = max(2, min(int(os.environ.get("CARELESS_HIP_BUILD_JOBS", "0")) or (os.cpu_count() or 4) + 2, 16))
        for s, stem, flags in UNITS:
            if not lane_ok:
                flags = [f for f in flags if f not in LANE_FLAG]
            o = os.path.join(work, stem + ".o")
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17"] + list(extra) + flags + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            while sum(1 for _, p in procs if p.poll() is None) >= jobs:      # (63 units: not all compilers at once on a small box)
                time.sleep(0.2)
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(o)
        for cmd, p in procs:
            if p.wait() != 0:
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
        tmp = os.path.join(work, os.path.basename(LIB))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-pthread", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        if os.environ.get("CARELESS_HIP_SKIP_ISA_CHECK") != "1":
            _isa_gate(tmp, verbose)
        os.replace(tmp, LIB)
    finally:
        for _, p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(work, ignore_errors=True)
    return LIB


def _isa_gate(lib: str, verbose: bool) -> None:
    """Build-time assertion (round 6): no kernel of the library may hold a pair of instructions closer than the gfx950 wait-state rules
    allow -- hipcc pads only the pairs it can see, and the kernels carry inline assembly (scripts/check_lane_isa.py; NOTEBOOK R6.1).  A
    library that fails is not installed."""
    import importlib.util
    path = os.path.join(HERE, "..", "scripts", "check_lane_isa.py")
    if not os.path.exists(path):          # (an installed copy without the scripts directory: the check is the repository's)
        return
    spec = importlib.util.spec_from_file_location("check_lane_isa", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("check_lane_isa", mod)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        return
    n, bad = mod.check_library(lib)
    if verbose:
        print(f"check_lane_isa: {n} kernels, {len(bad)} violations", flush=True)
    if bad:
        raise RuntimeError("gfx950 wait-state violations in the built library (scripts/check_lane_isa.py):\n" + "\n".join(bad[:20]))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
