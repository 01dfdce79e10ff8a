"""Diagnostic builds of the lane kernel (round 6, NOTEBOOK R6.1): the library with ONE compilation unit of elbo_lane.hip recompiled under
other flags, linked against the unchanged rest.  Cross-compiles here (no GPU); the libraries travel with the tree.

    python scripts/probe/build_lane_variants.py [name ...]       # -> careless_amd/lib/variants/libcareless_hip_<name>.so

A variant = (lane part, extra -D flags, keep the internal -amdgpu-mfma-vgpr-form flag?).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from careless_amd import build as B  # noqa: E402

CACHE = os.environ.get("CL_VARIANT_CACHE", "/tmp/cl_lane_variants")
OUT = os.path.join(B.LIBDIR, "variants")

OLD = ["-DCL_LANE_LRELU_ASM"]        # the inline-assembly LeakyReLU of rounds 2-5
VARIANTS = {
    # name: (part, defines, vgpr_form).  The round-6 sources ship the dZ_0-storing production instance with per-image layers
    # (<10, DM, true, false, true, NI>: lane part 4); what the first GPU call of round 6 ran (gpurun_out/r6_defect_variants.jsonl,
    # profiles/r6_lane_defect.txt) is these builds with the instance re-enabled by hand:
    "lrelu_asm": (4, OLD, True),                                          # round 5's LeakyReLU: one wait state short in two instances -> unrepeatable
    "lrelu_asm_pad": (4, OLD + ['-DCL_LANE_PAD_PRE="s_nop 1\\n\\t"', '-DCL_LANE_PAD_POST="\\n\\ts_nop 7\\n\\ts_nop 7"'], True),   # + wait states around every asm MFMA: still unrepeatable
    "lrelu_asm_sel_c": (4, OLD + ["-DCL_LANE_SEL_C"], True),              # + LeakyReLU derivative as plain C
    "shipped_part4": (4, [], True),                                       # the shipped form of the part (control)
    "novgprform": (4, [], False),                                         # without the internal LLVM flag
}


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stdout)


def base_objects(hipcc):
    """every unit of the production build, compiled once into the cache (keyed by the source hash)"""
    d = os.path.join(CACHE, "base_" + B.source_hash())
    os.makedirs(d, exist_ok=True)
    jobs = []
    objs = {}
    for s, stem, flags in B.UNITS:
        o = os.path.join(d, stem + ".o")
        objs[stem] = o
        if not os.path.exists(o):
            jobs.append([hipcc, f"--offload-arch={B.ARCH}", "-O3", "-fPIC", "-std=c++17"] + flags + ["-c", os.path.join(B.CSRC, s), "-o", o])
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(run, jobs))
    return objs


def main():
    names = sys.argv[1:] or list(VARIANTS)
    hipcc = B._hipcc()
    os.makedirs(OUT, exist_ok=True)
    objs = base_objects(hipcc)
    d = os.path.join(CACHE, "var_" + B.source_hash())
    os.makedirs(d, exist_ok=True)

    def one(name):
        part, defs, vf = VARIANTS[name]
        o = os.path.join(d, f"elbo_lane{part}_{name}.o")
        flags = [f"-DCL_LANE_PART={part}"] + (B.LANE_FLAG if vf else []) + B.NNAN + defs
        if not os.path.exists(o):
            run([hipcc, f"--offload-arch={B.ARCH}", "-O3", "-fPIC", "-std=c++17"] + flags + ["-c", os.path.join(B.CSRC, "elbo_lane.hip"), "-o", o])
        if os.environ.get("CL_VARIANT_ASM"):
            s = os.path.join(d, f"elbo_lane{part}_{name}.s")
            if not os.path.exists(s):
                run([hipcc, f"--offload-arch={B.ARCH}", "-O3", "-fPIC", "-std=c++17", "--cuda-device-only", "-S"] + flags + [os.path.join(B.CSRC, "elbo_lane.hip"), "-o", s])
        lib = os.path.join(OUT, f"libcareless_hip_{name}.so")
        link = [objs[st] if st != f"elbo_lane{part}" else o for st in objs]
        run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-pthread", "-o", lib] + link)
        return lib

    with ThreadPoolExecutor(4) as ex:
        for lib in ex.map(one, names):
            print(lib, flush=True)


if __name__ == "__main__":
    main()
