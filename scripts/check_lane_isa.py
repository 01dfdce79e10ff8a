"""Static check of the gfx950 code objects for data hazards that hipcc cannot see through inline assembly (round 6, NOTEBOOK R6.1).

hipcc pads the software wait states of gfx950 only between instructions it knows: an `asm` statement is opaque to its hazard recognizer,
so a pair with ONE end inside an `asm` statement gets nothing.  That is what made two instances of `elbo_lane_kernel` return results that
moved from run to run in round 5: the LeakyReLU was an inline-assembly `v_max_f32` whose result is the B operand of the next layer's first
MFMA, gfx950 wants TWO wait states between a vector-ALU write of a register and an MFMA that reads it, and in those two instances the
scheduler had left one.  The rules below are the ones the kernels of this repository can trip over (LLVM's GCNHazardRecognizer for
gfx940 / gfx950, checked against what hipcc 7.2 emits around compiler-known instructions: scripts/probe/hazard_padding_probe.hip):

  R1  vector-ALU write of a VGPR / AGPR            -> MFMA reads it as A, B or C                      2 wait states
  R2  MFMA writes D (fp32 MFMA of P passes)         -> anything reads or writes a register of D        P + 2
      (4x4x1: P = 2, 16x16x4: P = 8, 32x32x2: P = 16; an MFMA taking D whole as its C: 0; part of D as part of its C: P)
  R3  XDL MFMA reads C (not the fp32 / fp64 ones)  -> vector-ALU write of a register of C             P - 1
  R4  vector-ALU write of an SGPR pair / VCC        -> vector-ALU read of it                            2
                                                   -> vector memory instruction reads it (address)     5
                                                   -> v_readlane / v_writelane lane select             4
  R5  transcendental (exp, log, rcp, rsq, sqrt, sin, cos) -> vector-ALU read of its result             1

A wait state = one issued instruction (`s_nop N` = N + 1), counted along every path of the control-flow graph (both sides of a branch,
loop back edges included).  The input is the library itself: its gfx950 code objects are unbundled and disassembled with llvm-objdump,
so the check needs no GPU and sees exactly the instructions that run.

    python scripts/check_lane_isa.py                              # every kernel of careless_amd/lib/libcareless_hip.so
    python scripts/check_lane_isa.py --lib X.so --only elbo_lane  # one family of another build
    python scripts/check_lane_isa.py --asm file.s                 # a hipcc -S listing instead

Exit code 1 and one line per violation if there is any.  `tests/test_lane_isa.py` runs it over the shipped library.
"""
from __future__ import annotations

import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile
from collections import namedtuple
from typing import Dict, List, Optional, Set, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

Inst = namedtuple("Inst", "addr op ops text")
Reg = Tuple[str, int]

_REG = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b|\b(vcc|exec)(_lo|_hi)?\b")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
CARRY_OUT = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_")
VMEM = ("global_", "buffer_", "scratch_", "flat_", "tbuffer_")


def regs_of(tok: str) -> Set[Reg]:
    out: Set[Reg] = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            base = m.group(6)
            if m.group(7) in (None, "_lo"):
                out.add((base, 0))
            if m.group(7) in (None, "_hi"):
                out.add((base, 1))
    return out


def split_ops(s: str) -> List[str]:
    """operands of an instruction: commas at bracket depth 0; trailing modifiers (`cbsz:4 abid:1`, `offset:16`) stay on the last one"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def is_mfma(op: str) -> bool:
    return op.startswith("v_mfma_") or op.startswith("v_smfmac_")


def is_valu(op: str) -> bool:
    return op.startswith("v_") and not is_mfma(op)


def mfma_passes(op: str) -> int:
    m = re.search(r"_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 16
    mm = int(m.group(1))
    if mm == 4:
        return 2
    if op.endswith("_f64") or "f64" in op:
        return 4 if mm == 4 else 8 if mm == 16 else 16
    if mm == 16:
        blocks = re.search(r"_(\d+)b_", op)
        return 8 if not blocks else 8
    return 16


def mfma_is_xdl(op: str) -> bool:
    """fp32- and fp64-input MFMAs are not XDL operations on gfx940 / gfx950 (one wait state less)"""
    return not (re.search(r"x\d+_f32$", op) or re.search(r"x\d+_\d+b_f32$", op) or "f64" in op or op.endswith("xf32"))


def defs_uses(i: Inst) -> Tuple[Set[Reg], Set[Reg]]:
    """(registers written, registers read) of an instruction -- as far as the rules need them"""
    op, ops = i.op, i.ops
    if not ops:
        return set(), set()
    allr = [regs_of(o) for o in ops]
    if is_mfma(op):
        return allr[0], set().union(*allr[1:4]) if len(allr) > 1 else set()
    if op.startswith("v_"):
        nd = 1
        if op.startswith(CARRY_OUT) and (op.endswith("_e64") or op.startswith(("v_mad_", "v_div_scale"))):
            nd = 2
        if op.startswith("v_swap_") or op.startswith("v_permlane"):
            nd = 2
        if op.startswith("v_cmpx"):
            d = {("exec", 0), ("exec", 1)} | (allr[0] if op.endswith("_e64") and allr[0] and next(iter(allr[0]))[0] == "s" else set())
            return d, set().union(*allr)
        d = set().union(*allr[:nd])
        u = set().union(*allr[nd:]) if len(allr) > nd else set()
        if op.startswith("v_cndmask_b32_e32") or op.startswith(("v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32")) or op.startswith("v_div_fmas"):
            u |= {("vcc", 0), ("vcc", 1)}
        if op.startswith(("v_mac_", "v_fmac_", "v_pk_fmac", "v_writelane", "v_dot2c", "v_dot4c", "v_dot8c")):
            u |= d            # read-modify-write destinations
        return d, u
    if op.startswith(("ds_read", "ds_load")) or op.startswith(VMEM) and ("_load_" in op or "_atomic_" in op):
        if "_lds_" in op or op.endswith("_lds") or " lds" in i.text:
            return set(), set().union(*allr)
        if "_atomic_" in op and "sc0" not in i.text and "glc" not in i.text:
            return set(), set().union(*allr)
        return allr[0], set().union(*allr[1:]) if len(allr) > 1 else set()
    if op.startswith("s_"):
        if op.startswith(("s_cmp_", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_sendmsg")):
            return set(), set().union(*allr) if allr else set()
        return allr[0], set().union(*allr[1:]) if len(allr) > 1 else set()
    return set(), set().union(*allr)        # stores, ds_write, everything else: reads only


def wait_states(i: Inst) -> int:
    if i.op == "s_nop":
        try:
            return int(i.ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


# ---------------------------------------------------------------------------------------------------------------------------------
def unbundle(lib: str, outdir: str) -> List[str]:
    d = open(lib, "rb").read()
    paths, pos, n = [], 0, 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            break
        (ne,) = struct.unpack_from("<Q", d, i + 24)
        p = i + 32
        for _ in range(ne):
            off, size, ts = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + ts].decode(errors="replace")
            p += ts
            if "gfx950" in triple and size > 0:
                path = os.path.join(outdir, f"co_{n}.elf")
                with open(path, "wb") as f:
                    f.write(d[i + off:i + off + size])
                paths.append(path)
                n += 1
        pos = i + 24
    return paths


_DIS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")


def parse_objdump(text: str) -> Dict[str, Tuple[int, List[Inst]]]:
    """kernel name -> (start address, instructions)"""
    kernels: Dict[str, Tuple[int, List[Inst]]] = {}
    cur: Optional[List[Inst]] = None
    for ln in text.split("\n"):
        m = _SYM.match(ln)
        if m:
            cur = []
            kernels[m.group(2)] = (int(m.group(1), 16), cur)
            continue
        m = _DIS.match(ln)
        if m and cur is not None:
            op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
            cur.append(Inst(addr, op, split_ops(rest), ln.split("//")[0].strip()))
    return kernels


def parse_asm(text: str) -> Dict[str, Tuple[int, List[Inst]]]:
    """a hipcc -S listing: labels become pseudo addresses"""
    kernels: Dict[str, Tuple[int, List[Inst]]] = {}
    cur: Optional[List[Inst]] = None
    labels: Dict[str, int] = {}
    pend: List[Tuple[List[Inst], int, str]] = []
    n = 0
    for ln in text.split("\n"):
        t = ln.split(";")[0].rstrip() if not ln.strip().startswith(";;#") else ""
        s = t.strip()
        if not s or s.startswith("."):
            if s.startswith(".LBB") and s.endswith(":"):
                labels[s[:-1]] = n
            elif s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
                cur = None
            continue
        if s.endswith(":"):
            name = s[:-1]
            if name.startswith("_Z") or not name.startswith("."):
                cur = []
                kernels[name] = (n, cur)
            labels[name] = n
            continue
        if cur is None:
            continue
        parts = s.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ""
        cur.append(Inst(n, op, split_ops(rest), s))
        n += 4
    # resolve branch labels into pseudo addresses
    for name, (start, ins) in kernels.items():
        for k, i in enumerate(ins):
            if i.op.startswith(("s_cbranch", "s_branch")) and i.ops and i.ops[0] in labels:
                ins[k] = Inst(i.addr, i.op, [i.ops[0]], i.text + f" <{name}+{hex(labels[i.ops[0]] - start)}>")
    return {k: v for k, v in kernels.items() if v[1]}


def successors(ins: List[Inst], start: int) -> List[List[int]]:
    index = {i.addr: k for k, i in enumerate(ins)}
    succ: List[List[int]] = []
    for k, i in enumerate(ins):
        s: List[int] = []
        if i.op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"<[^>+]*\+(0x[0-9a-fA-F]+)>", i.text) or re.search(r"<[^>+]*\+(0x[0-9a-fA-F]+)>", " ".join(i.ops))
            tgt = index.get(start + int(m.group(1), 16)) if m else None
            if tgt is not None:
                s.append(tgt)
            if i.op != "s_branch" and k + 1 < len(ins):
                s.append(k + 1)
        elif i.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
            pass
        elif k + 1 < len(ins):
            s.append(k + 1)
        succ.append(s)
    return succ


def windows(ins: List[Inst], succ: List[List[int]], k: int, need: int):
    """(index, wait states between) of every instruction that can issue fewer than `need` wait states after instruction k.
    Time runs in wait states from the issue of instruction k; every instruction takes one (`s_nop N`: N + 1).  An MFMA cannot issue
    before the matrix pipe is free again: the pipe takes one MFMA per P passes (MI355X_MICROARCH.md, per-instruction cycle
    constants), so behind an MFMA the next MFMA issues P wait states later at the earliest.  (LLVM's own recognizer counts one wait
    state per instruction whatever it is, which is only more conservative.)"""
    seen: Dict[int, List[Tuple[int, int]]] = {}
    busy0 = mfma_passes(ins[k].op) if is_mfma(ins[k].op) else 0
    stack = [(n, 1, busy0) for n in succ[k]]
    while stack:
        j, t, busy = stack.pop()
        m = is_mfma(ins[j].op)
        if m and t < busy:
            t = busy
        if t - 1 >= need:
            continue
        st = seen.setdefault(j, [])
        if any(t0 <= t and b0 <= busy for t0, b0 in st):
            continue
        st.append((t, busy))
        yield j, t - 1
        t2 = t + wait_states(ins[j])
        busy2 = t + mfma_passes(ins[j].op) if m else busy
        for n in succ[j]:
            stack.append((n, t2, busy2))


def check_kernel(name: str, start: int, ins: List[Inst], quick_mfma_only: bool = False) -> List[str]:
    succ = successors(ins, start)
    du = [defs_uses(i) for i in ins]
    out: List[str] = []

    def report(rule, k, j, ws, need, regs):
        r = sorted(regs)[:4]
        out.append(f"{name}: {rule}: {ws} of {need} wait states between [{ins[k].addr:#x}] `{ins[k].text}` and [{ins[j].addr:#x}] `{ins[j].text}` on {r}")

    for k, i in enumerate(ins):
        d, u = du[k]
        op = i.op
        if is_mfma(op):
            P = mfma_passes(op)
            need = P + (3 if mfma_is_xdl(op) else 2)
            srcc = regs_of(i.ops[3]) if len(i.ops) > 3 else set()
            vec_d = {r for r in d if r[0] in "va"}
            for j, ws in windows(ins, succ, k, need):
                d2, u2 = du[j]
                hit = vec_d & (d2 | u2)
                if not hit:
                    continue
                if is_mfma(ins[j].op):
                    c2 = regs_of(ins[j].ops[3]) if len(ins[j].ops) > 3 else set()
                    ab2 = regs_of(ins[j].ops[1]) | regs_of(ins[j].ops[2])
                    if not (vec_d & (ab2 | c2)):
                        continue                      # only the destinations overlap: the matrix pipe writes in order
                    if not (vec_d & ab2):
                        if c2 == vec_d and mfma_is_xdl(ins[j].op) == mfma_is_xdl(op):
                            continue                  # D taken whole as the next MFMA's C (accumulate chain): no software wait states
                        # the result enters another MFMA as PART of its C, or is overwritten by it: P wait states (XDL: P + 2)
                        need_c = P + (2 if mfma_is_xdl(op) else 0)
                        if ws >= need_c:
                            continue
                        report("R2 MFMA result into another MFMA's C / D", k, j, ws, need_c, hit)
                        continue
                report("R2 MFMA result", k, j, ws, need, hit)
            if P > 1 and mfma_is_xdl(op):            # (gfx940 / gfx950: the fp32 / fp64 MFMAs have no such hazard)
                vc = {r for r in srcc if r[0] in "va"}
                for j, ws in windows(ins, succ, k, P - 1):
                    if is_valu(ins[j].op) and (vc & du[j][0]):
                        report("R3 MFMA reads C, vector ALU overwrites it", k, j, ws, P - 1, vc & du[j][0])
            continue
        if not is_valu(op):
            continue
        vec_d = {r for r in d if r[0] in "va"}
        sc_d = {r for r in d if r[0] in ("s", "vcc")}
        if vec_d:
            for j, ws in windows(ins, succ, k, 2):
                if is_mfma(ins[j].op) and (vec_d & du[j][1]):
                    report("R1 vector ALU result read by an MFMA", k, j, ws, 2, vec_d & du[j][1])
            if op.startswith(TRANS):
                for j, ws in windows(ins, succ, k, 1):
                    if is_valu(ins[j].op) and not ins[j].op.startswith(TRANS) and (vec_d & du[j][1]):
                        report("R5 transcendental result read by the next vector ALU instruction", k, j, ws, 1, vec_d & du[j][1])
        if sc_d and not quick_mfma_only:
            for j, ws in windows(ins, succ, k, 5):
                o2 = ins[j].op
                hit = sc_d & du[j][1]
                if not hit:
                    continue
                if is_valu(o2) or is_mfma(o2):
                    if o2.startswith(("v_readlane", "v_writelane")) and len(ins[j].ops) >= 3 and (sc_d & regs_of(ins[j].ops[2])):
                        if ws < 4:
                            report("R4 vector ALU writes SGPR, lane select reads it", k, j, ws, 4, hit)
                    elif ws < 2:
                        report("R4 vector ALU writes SGPR / VCC, vector ALU reads it", k, j, ws, 2, hit)
                elif o2.startswith(VMEM):
                    report("R4 vector ALU writes SGPR, vector memory reads it", k, j, ws, 5, hit)
    return out


def _check_code_object(arg) -> Tuple[int, List[str]]:
    co, only = arg
    syms = subprocess.run([OBJDUMP, "-t", co], stdout=subprocess.PIPE, text=True).stdout
    if only and only not in syms:
        return 0, []
    txt = subprocess.run([OBJDUMP, "-d", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    kernels = {k: v for k, v in parse_objdump(txt).items() if (not only or only in k) and v[1]}
    names = demangle(list(kernels))
    bad: List[str] = []
    for k, (start, ins) in kernels.items():
        bad += check_kernel(names[k], start, ins)
    return len(kernels), bad


def demangle(names: List[str]) -> Dict[str, str]:
    if not names:
        return {}
    try:
        r = subprocess.run([os.path.join(os.path.dirname(OBJDUMP), "llvm-cxxfilt")] + names, stdout=subprocess.PIPE, text=True, check=True)
        return dict(zip(names, r.stdout.strip().split("\n")))
    except Exception:
        return {n: n for n in names}


def check_library(lib: str, only: Optional[str] = None, jobs: int = 0) -> Tuple[int, List[str]]:
    """(number of kernels checked, violations) over every gfx950 code object bundled in `lib`"""
    from concurrent.futures import ProcessPoolExecutor
    with tempfile.TemporaryDirectory() as d:
        cos = unbundle(lib, d)
        if not cos:
            raise RuntimeError(f"no gfx950 code object found in {lib}")
        jobs = jobs or min(8, os.cpu_count() or 1, len(cos))
        if jobs > 1:
            with ProcessPoolExecutor(jobs) as ex:
                res = list(ex.map(_check_code_object, [(c, only) for c in cos]))
        else:
            res = [_check_code_object((c, only)) for c in cos]
    return sum(r[0] for r in res), [b for r in res for b in r[1]]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "careless_amd", "lib", "libcareless_hip.so"))
    ap.add_argument("--asm", default=None, help="a hipcc -S listing instead of the library")
    ap.add_argument("--only", default=None, help="substring of the (mangled) kernel names to check")
    ap.add_argument("--max", type=int, default=60)
    a = ap.parse_args()
    if a.asm:
        kernels = parse_asm(open(a.asm).read())
        if a.only:
            kernels = {k: v for k, v in kernels.items() if a.only in k}
        names = demangle(list(kernels))
        bad: List[str] = []
        for k, (start, ins) in kernels.items():
            bad += check_kernel(names[k], start, ins)
        n = len(kernels)
    else:
        n, bad = check_library(a.lib, a.only)
    for b in bad[: a.max]:
        print(b)
    print(f"{n} kernels checked, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
