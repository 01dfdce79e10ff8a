"""Instruction mix of a kernel's hottest loop from hipcc's assembly (-S): finds the backward branch whose body holds the most MFMAs
and counts MFMAs (by shape), other vector, LDS, memory, scalar, wait and nop instructions in it.
    python scripts/isa_count.py file.s kernel_mangled_name
"""
import re, sys, collections

def body(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
            continue
        if on and ln.startswith(".Lfunc_end"):
            break
        if on:
            out.append(ln.rstrip("\n"))
    return out

def main():
    lines = body(sys.argv[1], sys.argv[2])
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < i:
                n = sum("v_mfma" in x for x in lines[t:i])
                if best is None or n > best[0]:
                    best = (n, t, i)
    n, a, b = best
    cnt = collections.Counter()
    det = collections.Counter()
    for l in lines[a:b]:
        s = l.strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            cnt["mfma " + ("4x4" if "4x4x1" in op else "16x16" if "16x16" in op else op)] += 1
        elif op.startswith("v_accvgpr"):
            cnt["accvgpr moves"] += 1
        elif op.startswith("v_"):
            cnt["valu"] += 1
            det[op] += 1
        elif op.startswith("ds_"):
            cnt["lds"] += 1
            det[op] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cnt["vmem"] += 1
            det[op] += 1
        elif op == "s_waitcnt":
            cnt["s_waitcnt"] += 1
        elif op == "s_nop":
            cnt["s_nop"] += 1
        elif op.startswith("s_"):
            cnt["salu/branch"] += 1
        else:
            cnt["other"] += 1
    print(f"loop lines {a}..{b}: {dict(cnt)}")
    for k, v in det.most_common(40):
        print(f"  {k:28s} {v}")

main()
