"""The shipped code objects obey the gfx950 wait-state rules that hipcc cannot check across inline assembly (scripts/check_lane_isa.py;
NOTEBOOK R6.1: round 5's run-to-run defect was an inline-assembly v_max_f32 one wait state in front of an MFMA that read its result).
No GPU needed: the library's gfx950 code objects are unbundled and disassembled with llvm-objdump."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_lane_isa", os.path.join(ROOT, "scripts", "check_lane_isa.py"))
isa = importlib.util.module_from_spec(spec)
sys.modules["check_lane_isa"] = isa
spec.loader.exec_module(isa)

LIB = os.path.join(ROOT, "careless_amd", "lib", "libcareless_hip.so")


def _listing(body: str) -> str:
    return "k:\n" + "\n".join("\t" + ln for ln in body.strip().split("\n")) + "\n\ts_endpgm\n"


def _check(body: str):
    k = isa.parse_asm(_listing(body))
    (start, ins), = k.values()
    return isa.check_kernel("k", start, ins)


def test_scanner_sees_the_round5_defect():
    """the pair of round 5's withdrawn instances, as hipcc emitted it (one unrelated instruction between), and its repaired forms"""
    bad = _check("""
v_max_f32 v38, v6, v18
v_pk_mul_f32 v[16:17], s[98:99], v[8:9]
v_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, 0 cbsz:4
""")
    assert len(bad) == 1 and "R1" in bad[0] and "1 of 2" in bad[0], bad
    assert _check("v_max_f32 v38, v6, v18\nv_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, 0 cbsz:4")[0].count("0 of 2") == 1
    assert not _check("v_max_f32 v38, v6, v18\ns_nop 1\nv_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, 0 cbsz:4")
    assert not _check("v_max_f32 v38, v6, v18\nv_mov_b32 v1, v2\nv_mov_b32 v3, v2\nv_mfma_f32_16x16x4_f32 a[0:3], v34, v38, a[0:3]")
    assert not _check("ds_read_b32 v38, v1\ns_waitcnt lgkmcnt(0)\nv_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, 0 cbsz:4")      # a load is not a vector-ALU write


def test_scanner_follows_branches_and_loops():
    bad = _check("""
v_max_f32 v38, v6, v18
s_cbranch_scc1 .LBB0_2
v_mov_b32 v1, v2
v_mov_b32 v1, v2
.LBB0_2:
v_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, 0 cbsz:4
""")
    assert len(bad) == 1 and "1 of 2" in bad[0], bad          # the taken branch skips the two fillers
    bad = _check("""
.LBB0_1:
v_mfma_f32_4x4x1_16b_f32 v[10:13], v34, v38, v[10:13] cbsz:4
s_add_i32 s1, s1, -1
v_max_f32 v38, v6, v18
s_cbranch_scc1 .LBB0_1
""")
    assert len(bad) == 1 and "R1" in bad[0], bad              # across the back edge: v_max, branch, MFMA


def test_scanner_mfma_result_rules():
    # a 16x16x4 result read by the vector ALU ten wait states later is fine, nine is not; the accumulate chain needs none
    fill = "\n".join(["v_mov_b32 v1, v2"] * 9)
    assert _check(f"v_mfma_f32_16x16x4_f32 a[0:3], v4, v5, a[0:3]\n{fill}\nv_accvgpr_read_b32 v9, a1")
    assert not _check(f"v_mfma_f32_16x16x4_f32 a[0:3], v4, v5, a[0:3]\n{fill}\nv_mov_b32 v1, v2\nv_accvgpr_read_b32 v9, a1")
    assert not _check("v_mfma_f32_16x16x4_f32 a[0:3], v4, v5, a[0:3]\nv_mfma_f32_16x16x4_f32 a[0:3], v6, v7, a[0:3]")
    # behind an MFMA the next MFMA waits for the matrix pipe: two of them in between are sixteen wait states
    assert not _check("v_mfma_f32_16x16x4_f32 a[0:3], v4, v5, a[0:3]\nv_mfma_f32_16x16x4_f32 a[4:7], v4, v5, a[4:7]\n"
                      "v_mfma_f32_16x16x4_f32 a[8:11], v4, v5, a[8:11]\nv_accvgpr_read_b32 v9, a1")
    # vector ALU writes an SGPR pair, the next vector ALU instruction reads it: two wait states (gfx940 / gfx950)
    assert _check("v_cmp_lt_f32_e64 s[8:9], 0, v1\nv_cndmask_b32_e64 v2, v3, v4, s[8:9]")
    assert not _check("v_cmp_lt_f32_e64 s[8:9], 0, v1\nv_mov_b32 v7, v8\nv_mov_b32 v7, v8\nv_cndmask_b32_e64 v2, v3, v4, s[8:9]")


@pytest.mark.skipif(not os.path.exists(isa.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_every_shipped_kernel_obeys_the_wait_state_rules():
    if not os.path.exists(LIB):
        pytest.skip("libcareless_hip.so is not built")
    n, bad = isa.check_library(LIB)
    assert n >= 500, n                                        # (the library holds ~575 kernels: the unbundling found them)
    assert not bad, "\n".join(bad[:20])
