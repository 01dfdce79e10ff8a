"""The library's routing, as DESIGN.md section 4 states it, held by a test: which kernel instance a training step of a given scaler shape
runs (`engine.kernel_name()` = `cl_mlp_kernel_name`: the library's own routing restated next to `cl_launch_mlp`), whether the first
layer is peeled, whether the scaler runs as a chain of layer blocks or layer by layer.  A shape that silently falls off its kernel
costs 2 - 4 x (profiles/r5_envelope*.txt) and no parity test notices.  Reference flags that choose the shape: careless/args/scaling.py:21-31
(--mlp-layers, --mlp-width), args/positional_encoding.py:24-37 (metadata columns)."""
import pytest

from tests import util

pytestmark = pytest.mark.gpu

#        L   w   d    kernel-name fragment                          peel   chain blocks   wide
TABLE = [
    (20, 10, 5,  "elbo_lane_kernel<10, 8, false, false",          False, None, False),      # the CLI default
    (20, 10, 21, "elbo_lane_kernel<10, 0, false, false",          False, None, False),      # + two encoded keys: metadata as LDS rows
    (20, 10, 37, "elbo_lane_kernel<10, 15, false, false, true",   True,  None, False),      # + four encoded keys: peeled first layer
    (20, 8,  64, "elbo_lane_kernel<8, 8, false, false, true",     True,  None, False),
    (20, 12, 5,  "elbo_lane_kernel<12, 8, false, false",          False, None, False),      # widths 11, 12 (round 6): the lane kernel's twelve-wide register instances
    (20, 12, 12, "elbo_narrow_kernel<2, 3, 8",                    False, None, False),      # ... at the default depth only up to 8 columns (spilled registers beyond: the narrow kernel is ahead)
    (20, 13, 5,  "elbo_narrow_kernel<2, 4, 8",                    False, None, False),
    (10, 10, 5,  "elbo_lane_kernel<10, 15, false, false, false, 0, 10>",  False, None, False),   # other depths at widths 7 .. 10 (round 6): the lane kernel compiled for the depth
    (16, 8,  12, "elbo_lane_kernel<8, 15, false, false, false, 0, 16>",  False, None, False),
    (12, 4,  5,  "elbo_narrow_kernel<2, 2, 8",                    False, None, False),      # narrower than 5: the narrow kernel
    (10, 10, 21, "elbo_lane_kernel<10, 15, false, false, true, 0, 10>", True, None, False),  # other depths, more than 15 columns: peeled, the depth's dZ_0-storing lane instance (round 6)
    (9,  12, 21, "elbo_lane_kernel<12, 15, false, false, true, 0, 9>", True, None, False),
    (7,  15, 5,  "elbo_narrow_kernel<2, 4, 8",                    False, None, False),
    (7,  15, 40, "elbo_mlp_kernel<16, 64, 20, 0, KS=4",           False, None, False),      # widths 13 .. 15 on many columns: the 16-wide instance itself
    (10, 16, 5,  "elbo_mlp_kernel<16, 8, 20, 0, KS=5",            False, None, False),      # width exactly 16: its own instance
    (20, 16, 21, "elbo_mlp_kernel<16, 32, 20, 0, KS=5",           False, None, False),
    (10, 20, 5,  "elbo_mlp_kernel<32, 8, 10, 0",                  False, None, False),
    (5,  64, 21, "elbo_mlp_kernel<64, 32, 5, 0",                  False, None, False),      # the bench line's kernel
    (24, 10, 5,  "elbo_lane_kernel<10, 15, false, false, true",   False, 2,    False),      # deeper than one launch at width <= 10 (round 6): the last 20 layers on the lane kernel (dZ_0 out), 4 in front on the 16-wide one
    (45, 8,  5,  "elbo_lane_kernel<8, 8, false, false, true",     False, 3,    False),
    (24, 12, 5,  "elbo_lane_kernel<12, 15, false, false, true, 0, 19>", False, 2, False),   # ... widths 11, 12: the last NINETEEN layers on the twelve-wide lane instance, five in front
    (24, 14, 5,  "elbo_mlp_kernel<16, 32, 20, 0, chain",          False, 2,    False),      # ... wider: two layer blocks of the 16-wide kernel (the last one's input is 14 wide)
    (12, 64, 5,  "elbo_mlp_kernel<64, 64, 5, 0, chain",           False, 3,    False),
    (3,  128, 5, "wide_sq_kernel",                                False, None, True),       # layer by layer
]


@pytest.mark.parametrize("L,w,d,frag,peel,blocks,wide", TABLE, ids=[f"{r[0]}x{r[1]}_d{r[2]}" for r in TABLE])
def test_training_step_runs_on_the_kernel_the_design_names(L, w, d, frag, peel, blocks, wide):
    from careless_amd.engine import ElboEngine
    data, cfg, params, x, u_f, eta = util.make_problem(N=300, R=30, d0=d, L=L, w=w, S=1, perturb=0.02)
    eng = ElboEngine(util.build_model(data, cfg, params, L, w), util.reference_inputs(data), seed=1)
    name = eng.kernel_name()
    assert frag in name, name
    assert bool(eng.peel) == peel and eng.wide == wide
    assert (eng.blocks is None) == (blocks is None) and (blocks is None or len(eng.blocks) == blocks)
    eng.alloc_history(1)
    eng.train_step(0)                                           # ... and the step it names runs


#              L   w   d   K  laue   kernel-name fragment                                                  peel
IMGL_TABLE = [
    (20, 10, 5,  2, False, "elbo_lane_kernel<10, 8, true, false, false, 2> (image layers)",        False),   # `--image-layers 2` on the CLI default
    (20, 10, 21, 2, False, "elbo_lane_kernel<10, 15, true, false, true, 2> (image layers)",        True),    # ... + two encoded keys: peeled, dZ_0 out
    (10, 10, 5,  2, False, "elbo_lane_kernel<10, 15, true, false, false, 2, 10> (image layers)",   False),   # other depths (round 6): the depth's per-image-layer instance
    (14, 7,  12, 1, False, "elbo_lane_kernel<10, 15, true, false, false, 1, 14> (image layers)",   False),
    (6,  10, 21, 1, False, "elbo_lane_kernel<10, 15, true, false, true, 1, 6> (image layers)",     True),
    (12, 8,  0,  2, True,  "elbo_lane_kernel<10, 15, true, false, false, 2, 12> (image layers)",   False),   # single-pass Laue
    (10, 4,  5,  2, False, "elbo_mlp_kernel<16, 8, 24, 0, image layers",                           False),   # narrower than 5 at another depth: the 16-wide IMGL instance
    (20, 10, 5,  3, False, "elbo_lane_kernel<10, 15, true, false, false, 3> (image layers)",       False),   # three per-image layers on the default depth: a unit of their own
    (20, 8,  21, 3, False, "elbo_lane_kernel<10, 15, true, true, false, 3> (image layers)",        True),    # (... behind a peeled layer: the full instance)
    (12, 10, 5,  3, False, "elbo_lane_kernel<10, 15, true, false, false, 3, 12> (image layers)",   False),   # ... on another depth
    (5,  6,  21, 3, False, "elbo_lane_kernel<10, 15, true, true, false, 3, 5> (image layers)",     True),
    (19, 10, 5,  3, False, "elbo_mlp_kernel<16, 8, 24, 0, image layers",                           False),   # (19 + 3: the compiler gives up on the 22-layer full instance)
    (2,  11, 36, 2, False, "elbo_mlp_kernel<32, 64, 5, 0, image layers",                           False),   # width <= 15 on more than 32 columns: the 32-wide instance (the 16-wide one is withdrawn)
    (8,  13, 50, 1, False, "elbo_mlp_kernel<32, 64, 10, 0, image layers",                          False),
    (12, 12, 36, 1, False, "wide_gemm_kernel",                                                     False),   # ... deeper than it holds: layer by layer
    (20, 10, 5,  4, False, "elbo_mlp_kernel<16, 8, 24, 0, image layers",                           False),   # four of them: the 16-wide IMGL instance
]


@pytest.mark.parametrize("L,w,d,K,laue,frag,peel", IMGL_TABLE, ids=[f"{r[0]}x{r[1]}_d{r[2]}_K{r[3]}{'_laue' if r[4] else ''}" for r in IMGL_TABLE])
def test_per_image_layers_run_on_the_kernel_the_design_names(L, w, d, K, laue, frag, peel):
    """`--image-layers K` (careless/args/scaling.py:33-37) at the default and at other depths"""
    from careless_amd.engine import ElboEngine
    kw = dict(N=600, R=30, L=L, w=w, S=1, perturb=0.02, image_layers=K, n_images=5)
    if laue:
        kw.update(laue=True)
    else:
        kw.update(d0=d)
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    eng = ElboEngine(util.build_model(data, cfg, params, L, w), util.reference_inputs(data), seed=1)
    name = eng.kernel_name()
    assert frag in name, name
    assert bool(eng.peel) == peel
    eng.alloc_history(1)
    eng.train_step(0)

