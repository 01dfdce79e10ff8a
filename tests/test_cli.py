"""End-to-end runs of `careless_amd mono|poly` on the reference's fixture, modelled on the reference's tests/test_cli.py:36-228
(same flag combinations, same assertions: output files exist and are readable, space group carried through, resolution cut kept)."""
import os

import numpy as np
import pytest

from careless_amd.io.asu import inv_d2
from careless_amd.io.mtz import read_mtz
from tests.mtz_fixture import PYP

ON = os.path.join(os.path.dirname(PYP), "pyp_2ms.mtz")             # the reference's second time point (tests/data/pyp_2ms.mtz)
ON_P3 = os.path.join(os.path.dirname(PYP), "pyp_2ms_P3.mtz")       # ... re-indexed in P 3 (tests/data/pyp_2ms_P3.mtz)

pytestmark = pytest.mark.gpu
niter = 10


def _run(flags, files, out, separate):
    from careless_amd.careless import run_careless
    from careless_amd.parser import parser
    cmd = flags.split() + (["--separate-files"] if separate else []) + list(files) + [out]
    args = parser.parse_args(cmd)
    model, hist = run_careless(args)
    assert len(hist["loss"]) == args.iterations and np.all(np.isfinite(hist["loss"]))
    for i in range(len(files) if separate else 1):
        f = out + f"_{i}.mtz"
        assert os.path.exists(f)
        ds, src = read_mtz(f), read_mtz(files[i])
        assert ds.spacegroup_number == src.spacegroup_number and ds.symops == src.symops
        if args.anomalous:                                   # reference tests/test_cli.py:49-50: Friedel mates in separate columns
            assert "F(+)" in ds.columns and ds.types["F(+)"] == "G" and ds.keys()[3:13] == ["F(+)", "SigF(+)", "F(-)", "SigF(-)", "I(+)", "SigI(+)",
                                                                                              "I(-)", "SigI(-)", "N(+)", "N(-)"]
            f = np.concatenate([ds.columns["F(+)"], ds.columns["F(-)"]])
            assert len(ds) > 0 and np.isfinite(f).any() and np.all(f[np.isfinite(f)] > 0)
        else:
            assert len(ds) > 0 and np.all(np.isfinite(ds.columns["F"])) and np.all(ds.columns["SigF"] > 0) and np.all(ds.columns["N"] > 0)
        if args.dmin is not None:
            assert (1.0 / np.sqrt(inv_d2(ds.hkl(), ds.cell))).min() >= args.dmin
        assert os.path.exists(out + f"_predictions_{i}.mtz")
    for suffix in ("_history.csv", "_structure_factor", "_scale"):
        assert os.path.exists(out + suffix)
    return args, model, hist


@pytest.mark.parametrize("mode", ["mono", "poly"])
@pytest.mark.parametrize("change_sg", [False, True])
@pytest.mark.parametrize("ev11,dmin,anomalous,isigi,dof,separate", [(False, None, False, None, None, False), (True, 7.0, True, 1.0, 12.0, True),
                                                                    (True, None, False, None, 12.0, False), (False, 7.0, True, None, None, True)])
def test_twofile(tmp_path, mode, change_sg, ev11, dmin, anomalous, isigi, dof, separate):
    """reference tests/test_cli.py:63-90: the off / on pair, together or as separate outputs (then also with different space groups)"""
    if change_sg and not separate:
        pytest.skip("different space groups cannot be merged into one file (the reference skips this combination too)")
    flags = f"{mode} --disable-gpu --iterations={niter} --disable-progress-bar --mlp-layers 4 dHKL,image_id"
    flags += " --refine-uncertainties" if ev11 else ""
    flags += f" --dmin={dmin}" if dmin is not None else ""
    flags += " --anomalous" if anomalous else ""
    flags += f" --isigi-cutoff={isigi}" if isigi is not None else ""
    flags += f" --studentt-likelihood-dof={dof}" if dof is not None else ""
    _run(flags, [PYP, ON_P3 if change_sg else ON], str(tmp_path / "out"), separate)


def test_spacegroups_flag_overrides_the_file_header(tmp_path):
    """`--spacegroups` with one name per file (reference formatter.py:254-263, 301-302: the file's space group is replaced before
    the indices are mapped to the asymmetric unit): naming P 3 for the P 63 file gives what a copy of that file with a P 3 header gives"""
    from careless_amd.careless import run_careless
    from careless_amd.io.mtz import write_mtz
    from careless_amd.io.spacegroups import lookup
    from careless_amd.parser import parser
    flags = f"mono --iterations={niter} --disable-progress-bar --mlp-layers 3 --separate-files dHKL,image_id"
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    src, copy = read_mtz(ON), str(tmp_path / "on_as_p3.mtz")
    ops, name, number = lookup("P 3")
    write_mtz(copy, src.columns, src.types, src.cell, ops, name, number)
    run_careless(parser.parse_args(flags.split() + [PYP, copy, a]))
    run_careless(parser.parse_args(flags.split() + ["--spacegroups=P 63,P 3", PYP, ON, b]))
    for i in range(2):
        x, y = read_mtz(a + f"_{i}.mtz"), read_mtz(b + f"_{i}.mtz")
        assert x.spacegroup_number == y.spacegroup_number == (173, 143)[i] and np.array_equal(x.hkl(), y.hkl())
        assert np.allclose(x.columns["F"], y.columns["F"], rtol=1e-5) and np.array_equal(x.columns["N"], y.columns["N"])


@pytest.mark.parametrize("mode", ["mono", "poly"])
@pytest.mark.parametrize("optimize_r", [False, True])
def test_double_wilson(tmp_path, mode, optimize_r):
    flags = f"{mode} --iterations={niter} --disable-progress-bar --mlp-layers 3 dHKL,image_id --double-wilson-parents=None,0"
    flags += " --optimize-double-wilson-r" if optimize_r else ""
    _, model, hist = _run(flags + " --double-wilson-r=0.0,0.9", [PYP, ON], str(tmp_path / "out"), True)
    if optimize_r:
        assert "rDW_1" in hist and abs(hist["rDW_1"][0] - 0.9) < 1e-6
    with pytest.raises(ValueError):
        _run(flags + " --double-wilson-r=0.0,1.0", [PYP, ON], str(tmp_path / "out2"), True)


def test_image_layers_crossvalidation_and_reloading(tmp_path):
    out = str(tmp_path / "a")
    flags = (f"mono --iterations={niter} --disable-progress-bar --mlp-layers 3 --mlp-width 8 --image-layers 2 --test-fraction 0.2 "
             "--merge-half-datasets --half-dataset-repeats 2 --positional-encoding-keys X,Y dHKL,image_id,Hobs,Kobs,Lobs")
    args, model, hist = _run(flags, [PYP], out, False)
    assert "NLL_val" in hist and os.path.exists(out + "_xval_0.mtz")
    x = read_mtz(out + "_xval_0.mtz")
    assert set(np.unique(x.columns["half"])) == {0.0, 1.0} and set(np.unique(x.columns["repeat"])) == {0.0, 1.0}
    p = read_mtz(out + "_predictions_0.mtz")
    assert set(np.unique(p.columns["test"])) == {0.0, 1.0} and np.all(np.isfinite(p.columns["Ipred"]))
    # second run: start from the saved weights, freeze the scales (reference careless.py:48-56)
    out2 = str(tmp_path / "b")
    flags2 = (f"mono --iterations=3 --disable-progress-bar --mlp-layers 3 --mlp-width 8 --image-layers 2 --scale-file {out}_scale "
              f"--structure-factor-file {out}_structure_factor --freeze-scales --positional-encoding-keys X,Y dHKL,image_id,Hobs,Kobs,Lobs")
    _, model2, _ = _run(flags2, [PYP], out2, False)
    assert np.array_equal(model2.scaling_model.flat.cpu().numpy(), model.scaling_model.flat.cpu().numpy())   # frozen => unchanged


@pytest.mark.parametrize("scale_bijector", ["exp", "softplus"])
@pytest.mark.parametrize("image_layers", [None, 2])
def test_scale_bijector(tmp_path, scale_bijector, image_layers):
    """reference tests/test_cli.py:211-228: default scaler (20 x 10), optionally with two per-image layers on top"""
    flags = f"mono --disable-gpu --iterations={niter} --disable-progress-bar --scale-bijector={scale_bijector} dHKL,image_id"
    if image_layers is not None:
        flags = flags.replace("mono ", f"mono --image-layers={image_layers} ")
    _run(flags, [PYP], str(tmp_path / "out"), False)


@pytest.mark.parametrize("extra", ["--freeze-structure-factors", "--freeze-scales", "--clipvalue=1.", "--clipnorm=1.", "--global-clipnorm=1.",
                                   "--kl-weight=0.5", "--wilson-prior-b=20.", "--mc-samples=3 --studentt-likelihood-dof=8 --refine-uncertainties",
                                   "--disable-image-scales --mlp-width=6 --mlp-layers=25", "--disable-metadata-standardization"])
def test_flags_execute(tmp_path, extra):
    """reference tests/test_cli.py:165-204 (freeze flags, clipping) and further flags of the path, default 20 x 10 scaler"""
    _run(f"mono --disable-gpu --iterations={niter} --disable-progress-bar {extra} dHKL,image_id", [PYP], str(tmp_path / "out"), False)


@pytest.mark.parametrize("which", ["scale", "structure_factor"])
def test_weight_save_and_load(tmp_path, which):
    """reference tests/test_cli.py:120-163"""
    out = str(tmp_path / "out")
    flags = f"mono --disable-gpu --iterations={niter} --disable-progress-bar dHKL,image_id"
    _, model, _ = _run(flags, [PYP], out, False)
    flag = f"--scale-file={out}_scale" if which == "scale" else f"--structure-factor-file={out}_structure_factor"
    out2 = str(tmp_path / "out_reloaded")
    from careless_amd.careless import run_careless
    from careless_amd.parser import parser
    args = parser.parse_args((flags.replace(f"--iterations={niter}", "--iterations=1") + f" {flag}").split() + [PYP, out2])
    model2, hist2 = run_careless(args)
    assert os.path.exists(out2 + "_0.mtz") and np.isfinite(hist2["loss"][0])


def test_crystfel(tmp_path):
    """reference tests/test_cli.py:112-120"""
    stream = os.path.join(os.path.dirname(PYP), "crystfel.stream")
    from careless_amd.careless import run_careless
    from careless_amd.parser import parser
    out = str(tmp_path / "out")
    _, hist = run_careless(parser.parse_args(f"mono --disable-gpu --iterations={niter} --disable-progress-bar --spacegroups=1 dHKL,image_id {stream} {out}".split()))
    assert len(hist["loss"]) == niter and np.all(np.isfinite(hist["loss"])) and read_mtz(out + "_0.mtz").spacegroup_number == 1
    with pytest.raises(ValueError):                                    # careless poly should fail with a clear error message
        run_careless(parser.parse_args(f"poly --iterations={niter} --spacegroups=1 dHKL,image_id {stream} {out}".split()))


def test_save_data_manager(tmp_path):
    import pickle
    out = str(tmp_path / "out")
    _run(f"mono --iterations=3 --disable-progress-bar --mlp-layers 2 --save-data-manager dHKL,image_id", [PYP], out, False)
    dm = pickle.load(open(out + "_data_manager.pickle", "rb"))
    assert len(dm.inputs) == 6 and len(dm.asu_collection.centric) == 485


def test_preformatted_npz_input(tmp_path):
    from careless_amd.io.formats import save_inputs_npz
    from careless_amd.io.formatter import MonoFormatter
    inputs, rac = MonoFormatter(None, None, None, ["dHKL", "image_id"], False, False).format_files([PYP])
    npz = str(tmp_path / "in.npz")
    save_inputs_npz(npz, inputs, rac)
    from careless_amd.careless import run_careless
    from careless_amd.parser import parser
    out = str(tmp_path / "o")
    _, hist = run_careless(parser.parse_args(f"mono --iterations=5 --disable-progress-bar --mlp-layers 2 dHKL,image_id {npz} {out}".split()))
    assert len(hist["loss"]) == 5 and read_mtz(out + "_0.mtz").spacegroup_number == 173


@pytest.mark.parametrize("split", ["rows", "owners", "rows_two_piece_message"])
def test_data_parallel_cli_two_ranks_match_one(tmp_path, split):
    """`python -m careless_amd mono ...` as two one-process-per-GPU ranks (here: both on this GPU, gloo backend) writes the same
    merged amplitudes and history as the single-process run: observations sharded, one all-reduce per step, in-kernel noise keyed
    by global indices, rank 0 writes the files (careless_amd/careless.py: _data_parallel).  Both splits: rows (what two ranks run
    by default) and reflection owners (opt-in since round 5, CARELESS_HIP_OWNER_SHARD=1: every rank updates its own reflections' q(F) only, the ranks
    exchange them after training, the validation rows follow their reflection's owner); and the row split with its message in two
    pieces (round 4: the scaler's part all-reduced beside cl_tn_backward, a and b after it)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = f"mono --iterations={niter} --disable-progress-bar --mlp-layers 3 --test-fraction 0.2 dHKL,image_id".split()
    one = str(tmp_path / "one")
    _run(" ".join(flags), [PYP], one, False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = str(tmp_path / "two")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CARELESS_DIST_BACKEND="gloo", PYTHONPATH=root, CARELESS_HIP_OWNER_SHARD="1" if split == "owners" else "0",
                   CARELESS_HIP_SPLIT_MESSAGE="1" if split == "rows_two_piece_message" else "0")
        procs.append(subprocess.Popen([sys.executable, "-m", "careless_amd"] + flags + [PYP, two], env=env, cwd=root))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    a, b = read_mtz(one + "_0.mtz"), read_mtz(two + "_0.mtz")
    assert np.array_equal(a.hkl(), b.hkl()) and np.array_equal(a.columns["N"], b.columns["N"])
    assert np.allclose(a.columns["F"], b.columns["F"], rtol=1e-4) and np.allclose(a.columns["SigF"], b.columns["SigF"], rtol=1e-3)
    ha = np.genfromtxt(one + "_history.csv", delimiter=",", names=True)
    hb = np.genfromtxt(two + "_history.csv", delimiter=",", names=True)
    assert np.allclose(ha["loss"], hb["loss"], rtol=1e-5) and np.allclose(ha["NLL_val"], hb["NLL_val"], rtol=1e-4)


def test_deterministic_command_line_runs_write_identical_files(tmp_path, monkeypatch):
    """CARELESS_HIP_DETERMINISTIC=1: two runs of the command line on the same files write bit-identical merged amplitudes and
    histories (no float atomics anywhere in the step: include/careless_hip.h, deterministic mode); with positional encodings, i.e.
    the metadata width the default scaler gets from `--positional-encoding-keys`, and with the shard cut into several launches."""
    monkeypatch.setenv("CARELESS_HIP_DETERMINISTIC", "1")
    monkeypatch.setenv("CARELESS_HIP_MAX_LAUNCH_BYTES", str(4 * 24 * 60))         # a few dozen rows per launch
    flags = f"mono --iterations={niter} --disable-progress-bar --mlp-layers 3 --mc-samples 3 --studentt-likelihood-dof 8 " \
            "--positional-encoding-keys X,Y --positional-encoding-frequencies 2 dHKL,image_id,X,Y"
    outs = []
    for k in range(2):
        out = str(tmp_path / f"run{k}")
        _run(flags, [PYP], out, False)
        outs.append(out)
    a, b = read_mtz(outs[0] + "_0.mtz"), read_mtz(outs[1] + "_0.mtz")
    for col in ("F", "SigF", "I", "SigI"):
        assert np.array_equal(a.columns[col], b.columns[col]), col
    ha = np.genfromtxt(outs[0] + "_history.csv", delimiter=",", names=True)
    hb = np.genfromtxt(outs[1] + "_history.csv", delimiter=",", names=True)
    assert np.array_equal(ha["NLL"], hb["NLL"]) and np.allclose(ha["loss"], hb["loss"], rtol=1e-12)
    # ... and a `poly` command (single-pass Laue: packed layout, per-observation stores by the caller's row; round 4)
    monkeypatch.delenv("CARELESS_HIP_MAX_LAUNCH_BYTES")
    flags = f"poly --iterations={niter} --disable-progress-bar --mlp-layers 3 --mc-samples 2 dHKL,image_id"
    outs = []
    for k in range(2):
        out = str(tmp_path / f"poly{k}")
        _run(flags, [PYP], out, False)
        outs.append(out)
    a, b = read_mtz(outs[0] + "_0.mtz"), read_mtz(outs[1] + "_0.mtz")
    for col in ("F", "SigF", "I", "SigI"):
        assert np.array_equal(a.columns[col], b.columns[col]), col
    ha = np.genfromtxt(outs[0] + "_history.csv", delimiter=",", names=True)
    hb = np.genfromtxt(outs[1] + "_history.csv", delimiter=",", names=True)
    assert np.array_equal(ha["NLL"], hb["NLL"])
    # ... and `--refine-uncertainties` on the default scaler's depth and on a scaler wider than 64 (round 4: the Evans-2011 gradients
    # leave the kernels as per-wave stores; the layer-by-layer path's slot kernel stores per (row, sample))
    # ... and `--image-layers 2` on the default scaler and on a shallower one (round 6: the lane kernel's per-image-layer instances, one wave per image)
    for tag, extra in (("ev11", "--refine-uncertainties --mlp-layers 20 --mc-samples 2"), ("wide", "--mlp-layers 2 --mlp-width 72 --mc-samples 4"),
                       ("imgl", "--image-layers 2 --mc-samples 2"), ("imgl_depth", "--image-layers 1 --mlp-layers 6 --mc-samples 3 --studentt-likelihood-dof 8")):
        flags = f"mono --iterations={niter} --disable-progress-bar {extra} dHKL,image_id"
        outs = []
        for k in range(2):
            out = str(tmp_path / f"{tag}{k}")
            _run(flags, [PYP], out, False)
            outs.append(out)
        a, b = read_mtz(outs[0] + "_0.mtz"), read_mtz(outs[1] + "_0.mtz")
        for col in ("F", "SigF", "I", "SigI"):
            assert np.array_equal(a.columns[col], b.columns[col]), (tag, col)


def test_default_scaler_on_four_positionally_encoded_keys(tmp_path):
    """`--positional-encoding-keys` with four keys at the default `-L 4` gives 6 + 32 = 38 metadata columns on the default 20 x 10 scaler
    (careless/args/positional_encoding.py:24-37, args/scaling.py:21-31): past the lane kernel's 31 columns the first layer is peeled
    (csrc/elbo_peel.hip).  The run trains, validates, merges half datasets with the scaler frozen and writes every file; reloading its
    weights and freezing the scaler leaves the first layer's parameters untouched."""
    out = str(tmp_path / "a")
    flags = (f"mono --iterations={niter} --disable-progress-bar --test-fraction 0.2 --merge-half-datasets --mc-samples 3 "
             "--positional-encoding-keys X,Y,Hobs,Kobs dHKL,image_id,X,Y,Hobs,Kobs")
    args, model, hist = _run(flags, [PYP], out, False)
    eng = model._engine
    assert eng.peel and eng.d == 38 and "elbo_lane_kernel" in eng.kernel_name()
    assert "NLL_val" in hist and np.all(np.isfinite(hist["NLL_val"])) and os.path.exists(out + "_xval_0.mtz")
    flat = lambda m: m._engine.mlp.flat.cpu().numpy()
    w0 = flat(model)[: 10 * 38]
    assert np.abs(w0 - np.eye(10, 38).reshape(-1)).max() > 0                  # the peeled layer's kernel was trained (identity-initialised, nn.py:62-67)
    out2 = str(tmp_path / "b")
    flags2 = (f"mono --iterations=3 --disable-progress-bar --scale-file {out}_scale --structure-factor-file {out}_structure_factor --freeze-scales "
              "--mc-samples 3 --positional-encoding-keys X,Y,Hobs,Kobs dHKL,image_id,X,Y,Hobs,Kobs")
    _, model2, _ = _run(flags2, [PYP], out2, False)
    assert np.array_equal(flat(model2), flat(model))


@pytest.mark.parametrize("mode", ["mono", "poly"])
def test_default_scaler_with_image_layers_and_half_datasets(tmp_path, mode):
    """`careless mono|poly --image-layers 2 --merge-half-datasets` with the default scaler (reference careless/args/scaling.py:21-40,
    careless.py:102-128): the main training runs the lane kernel's per-image-layer instance (round 5), the half-dataset trainings -- scaler
    frozen -- the sampling / likelihood kernels only; every output file is written."""
    out = str(tmp_path / "a")
    flags = f"{mode} --iterations={niter} --disable-progress-bar --image-layers 2 --merge-half-datasets --test-fraction 0.1 dHKL,image_id"
    args, model, hist = _run(flags, [PYP], out, False)
    eng = model._engine
    name = eng.kernel_name()
    assert name.startswith("elbo_lane_kernel<10, 8, true, ") and name.rstrip().endswith("2> (image layers)"), name
    assert not eng.scaler_frozen and eng.obs.row_map is not None                   # trained on the packed-by-image layout
    x = read_mtz(out + "_xval_0.mtz")
    assert set(np.unique(x.columns["half"])) == {0.0, 1.0} and np.all(np.isfinite(x.columns["F"]))
    assert "NLL_val" in hist and np.all(np.isfinite(hist["NLL_val"]))
