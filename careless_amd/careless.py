"""`run_careless(parser)` -- the driver around the ELBO path with the reference's sequence of steps and output files
(reference careless/careless.py:11-128): format the reflection files, optional train/test split, build the model, optional
weight loading / freezing, train, write `<out>_<i>.mtz`, `<out>_history.csv`, `<out>_structure_factor`, `<out>_scale`,
`<out>_predictions_<i>.mtz` and, with --merge-half-datasets, `<out>_xval_<i>.mtz`.
Inputs: `.mtz` reflection files (formatted here) or ONE pre-formatted `.npz` written by `careless_amd.io.formats.save_inputs_npz`."""
from __future__ import annotations

import numpy as np


def _format(parser):
    from careless_amd.io.formats import load_inputs_npz
    from careless_amd.io.formatter import LaueFormatter, MonoFormatter
    files = list(parser.reflection_files)
    if len(files) == 1 and files[0].endswith(".npz"):
        return load_inputs_npz(files[0])
    fmt = LaueFormatter.from_parser(parser) if parser.type == "poly" else MonoFormatter.from_parser(parser)
    return fmt.format_files(files)


def _format_once_per_node(parser, rank: int, world: int):
    """Data-parallel launch: ONE rank per node -- local rank 0 -- reads and formats the reflection files (MTZ parsing, ASU mapping,
    metadata standardisation: host work of the order of the file size) and leaves the formatted arrays as `.npy` files in its node's
    /dev/shm; after a flag all-reduce (so that a formatting error ends every rank instead of leaving them in a barrier) the ranks
    map the files read-only -- a node holds ONE copy of the inputs, the engine of a rank copies only its shard's rows
    (engine.ObsData) -- and, after a second flag all-reduce over the loads (a rank whose pickle / map fails must not leave the
    others in a barrier), the formatting rank removes them (the maps keep the pages alive).  The ASU collection (small) travels as a
    pickle.  The node is told apart by LOCAL_RANK / LOCAL_WORLD_SIZE as `torch.distributed.run` sets them (absent: one node);
    formatting is deterministic, so every node's copy is the same.  The directory also goes away if the formatting rank dies
    in between (atexit) -- short of a SIGKILL."""
    if world <= 1:
        return _format(parser)
    import atexit
    import os
    import pickle
    import shutil
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if not (0 <= local_rank < local_world <= world) or world % local_world != 0:
        raise RuntimeError(f"careless_amd: inconsistent launch environment (RANK {rank}, WORLD_SIZE {world}, LOCAL_RANK {local_rank}, "
                           f"LOCAL_WORLD_SIZE {local_world})")
    formats = local_rank == 0
    node = rank // local_world                                  # (in the name only so that a one-machine rehearsal of two "nodes" works)
    path = f"/dev/shm/careless_amd_{os.environ.get('MASTER_ADDR', 'local')}_{os.environ.get('MASTER_PORT', '0')}_n{node}"
    cleanup = lambda: shutil.rmtree(path, ignore_errors=True)
    err = None
    if formats:
        atexit.register(cleanup)
        try:
            cleanup()
            os.makedirs(path)
            inputs, rac = _format(parser)
            for i, a in enumerate(inputs):
                np.save(os.path.join(path, f"input_{i:02d}.npy"), np.ascontiguousarray(a))
            with open(os.path.join(path, "rac.pickle"), "wb") as f:
                pickle.dump((len(inputs), rac), f)
        except Exception as e:                                  # noqa: BLE001  (every rank raises below)
            err = e
    try:
        if not _all_ranks_ok(err is None, world):
            raise err if err is not None else RuntimeError("careless_amd: a rank could not format the reflection files (see its traceback)")
        inputs = rac = None
        try:
            with open(os.path.join(path, "rac.pickle"), "rb") as f:
                n, rac = pickle.load(f)
            inputs = tuple(np.load(os.path.join(path, f"input_{i:02d}.npy"), mmap_mode="r") for i in range(n))
        except Exception as e:                                  # noqa: BLE001
            err = e
        if not _all_ranks_ok(err is None, world):               # (also the barrier before the files go away)
            raise err if err is not None else RuntimeError("careless_amd: a rank could not map the formatted inputs (see its traceback)")
    finally:
        if formats:
            cleanup()
            atexit.unregister(cleanup)
    return inputs, rac


def _prediction_tables(dm, model, inputs, test_value):
    """Per-ASU prediction tables (reference manager.py:89-161): one row per observation (per harmonic group for Laue data)."""
    from careless_amd.models.base import BaseModel
    rac = dm.asu_collection
    laue = BaseModel.is_laue(inputs)
    refl_id = np.asarray(BaseModel.get_refl_id(inputs)).reshape(-1)
    asu_id, H = rac.to_asu_id_and_miller_index(refl_id)
    hid = np.asarray(BaseModel.get_harmonic_id(inputs)).reshape(-1) if laue else np.arange(len(refl_id))
    _, idx = np.unique(hid, return_index=True)
    n = len(idx)
    pred = dm.get_predictions(model, inputs)
    cols = {"H": H[idx, 0], "K": H[idx, 1], "L": H[idx, 2], "asu_id": asu_id[idx],
            "image_id": np.asarray(BaseModel.get_image_id(inputs)).reshape(-1)[idx],
            "file_id": np.asarray(BaseModel.get_file_id(inputs)).reshape(-1)[idx], "test": np.full(n, test_value),
            "Iobs": np.asarray(BaseModel.get_intensities(inputs)).reshape(-1)[:n],
            "SigIobs": np.asarray(BaseModel.get_uncertainties(inputs)).reshape(-1)[:n],
            "Ipred": pred["Ipred"][:n], "SigIpred": pred["SigIpred"][:n], "Scale": pred["Scale"][:n], "SigScale": pred["SigScale"][:n]}
    return [{k: np.asarray(v)[cols["asu_id"] == i] for k, v in cols.items()} for i in range(len(rac))]


def _data_parallel():
    """(rank, world) of a one-process-per-GPU launch (`python -m torch.distributed.run --nproc-per-node N -m careless_amd mono ...`):
    RANK / LOCAL_RANK / WORLD_SIZE in the environment select the data-parallel engine -- observations sharded over the ranks,
    one all-reduce of the flat gradient per step (careless_amd/distributed.py).  The reference has no counterpart (it pins one
    GPU, careless/parser.py:26-40).  Rank 0 formats the files once for the node (`_format_once_per_node`), every rank draws the same
    splits from the shared arrays; rank 0 writes the outputs."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CARELESS_FORCE_DIST=1: a one-rank run still initialises the process group and all-reduces its gradient -- RCCL with nobody to talk
    # to, but every call of the multi-GPU step is made (tests/test_rccl.py runs it on the one-GPU box the driver has)
    if world <= 1 and os.environ.get("CARELESS_FORCE_DIST", "0") != "1":
        return 0, 1
    world = max(world, 1)
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    global _CREATED_GROUP
    if not dist.is_initialized():
        _CREATED_GROUP = True                   # (an embedding caller's own process group is left alone at the end)
        backend = os.environ.get("CARELESS_DIST_BACKEND", "nccl")        # nccl = RCCL on ROCm; gloo rehearses on a single GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


_CREATED_GROUP = False


def _all_ranks_ok(ok: bool, world: int) -> bool:
    """Rank 0's output step has no collective of its own: all ranks learn whether it went through (one MIN all-reduce of a flag, as
    bench.py's `agree`), so a failed write ends every rank with an error instead of leaving the others in a barrier until the RCCL
    watchdog fires."""
    if world <= 1:
        return ok
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def _leave_data_parallel(world: int) -> None:
    """End of a one-process-per-GPU run: tear down the process group this module created (RCCL otherwise warns -- or hangs -- at
    interpreter exit); a group the embedding caller had initialised is the caller's to destroy."""
    global _CREATED_GROUP
    if world <= 1 and not _CREATED_GROUP:
        return
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        if _CREATED_GROUP:
            dist.destroy_process_group()
            _CREATED_GROUP = False


def run_careless(parser):
    from careless_amd.io.formats import PREDICTION_TYPES, results_tables, write_history_csv, write_table_mtz
    from careless_amd.manager import DataManager

    rank, world = _data_parallel()
    np.random.seed(parser.seed)                                    # reference parser.py:22-23
    inputs, rac = _format_once_per_node(parser, rank, world)
    dm = DataManager(inputs, rac, parser=parser)
    if parser.test_fraction is not None:
        train, test = dm.split_data_by_refl(parser.test_fraction)
    else:
        train, test = dm.inputs, None

    model = dm.build_model()
    if world > 1:
        model.set_data_parallel(rank, world)
    if parser.scale_file is not None:
        model.scaling_model.load_weights(parser.scale_file)
    if parser.freeze_scales:
        model.scaling_model.trainable = False
    if parser.structure_factor_file is not None:
        model.surrogate_posterior.load_weights(parser.structure_factor_file)
    if parser.freeze_structure_factors:
        model.surrogate_posterior.trainable = False

    progress = not parser.disable_progress_bar and rank == 0
    history = model.train_model(train, parser.iterations, message="Training", validation_data=test,
                                validation_frequency=parser.validation_frequency, progress=progress)

    asus = list(rac)
    # The output step has no collective and the parameters are identical on every rank: rank 0 alone computes and writes it (the
    # other ranks skip the result tables and the full-size prediction pass instead of computing them for no-op writers)
    def output_step():
        for i, table in enumerate(results_tables(dm.get_results(model.surrogate_posterior, inputs=train), rac)):
            write_table_mtz(parser.output_base + f"_{i}.mtz", table, asus[i])
        write_history_csv(parser.output_base + "_history.csv", history)
        model.surrogate_posterior.save_weights(parser.output_base + "_structure_factor")
        model.scaling_model.save_weights(parser.output_base + "_scale")
        if getattr(parser, "save_data_manager", False):             # reference careless.py:81-84
            import pickle
            with open(parser.output_base + "_data_manager.pickle", "wb") as out:
                pickle.dump(dm, out)

        tables = _prediction_tables(dm, model, train, 0)
        if test is not None:
            tables = [{k: np.concatenate([a[k], b[k]]) for k in a} for a, b in zip(tables, _prediction_tables(dm, model, test, 1))]
        for i, table in enumerate(tables):
            write_table_mtz(parser.output_base + f"_predictions_{i}.mtz", table, asus[i], PREDICTION_TYPES)

    out_err = None
    if rank == 0:
        try:
            output_step()
        except Exception as e:                                  # noqa: BLE001  (reported on every rank below)
            out_err = e
    if not _all_ranks_ok(out_err is None, world):
        _leave_data_parallel(world)
        if out_err is not None:
            raise out_err
        raise RuntimeError("careless_amd: rank 0 failed in the output step (see its traceback); nothing was left waiting")

    if parser.merge_half_datasets:
        scaling_model = model.scaling_model
        scaling_model.trainable = False
        xval = [None] * len(asus)
        for repeat in range(parser.half_dataset_repeats):
            for half_id, half in enumerate(dm.split_data_by_image()):
                m = dm.build_model(scaling_model=scaling_model)
                if world > 1:
                    m.set_data_parallel(rank, world)
                m.train_model(half, parser.iterations, message=f"Merging repeat {repeat + 1} half {half_id + 1}", progress=progress)
                if rank != 0:                                       # (the trainings are collective; the tables are rank 0's)
                    continue
                for file_id, t in enumerate(results_tables(dm.get_results(m.surrogate_posterior, inputs=half), rac)):
                    t["repeat"] = np.full(len(t["H"]), repeat)
                    t["half"] = np.full(len(t["H"]), half_id)
                    xval[file_id] = t if xval[file_id] is None else {k: np.concatenate([xval[file_id][k], t[k]]) for k in t}
        for file_id, t in enumerate(xval if rank == 0 else []):
            types = {"H": "H", "K": "H", "L": "H", "F": "F", "SigF": "Q", "I": "J", "SigI": "Q", "N": "I", "repeat": "I", "half": "I"}
            write_table_mtz(parser.output_base + f"_xval_{file_id}.mtz", t, asus[file_id], types)
    _leave_data_parallel(world)
    return model, history


def main(argv=None):
    from careless_amd.parser import parser
    run_careless(parser.parse_args(argv))


if __name__ == "__main__":
    main()
