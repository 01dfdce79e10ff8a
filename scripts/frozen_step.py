"""Step time of a training whose scaling model is frozen (--freeze-scales; the half-dataset trainings of --merge-half-datasets) against the
same step with the scaler trainable, on the bench workloads' data (one MI355X): the frozen step takes (loc, sigma) of every observation
once and then runs only the sampling / likelihood kernels (ElboEngine._data_term_frozen).  Usage: frozen_step.py [WORKLOAD ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from careless_amd.workloads import make_workload
names = sys.argv[1:] or ["mono_10M_cli_default_20x10_S1", "mono_10M_studentt_posenc_5x64_S8", "laue_5M_normal_5x64_S1", "mono_10M_20x10_img2_S1"]
for name in names:
    row = [name]
    for frozen in (False, True):
        model, inputs, data, spec = make_workload(name)
        model.scaling_model.trainable = not frozen
        eng = model.engine(inputs)
        eng.alloc_history(70)
        for i in range(10):
            eng.train_step(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(10, 60):
            eng.train_step(i)
        torch.cuda.synchronize()
        row.append(1e3 * (time.perf_counter() - t) / 50)
        del eng, model
        torch.cuda.empty_cache()
    print("%-40s trainable %.3f ms per step, frozen %.3f ms per step (%.1f x)" % (row[0], row[1], row[2], row[1] / row[2]), flush=True)
