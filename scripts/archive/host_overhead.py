import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from careless_amd.workloads import make_workload
model, inputs, data, spec = make_workload("mono_10M_studentt_posenc_5x64_S8", N=2000)
eng = model.engine(inputs)
eng.alloc_history(2000)
for i in range(50): eng.train_step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50, 1050): eng.train_step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e6*(t1-t0)/1000:.1f} us; with final sync {1e6*(t2-t0)/1000:.1f} us")
