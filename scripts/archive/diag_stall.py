#!/usr/bin/env python3
"""Which step of a short row-split run stalls, and on which side (host enqueue vs device): per-step host timestamps and device events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ["CARELESS_HIP_OWNER_SHARD"] = os.environ.get("CARELESS_HIP_OWNER_SHARD", "0")
import torch, torch.distributed as dist
from careless_amd.workloads import make_workload
steps, warmup = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 5
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model, inputs, data, spec = make_workload("mono_10M_cli_default_20x10_S1")
model.set_data_parallel(0, 8)
eng = model.engine(inputs)
eng.force_allreduce = True
eng.alloc_history(warmup + steps)
for i in range(warmup):
    eng.train_step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
ev[0].record()
t0 = time.perf_counter()
for i in range(steps):
    eng.train_step(warmup + i)
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
total = time.perf_counter() - t0
dev = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
hd = [1e3 * (host[i] - (host[i - 1] if i else 0.0)) for i in range(steps)]
print(f"steps {steps}: wall {1e3 * total:.2f} ms = {1e3 * total / steps:.4f} ms/step; device sum {sum(dev):.2f} ms")
print("device ms per step:", " ".join(f"{d:.2f}" for d in dev))
print("host enqueue ms per step:", " ".join(f"{h:.2f}" for h in hd))
dist.destroy_process_group()
