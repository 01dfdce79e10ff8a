import os,sys,time,torch,numpy as np
sys.path.insert(0,".")
from careless_amd.workloads import build_model, reference_inputs
from careless_amd.synthetic import make_synthetic
N=4000000
data=make_synthetic(N, d0=5, posenc=False, outliers=True)
for S,dof in [(int(x), 16.0) for x in os.environ.get("SAMPLES", "1,2,3,4,6,8").split(",")]:
    model=build_model(data, 20, 10, S, dof=dof)
    eng=model.engine(reference_inputs(data)); eng.alloc_history(30)
    for i in range(5): eng.train_step(i)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(20): eng.train_step(5+i)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/20
    print("mono 4M 20x10 S=%d %s: %.3f ms/step %.3e refl/s LANE=%s NARROW=%s"%(S, "studentt" if dof else "normal", 1e3*t, N/t, os.environ.get("CARELESS_HIP_LANE","1"), os.environ.get("CARELESS_HIP_NARROW","1")))
    del eng, model
