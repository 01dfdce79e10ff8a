import numpy as np, time, sys
sys.path.insert(0,'/root/repo')
from careless_amd.io.mtz import write_mtz
n=int(sys.argv[1]); rng=np.random.default_rng(0)
cell=(79.,79.,38.,90.,90.,90.)
# P43212 (#96) HEWL-like: hkl up to +-40, unmerged, full sphere
H=rng.integers(-40,41,size=(n,3))
H=H[(np.abs(H).sum(1)>0)]
n=len(H)
batch=np.sort(rng.integers(1,10001,size=n))
I=rng.gamma(1.0,100.0,size=n); sig=np.sqrt(I)+5
cols={"H":H[:,0],"K":H[:,1],"L":H[:,2],"BATCH":batch,"I":I,"SIGI":sig,"XDET":rng.uniform(0,2000,n),"YDET":rng.uniform(0,2000,n)}
types={"H":"H","K":"H","L":"H","BATCH":"B","I":"J","SIGI":"Q","XDET":"R","YDET":"R"}
from careless_amd.io.spacegroups import *
t=time.time()
write_mtz(sys.argv[2],cols,types,cell,spacegroup_name="P 43 21 2",spacegroup_number=96,symops=["X, Y, Z","-X, -Y, Z+1/2","-Y+1/2, X+1/2, Z+3/4","Y+1/2, -X+1/2, Z+1/4","-X+1/2, Y+1/2, -Z+3/4","X+1/2, -Y+1/2, -Z+1/4","Y, X, -Z","-Y, -X, -Z+1/2"])
print("wrote",n,"rows in",time.time()-t)
