import sys, time, ctypes as C
sys.path.insert(0,'/root/repo')
from lasso_amd import HostProver, _abi
hp = HostProver()
s=1<<24; idx = hp.gen_indices(s, 1<<16, 1)
gens = hp.gens(1, s, 1, 16)
for i in range(3):
    t0=time.time(); dense = hp.densify(idx, 16); t1=time.time(); comm = hp.commit(dense, gens); t2=time.time()
    print("densify %.1f ms commit %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3))
    hp.free(dense, None)
