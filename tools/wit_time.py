import time, sys
sys.path.insert(0,'/root/repo')
from openzl_amd.backend import Circuit
from openzl_amd import ZL_BLS12_381, ZL_BN254
for curve in (ZL_BLS12_381, ZL_BN254):
    for rep in range(3):
        t0=time.perf_counter(); c=Circuit(curve, 4096, x0=5+rep, x1=7, witness_only=True); dt=time.perf_counter()-t0; c.close()
        print(curve, 'witness-only synthesis k=4096: %.1f ms'%(dt*1e3))
t0=time.perf_counter(); c=Circuit(ZL_BLS12_381, 4096); print('full synthesis %.2f s'%(time.perf_counter()-t0)); c.close()
