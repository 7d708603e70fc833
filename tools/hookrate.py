import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch; torch.cuda.init()
from openzl_amd import Backend
from openzl_amd.backend import hook_fq_mul_rate
be = Backend(0)
for it in (12000, 50000):
    print("torch" if "--torch" in sys.argv else "no torch", "hook live-data rate at 3 waves/SIMD, iters", it, [round(hook_fq_mul_rate(be, 3, it), 2) for _ in range(4)], flush=True)
