"""where does MB_ADAMW_RIDE=1 differ from the plain step? (diagnostic for tests/test_model_gpu.py::test_adamw_riding...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MB_DETERMINISTIC"] = "1"
os.environ["MB_GROUP_WGRAD"] = sys.argv[1] if len(sys.argv) > 1 else "128"
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_model_gpu as T
cdt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.environ["MB_ADAMW_RIDE"] = "0"
ref = T._trajectory(cdt, True, nsteps=nsteps)
os.environ["MB_ADAMW_RIDE"] = "1"
ride = T._trajectory(cdt, True, nsteps=nsteps)
for k in ("p", "m", "v", "shadow", "g"):
    a, b = ride[k].float(), ref[k].float()
    d = (a - b).abs()
    nz = torch.nonzero(d).flatten()
    print(k, "n", a.numel(), "ndiff", nz.numel(), "max", float(d.max()), "first", int(nz[0]) if nz.numel() else -1, "last", int(nz[-1]) if nz.numel() else -1)
    if nz.numel():
        i = int(nz[0])
        print("   ride", a[i:i + 4].tolist(), "ref", b[i:i + 4].tolist())
