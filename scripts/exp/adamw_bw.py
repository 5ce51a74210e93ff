"""AdamW kernel bandwidth on the flat buffers of the bench model (110.85 M params): ms per full update and TB/s."""
import sys, torch
sys.path.insert(0, ".")
from bert_multimodal_transformer_amd import _lib
L = _lib.lib()
n = 110853184
dev = "cuda:0"
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(L.mb_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, n, 0, 85524480, 1e-5, 0.9, 0.999,
                               1e-6, 0.01, 3, 1, 1.0, 1, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byts = n * 32 + 85524480 * 2
print("adamw full update: %.3f ms, %.2f TB/s (%.0f MB)" % (ms, byts / ms / 1e9, byts / 1e6))
