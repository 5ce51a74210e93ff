"""Which part of PinnedBatchRing._pack stalls while steps are in flight?  (diagnostic, GPU)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from bert_multimodal_transformer_amd import (AdamW, BertConfig, MAG_BertForSequenceClassification, MultimodalConfig)
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
from bert_multimodal_transformer_amd import prefetch as PF

B, L, V, A = 48, 50, 47, 74
torch.manual_seed(1)
model = MAG_BertForSequenceClassification(BertConfig(num_labels=1), MultimodalConfig(1.0, 0.5), compute_dtype=torch.bfloat16).train()
opt = AdamW(optimizer_grouped_parameters(model), lr=1e-5)
batches = bench.make_batches(8, B, L, V, A, seed=1)
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
if variant == "threads1":
    torch.set_num_threads(1)
log = []
def timed_pack(self, batch):
    T = [time.perf_counter()]
    batch = list(batch)
    offs, off = [], 0
    for i, t in enumerate(batch):
        dt = self.DTYPES[i]
        offs.append((off, dt))
        off = (off + t.numel() * t.element_size() + 255) // 256 * 256
    nbytes = max(off, 256)
    k = self._next
    self._next = (self._next + 1) % self.nblocks
    blk = self.blocks[k]
    if blk is None:
        blk = self.blocks[k] = PF._Block(nbytes)
    T.append(time.perf_counter())
    if blk.busy:
        if variant == "query":
            while not blk.done.query():
                time.sleep(0.0002)
        else:
            blk.done.synchronize()
        blk.busy = False
    T.append(time.perf_counter())
    views = []
    for t, (o, dt) in zip(batch, offs):
        n = t.numel() * t.element_size()
        v = blk.host[o: o + n].view(dt).view(t.shape)
        if variant == "numpy":
            np.copyto(v.numpy(), t.numpy())
        else:
            v.copy_(t)
        views.append(v)
    T.append(time.perf_counter())
    log.append([round((b - a) * 1e3, 3) for a, b in zip(T, T[1:])])
    return blk, tuple(views)
PF.PinnedBatchRing._pack = timed_pack
n = 40
with model.stream_scope():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in PF.PinnedBatchRing((batches[i % 8] for i in range(n)), dev):
        model.train_step(*b, optimizer=opt)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("variant %s: %.3f ms/step host %.3f ms/step" % (variant, dt / n * 1e3, th / n * 1e3))
a = np.array(log)
print("[alloc, sync, hostcopy] mean", a.mean(0).round(3), "max", a.max(0).round(3))
for i, r in enumerate(log):
    if sum(r) > 2:
        print(i, r)
