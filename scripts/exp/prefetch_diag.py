"""Where does the host time of the prefetching loop go?  (diagnostic, GPU)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from bert_multimodal_transformer_amd import (AdamW, BertConfig, MAG_BertForSequenceClassification, MultimodalConfig,
                                             get_linear_schedule_with_warmup)
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
from bert_multimodal_transformer_amd import prefetch as PF

B, L, V, A = 48, 50, 47, 74
torch.manual_seed(1)
model = MAG_BertForSequenceClassification(BertConfig(num_labels=1), MultimodalConfig(1.0, 0.5), compute_dtype=torch.bfloat16).train()
opt = AdamW(optimizer_grouped_parameters(model), lr=1e-5)
sch = get_linear_schedule_with_warmup(opt, 100, 1000)
batches = bench.make_batches(8, B, L, V, A, seed=1)
dev = torch.device("cuda:0")
resident = [tuple(t.to(dev) for t in b) for b in batches]
mode = sys.argv[1] if len(sys.argv) > 1 else "all"

def loop(name, n, graph, pre):
    with model.stream_scope():
        src = PF.PinnedBatchRing((batches[i % 8] for i in range(n)), dev) if pre else (resident[i % 8] for i in range(n))
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        marks = []
        it = iter(src)
        for i in range(n):
            t0 = time.perf_counter()
            b = next(it)
            t1 = time.perf_counter()
            model.train_step(*b, optimizer=opt, graph=graph)
            t2 = time.perf_counter()
            sch.step()
            marks.append((t1 - t0, t2 - t1))
        t_host = time.perf_counter() - ts[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - ts[0]
    m = np.array(marks) * 1e3
    print("%-28s %6.3f ms/step  host %6.3f ms/step | next(batch): mean %.3f max %.3f | train_step: mean %.3f max %.3f" %
          (name, dt / n * 1e3, t_host / n * 1e3, m[:, 0].mean(), m[:, 0].max(), m[:, 1].mean(), m[:, 1].max()), flush=True)

for rep in range(1):
    loop("python-driven, resident", 30, False, False)
    loop("single call, resident", 30, None, False)
    loop("graph, resident", 30, True, False)
    loop("python-driven, pinned ring", 30, False, True)
    loop("single call, pinned ring", 30, None, True)
    loop("graph, pinned ring", 30, True, True)

