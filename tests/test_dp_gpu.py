"""GPU: the data-parallel path end to end on ONE device.  Two ranks share cuda:0 and use the gloo backend (RCCL refuses
two ranks on one GPU), which exercises exactly the product code that runs under RCCL on 8 GPUs -- DataParallel stage
hooks, side-stream all-reduce of flat gradient ranges, 1/world folded into AdamW -- and checks that two ranks with half
the batch each end up with the same parameters as one process with the whole batch (fp32 parity mode, dropout off)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from test_model_gpu import build, tb, weights, DEV
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.distributed import DataParallel
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
m = build(layers=2, p_mag=0.0, hidden_p=0.0, attn_p=0.0).train()
opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
sch = get_linear_schedule_with_warmup(opt, 0, 100)
if world > 1:
    dp = DataParallel(m, opt)
    dp.broadcast_parameters(0)
for s in range(2):
    b = weights.synthetic_bert_batch(8, 50, 47, 74, seed=90 + s)
    ids, vis, aco, mask, seg, lab = tb(b, DEV)
    lo, hi = (0, 8) if world == 1 else (rank * 4, rank * 4 + 4)
    m.training_step(ids[lo:hi], vis[lo:hi], aco[lo:hi], mask[lo:hi], seg[lo:hi], lab[lo:hi])
    opt.step(); sch.step(); opt.zero_grad()
torch.cuda.synchronize()
torch.save(m.flat_params.cpu(), os.environ["OUT"] + ".%d.%d" % (world, rank))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("OK", world, rank)
'''


def test_two_ranks_equal_one_process(tmp_path):
    import torch
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    out = str(tmp_path / "params")
    port = 29600 + os.getpid() % 1000

    def launch(world):
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       REPO_ROOT=ROOT, OUT=out)
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0, o.decode()[-3000:]

    launch(1)
    launch(2)
    ref = torch.load(out + ".1.0")
    a, b = torch.load(out + ".2.0"), torch.load(out + ".2.1")
    assert torch.equal(a, b)                                   # replicas stay in lock-step
    d = (a - ref).abs()
    frac = float((d > 2e-6).float().mean())
    print("DP(2 x 4) vs single(8): max |dparam| %.3e, moved fraction %.3e" % (float(d.max()), frac))
    # identical up to the summation order of the two half-batch gradients (Adam amplifies ~0 gradients to +-lr)
    assert float(d.max()) <= 2 * 1e-3 * 1.1 and frac < 2e-2
