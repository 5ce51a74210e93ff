"""GPU: the data-parallel path end to end on ONE device.  Two ranks share cuda:0 and use the gloo backend (RCCL refuses
two ranks on one GPU), which exercises exactly the product code that runs under RCCL on 8 GPUs -- DataParallel stage
hooks, side-stream all-reduce of flat gradient ranges, 1/world folded into AdamW -- and checks that two ranks with half
the batch each end up with the same parameters as one process with the whole batch (fp32 parity mode, dropout off)."""
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
kind = os.environ.get("KIND", "bert")
if kind == "xlnet":
    from test_xlnet_gpu import build, tb, weights, DEV
    make = lambda: build(layers=2, p_mag=0.0, p=0.0)
    batch = lambda seed: weights.synthetic_xlnet_batch(8, 50, 47, 74, seed=seed)
else:
    from test_model_gpu import build, tb, weights, DEV
    make = lambda: build(layers=2, p_mag=0.0, hidden_p=0.0, attn_p=0.0)
    batch = lambda seed: weights.synthetic_bert_batch(8, 50, 47, 74, seed=seed)
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.distributed import DataParallel
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
m = make().train()
opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
sch = get_linear_schedule_with_warmup(opt, 0, 100)
dp = None
if world > 1:
    dp = DataParallel(m, opt)
    dp.broadcast_parameters(0)
grads = []
for s in range(2):
    ids, vis, aco, mask, seg, lab = tb(batch(90 + s), DEV)
    lo, hi = (0, 8) if world == 1 else (rank * 4, rank * 4 + 4)
    m.training_step(ids[lo:hi], vis[lo:hi], aco[lo:hi], mask[lo:hi], seg[lo:hi], lab[lo:hi])
    if dp is not None:
        dp.finish()                                  # every piece of the exchange has landed
    torch.cuda.synchronize()
    grads.append((m.flat_grads * (1.0 / world)).cpu())      # the mean over the global batch, as AdamW will see it
    opt.step(); sch.step(); opt.zero_grad()
torch.cuda.synchronize()
torch.save(dict(p=m.flat_params.cpu(), g=grads, sparse=bool(dp is not None and dp.word is not None)), os.environ["OUT"] + ".%d.%d" % (world, rank))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("OK", world, rank)
'''


@pytest.mark.parametrize("kind,sparse", [("bert", "1"), ("bert", "0"), ("xlnet", "1"), ("bert-stage-graphs", "1")])
def test_two_ranks_equal_one_process(tmp_path, kind, sparse):
    """Two ranks with half the batch each vs one process with the whole batch (fp32 parity mode, dropout off).  The quantity that
    must agree is the ALL-REDUCED GRADIENT (the mean over the global batch): <= 5e-6 of the largest gradient everywhere, for the
    row-wise word-embedding exchange and for the dense one, for MAG-BERT and for MAG-XLNet (whose no-decay block round 1 reduced
    twice).  Parameters after AdamW are only bounded (Adam turns a ~0 gradient whose sign flips with the summation order into a
    +-lr move); replicas must stay bit-identical."""
    import torch
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    out = str(tmp_path / "params")
    port = free_port()

    graphs = kind.endswith("-stage-graphs")        # MB_DP_GRAPH=1: every pass / backward stage of the ranks is ONE replayed graph
    kind = kind.split("-")[0]

    def launch(world):
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       REPO_ROOT=ROOT, OUT=out, KIND=kind, MB_DP_SPARSE_EMB=sparse, MB_DP_GRAPH="1" if graphs else "0")
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0, o.decode()[-3000:]

    launch(1)
    launch(2)
    ref = torch.load(out + ".1.0")
    a, b = torch.load(out + ".2.0"), torch.load(out + ".2.1")
    assert a["sparse"] == (sparse == "1")
    assert torch.equal(a["p"], b["p"])                         # replicas stay in lock-step
    for s in range(2):
        assert torch.equal(a["g"][s], b["g"][s])               # ... and so do the reduced gradients
    g0, r0 = a["g"][0], ref["g"][0]                            # step 0: identical parameters on both sides
    gerr = float((g0 - r0).abs().max()) / float(r0.abs().max())
    print("%s DP(2 x 4) vs single(8): reduced gradient max |d| / max |g| = %.3e" % (kind, gerr))
    assert gerr <= 5e-6
    d = (a["p"] - ref["p"]).abs()
    print("max |dparam| %.3e, moved fraction %.3e" % (float(d.max()), float((d > 2e-6).float().mean())))
    assert float(d.max()) <= 2 * 1e-3 * 1.1


_DRIVER_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["REPO_ROOT"])
from bert_multimodal_transformer_amd import multimodal_driver as D
D.args = D.parse_args(["--synthetic", "240", "--n_epochs", "8", "--seed", "5", "--train_batch_size", "24", "--learning_rate", "2e-4",
                       "--model", os.environ["MODEL"], "--gradient_accumulation_step", os.environ["ACCUM"]])
D.set_random_seed(D.args.seed)
tr, dev, te, nsteps = D.set_up_data_loader()            # initialises the process group (MB_DIST_BACKEND=gloo here)
model, opt, sch = D.prep_for_training(nsteps)
l0 = D.train_epoch(model, tr, opt, sch)
for _ in range(4):
    l1 = D.train_epoch(model, tr, opt, sch)
vl = D.eval_epoch(model, dev, opt)
acc, mae, corr, f1 = D.test_score_model(model, te)
npred = len(D.test_epoch(model, te)[0])
torch.cuda.synchronize()
rank = int(os.environ["RANK"])
torch.save(dict(p=model.flat_params.cpu(), l0=l0, l1=l1, vl=vl, steps=len(tr), nsteps=nsteps, acc=float(acc), mae=float(mae), npred=npred,
                ntest=len(te.dataset)), os.environ["OUT"] + ".%d" % rank)
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("model,accum", [("bert-base-uncased", "1"), ("bert-base-uncased", "2"), ("xlnet-base-cased", "1")])
def test_driver_under_two_ranks(tmp_path, model, accum):
    """the bundled driver launched as two ranks (torchrun environment, gloo on one GPU): per-rank sharded batches, all-reduce
    hooked into the backward, replicas stay bit-identical, the loss goes down; also with gradient accumulation (no all-reduce on
    the non-final micro-step) and for MAG-XLNet"""
    import torch
    script = tmp_path / "d.py"
    script.write_text(_DRIVER_WORKER)
    out = str(tmp_path / "drv")
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO_ROOT=ROOT, OUT=out, MB_DIST_BACKEND="gloo", MODEL=model, ACCUM=accum)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=900)
        assert p.returncode == 0, o.decode()[-3000:]
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(a["p"], b["p"])                          # replicas in lock-step after 5 epochs
    assert a["steps"] == b["steps"] == 240 // 48                # 5 global batches of 2 x 24
    assert a["nsteps"] == int(240 / 48 / int(accum)) * 8
    print("losses rank0 %.4f -> %.4f, rank1 %.4f -> %.4f" % (a["l0"], a["l1"], b["l0"], b["l1"]))
    for d in (a, b):
        assert d["l0"] == d["l0"] and d["l1"] < d["l0"] and d["vl"] == d["vl"], (d["l0"], d["l1"], d["vl"])
    # sharded evaluation: both ranks report the same dev loss and the same test metrics, computed over EVERY test sample
    assert a["vl"] == b["vl"] and a["acc"] == b["acc"] and a["mae"] == b["mae"]
    assert a["npred"] == b["npred"] == a["ntest"]


_RCCL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
from test_model_gpu import build, tb, weights, DEV
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.distributed import DataParallel
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
torch.cuda.set_device(0)
use_dp = os.environ["USE_DP"] == "1"
if use_dp:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # nccl == RCCL on ROCm
m = build(layers=2, p_mag=0.0, hidden_p=0.0, attn_p=0.0, cdt=torch.bfloat16 if os.environ.get("CDT") == "bf16" else torch.float32).train()
opt = AdamW(optimizer_grouped_parameters(m), lr=1e-3)
sch = get_linear_schedule_with_warmup(opt, 0, 100)
if use_dp:
    dp = DataParallel(m, opt)
    assert dp.reducer.active
    assert dp.reducer.wire_dtype == (torch.bfloat16 if os.environ.get("CDT") == "bf16" else torch.float32)
    dp.broadcast_parameters(0)
with m.stream_scope():
    for s in range(3):
        ids, vis, aco, mask, seg, lab = tb(weights.synthetic_bert_batch(8, 50, 47, 74, seed=90 + s), DEV)
        m.training_step(ids, vis, aco, mask, seg, lab)
        opt.step(); sch.step(); opt.zero_grad()
torch.cuda.synchronize()
torch.save(m.flat_params.cpu(), os.environ["OUT"] + "." + os.environ["USE_DP"])
if use_dp:
    dist.barrier(); dist.destroy_process_group()
print("OK")
'''


@pytest.mark.parametrize("cdt", ["fp32", "bf16"])
def test_rccl_call_path_single_rank(tmp_path, cdt):
    """The RCCL (backend "nccl") call path of the product -- init with device_id, broadcast of the flat parameters, all-reduce
    of flat gradient views on the comm stream hooked into the backward stages, the wait before AdamW, barrier -- on ONE GPU
    with a 1-rank group (MB_DP_FORCE=1 issues the collectives although they are identities).  Result == plain single process."""
    import torch
    script = tmp_path / "r.py"
    script.write_text(_RCCL_WORKER)
    out = str(tmp_path / "rccl")
    for use_dp in ("0", "1"):
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                   REPO_ROOT=ROOT, OUT=out, USE_DP=use_dp, MB_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CDT=cdt,
                   MB_DP_GRAD_DTYPE=cdt)       # (the bf16 wire is the automatic choice of two-GPU groups only: asked for here)
        p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    d = (a - b).abs()
    frac = float((d > 2e-6).float().mean())
    print("RCCL 1-rank DP vs plain: max |dparam| %.3e, moved fraction %.3e" % (float(d.max()), frac))
    if cdt == "bf16":
        # bf16 perf mode: gradients travel as bf16 (rounded once per rank, 2^-9 relative) -> every Adam update moves by a fraction
        # of a percent of lr; bounded on average, and no element moves by more than the +-lr sign-flip bound per step
        print("mean |dparam| %.3e" % float(d.mean()))
        assert float(d.max()) <= 3 * 2 * 1e-3 * 1.1 and float(d.mean()) <= 2e-5
        return
    # same criterion as test_two_ranks_equal_one_process: Adam turns a ~0 gradient whose sign flips with the fp32 summation order
    # into a +-lr move; a missing stream dependency would instead corrupt whole contiguous ranges
    assert float(d.max()) <= 3 * 2 * 1e-3 * 1.1 and frac < 2e-2


# ------------------------------------------------------------------------------------------------ the exchange inside the engine call
_ENGINE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
kind = os.environ.get("KIND", "bert")
cdt = torch.bfloat16 if os.environ.get("CDT") == "bf16" else torch.float32
if kind == "xlnet":
    from test_xlnet_gpu import build, tb, weights, DEV
    make = lambda: build(layers=2, p_mag=0.0, p=0.0)
    batch = lambda seed: weights.synthetic_xlnet_batch(8, 50, 47, 74, seed=seed)
else:
    from test_model_gpu import build, tb, weights, DEV
    make = lambda: build(layers=int(os.environ.get("LAYERS", "2")), p_mag=0.0, hidden_p=0.0, attn_p=0.0, cdt=cdt)
    batch = lambda seed: weights.synthetic_bert_batch(8, 50, 47, 74, seed=seed)
from bert_multimodal_transformer_amd import AdamW, get_linear_schedule_with_warmup
from bert_multimodal_transformer_amd.distributed import DataParallel
from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("BACKEND", "gloo")
use_dp = os.environ.get("USE_DP", "1") == "1" and (world > 1 or os.environ.get("MB_DP_FORCE") == "1")
torch.cuda.set_device(0)
if use_dp:
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
m = make().train()
opt = AdamW(optimizer_grouped_parameters(m), lr=float(os.environ.get("LR", "1e-3")))
sch = get_linear_schedule_with_warmup(opt, 0, 100)
dp = None
if use_dp:
    dp = DataParallel(m, opt)
    dp.broadcast_parameters(0)
first_m = None
every_m = []
accum = int(os.environ.get("ACCUM", "1")) if use_dp else 1
micro_fused = []
with m.stream_scope():
    for s in range(int(os.environ.get("STEPS", "2"))):
        ids, vis, aco, mask, seg, lab = tb(batch(90 + s), DEV)
        lo, hi = (0, 8) if world == 1 else (rank * 4, rank * 4 + 4)
        n = (hi - lo) // accum
        for k in range(accum):               # a gradient-accumulation window: micro-steps without the optimizer, then the update
            a0, a1, last = lo + k * n, lo + (k + 1) * n, k == accum - 1
            if dp is not None:
                dp.sync = last
            m.train_step(ids[a0:a1], vis[a0:a1], aco[a0:a1], mask[a0:a1], seg[a0:a1], lab[a0:a1], optimizer=opt if last else None,
                         loss_scale=1.0 / accum, graph=None if os.environ.get("GRAPH", "1") == "1" else "launches")
            if dp is not None and not last:
                micro_fused.append(bool(dp._last_fused))
        sch.step()
        if first_m is None:
            torch.cuda.synchronize()
            first_m = m._core._adam_m.cpu().clone()        # (1 - beta1) x the mean gradient over the global batch: what the exchange delivered
        if os.environ.get("EVERY_M") == "1":
            torch.cuda.synchronize()
            every_m.append(m._core._adam_m.cpu().clone())
torch.cuda.synchronize()
fused = bool(dp is not None and dp._last_fused)
stats = dp.comm.stats() if fused else (0, 0)
shards = dp.shards.bounds if (dp is not None and dp.shards is not None) else None
slices = dp.comm.shard_slices() if (fused and dp.comm.sharding) else None          # the sharded update INSIDE the engine call
if fused and dp.comm.sharding:
    m._core._comm_join()
    torch.cuda.synchronize()
p_before = m.flat_params.cpu().clone()
m_before = m._core._adam_m.cpu().clone()
sh = m._core.shadow
shadow = sh[m._core.sh_begin: m._core.sh_end].float().cpu() if sh.numel() > 1 else None
if os.environ.get("GATHER_MASTERS") == "1":
    if shards is not None:
        dp.shards.gather_masters()
    elif slices is not None:
        m._core.refresh_sharded_state()
    torch.cuda.synchronize()
torch.save(dict(p=m.flat_params.cpu(), m=first_m, fused=fused, stats=stats, sparse=bool(fused and dp.comm.sparse), shards=shards,
                p_before_gather=p_before, shadow=shadow, every_m=every_m, slices=slices, micro_fused=micro_fused,
                adam_m=m._core._adam_m.cpu(), adam_m_before_gather=m_before),
           os.environ["OUT"] + ".%d.%d.%s" % (world, rank, os.environ.get("USE_DP", "1")))
if use_dp:
    dist.barrier(); dist.destroy_process_group()
print("OK", world, rank)
"""


def _run_engine_workers(tmp_path, world, env_extra, tag="w"):
    script = tmp_path / (tag + ".py")
    script.write_text(_ENGINE_WORKER)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REPO_ROOT=ROOT,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]


@pytest.mark.parametrize("kind,sparse,chunk", [("bert", "1", "2"), ("bert", "0", "1"), ("xlnet", "1", "2")])
def test_single_call_dp_step_two_ranks_equal_one_process(tmp_path, kind, sparse, chunk):
    """mb_*_train_step_dp -- ONE engine call per optimizer step with the gradient exchange issued from C between the graphs of the
    step -- as two ranks with half the batch each (gloo through the callback backend: two ranks share this GPU) vs the plain
    single-call step over the whole batch.  fp32, dropout off.  What the exchange delivered is read back from Adam's first moment
    after step 1 (m = 0.1 x the mean gradient over the global batch): <= 5e-6 of its largest element everywhere, row-wise and dense
    word-embedding exchange, one or two layers per piece; the replicas stay bit-identical."""
    import torch
    out = str(tmp_path / "eng")
    common = dict(OUT=out, KIND=kind, MB_DP_SPARSE_EMB=sparse, MB_DP_CHUNK=chunk)
    _run_engine_workers(tmp_path, 1, common)
    _run_engine_workers(tmp_path, 2, common)
    ref = torch.load(out + ".1.0.1")
    a, b = torch.load(out + ".2.0.1"), torch.load(out + ".2.1.1")
    assert a["fused"] and b["fused"] and not ref["fused"]
    assert a["sparse"] == (sparse == "1")
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["m"], b["m"])
    err = float((a["m"] - ref["m"]).abs().max()) / float(ref["m"].abs().max())
    print("%s single-call DP(2 x 4) vs single(8): first-moment max |d| / max = %.3e; %d collectives, %.1f MB" %
          (kind, err, a["stats"][0], a["stats"][1] * 1e-6))
    assert err <= 5e-6
    d = (a["p"] - ref["p"]).abs()
    assert float(d.max()) <= 2 * 1e-3 * 1.1


@pytest.mark.parametrize("kind,event_mode,control", [("bert", "2", "0"), ("bert", "3", "0"), ("bert", "0", "0"), ("xlnet", "2", "0"),
                                                     ("bert", "2", "1")])
def test_single_call_dp_step_replayed_steps_with_a_delayed_backward(tmp_path, kind, event_mode, control):
    """The compute -> comm hand-off of REPLAYED steps: lr = 0 so that the parameters stay put and Adam's first
    moment after step s is a fixed function of the all-reduced gradients of steps 1..s (a different batch every step); every backward
    segment starts with a 3 ms spin kernel (MB_DP_TEST_DELAY_US), so the host has issued the segment's collectives long before its
    gradients exist.  Two ranks (4 + 4 samples) must match one process (8 samples) <= 5e-6 after EVERY one of 4 steps (steps 2-4
    replay the captured graphs), for the event-record node (mode 2), the wait nodes (mode 3) and host-side records (mode 0).
    control = 1 is the negative control: MB_DP_DEBUG=4 drops the comm stream's wait -- the same comparison must then FAIL, i.e.
    this test does detect an exchange that runs ahead of the backward (round 4's plain hipEventRecord inside the capture did)."""
    import torch
    out = str(tmp_path / "dl")
    common = dict(OUT=out, KIND=kind, MB_DP_SPARSE_EMB="1", MB_DP_CHUNK="1", LR="0", STEPS="4", EVERY_M="1")
    _run_engine_workers(tmp_path, 1, common)
    extra = dict(MB_DP_EVENT_MODE=event_mode, MB_DP_TEST_DELAY_US="3000")
    if control == "1":
        extra["MB_DP_DEBUG"] = "4"
    _run_engine_workers(tmp_path, 2, dict(common, **extra))
    ref = torch.load(out + ".1.0.1")
    a, b = torch.load(out + ".2.0.1"), torch.load(out + ".2.1.1")
    assert a["fused"] and b["fused"] and not ref["fused"] and len(a["every_m"]) == 4
    errs = []
    for s in range(4):
        assert torch.equal(a["every_m"][s], b["every_m"][s]) or control == "1"
        errs.append(float((a["every_m"][s] - ref["every_m"][s]).abs().max()) / float(ref["every_m"][s].abs().max()))
    print("%s event mode %s%s: first-moment max |d| / max after steps 1..4 = %s" %
          (kind, event_mode, " NEGATIVE CONTROL (no wait)" if control == "1" else "", ", ".join("%.2e" % e for e in errs)))
    if control == "1":
        assert max(errs) > 1e-3, "the negative control passed: this test cannot see a missing dependency"
    else:
        assert max(errs) <= 5e-6


@pytest.mark.parametrize("kind", ["bert", "xlnet"])
def test_gradient_accumulation_in_the_single_call_lane(tmp_path, kind):
    """multimodal_driver.py:375-376, 383-386 under data parallel, fast lane: two ranks x two micro-steps of 2 samples -- the micro-steps
    are plain mb_*_train_step calls without the optimizer (nothing exchanged), the step that ends the window is mb_*_train_step_dp
    with the word-embedding table moved DENSELY (its rows are the union over the micro-steps) -- against one process with the
    whole batch of 8: Adam's first moment after step 1 <= 5e-6, replicas bit-identical, after 2 steps."""
    import torch
    out = str(tmp_path / "acc")
    common = dict(OUT=out, KIND=kind, MB_DP_SPARSE_EMB="1", MB_DP_CHUNK="1")
    _run_engine_workers(tmp_path, 1, common)
    _run_engine_workers(tmp_path, 2, dict(common, ACCUM="2"))
    ref = torch.load(out + ".1.0.1")
    a, b = torch.load(out + ".2.0.1"), torch.load(out + ".2.1.1")
    assert a["fused"] and b["fused"] and a["micro_fused"] == [True, True], a["micro_fused"]
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["m"], b["m"])
    err = float((a["m"] - ref["m"]).abs().max()) / float(ref["m"].abs().max())
    print("%s accumulation 2 ranks x 2 micro-steps x 2 samples vs single(8): first-moment max |d| / max = %.3e; %d collectives, %.1f MB" %
          (kind, err, a["stats"][0], a["stats"][1] * 1e-6))
    assert err <= 5e-6
    assert float((a["p"] - ref["p"]).abs().max()) <= 2 * 1e-3 * 1.1


@pytest.mark.parametrize("kind,cdt", [("bert", "fp32"), ("bert", "bf16"), ("xlnet", "fp32")])
def test_sharded_update_inside_the_engine_call_equals_replicated(tmp_path, kind, cdt):
    """MB_DP_SHARD_OPT=1 in the single-call lane (csrc/comm.hip: every layer piece reduce-scattered, AdamW over this rank's slices,
    in-place all-gathers of the next forward's operands on the comm stream) against the replicated single-call step, two ranks over
    the callback backend, three steps, deterministic mode.  fp32: parameters on every rank bit-identical to the replicated run
    after every gather (the forward reads the gathered fp32 parameters).  bf16: each rank's own slices of the masters bit-identical,
    the foreign ones stale by design, the bf16 shadow everybody's; after refresh_sharded_state() masters and Adam moments match
    the replicated run everywhere."""
    import torch
    out_a, out_b = str(tmp_path / "rep"), str(tmp_path / "shd")
    common = dict(KIND=kind, STEPS="3", CDT=cdt, MB_DETERMINISTIC="1", MB_DP_CHUNK="1")
    _run_engine_workers(tmp_path, 2, dict(common, OUT=out_a, MB_DP_SHARD_OPT="0"), tag="rep")
    _run_engine_workers(tmp_path, 2, dict(common, OUT=out_b, MB_DP_SHARD_OPT="1", GATHER_MASTERS="1"), tag="shd")
    rep = [torch.load(out_a + ".2.%d.1" % r) for r in range(2)]
    shd = [torch.load(out_b + ".2.%d.1" % r) for r in range(2)]
    assert rep[0]["fused"] and shd[0]["fused"] and shd[0]["slices"] and not rep[0]["slices"]
    assert torch.equal(rep[0]["p"], rep[1]["p"])
    n_sharded = sum(e - b for b, e in shd[0]["slices"])
    for r in range(2):
        assert torch.equal(shd[r]["p"], rep[0]["p"]), "masters after the gather"
        assert torch.equal(shd[r]["adam_m"], rep[0]["adam_m"]), "Adam moments after the gather"
        mine = torch.zeros(rep[0]["p"].numel(), dtype=torch.bool)
        for b, e in shd[r]["slices"]:
            mine[b:e] = True
            assert torch.equal(shd[r]["p_before_gather"][b:e], rep[0]["p"][b:e])
        other = torch.zeros_like(mine)
        for b, e in shd[1 - r]["slices"]:
            other[b:e] = True
        assert not bool((mine & other).any())
        assert torch.equal(shd[r]["p_before_gather"][~other], rep[0]["p"][~other])          # everything this rank updates itself
        assert not torch.equal(shd[r]["adam_m_before_gather"][other], rep[0]["adam_m"][other])     # foreign moments: never touched here
        if cdt == "fp32":
            assert torch.equal(shd[r]["p_before_gather"], rep[0]["p"])                       # the gathers deliver the fp32 parameters
        else:
            assert not torch.equal(shd[r]["p_before_gather"][other], rep[0]["p"][other])    # stale masters by design
            assert torch.equal(shd[r]["shadow"], rep[0]["shadow"])                           # ... while the bf16 operands are everybody's
    print("%s sharded update in the engine call (%s): 2 ranks == replicated bit for bit; %d of %d parameters sharded, %d slices per rank; "
          "%d collectives, %.1f MB reduced" % (kind, cdt, 2 * n_sharded, rep[0]["p"].numel(), len(shd[0]["slices"]), shd[0]["stats"][0], shd[0]["stats"][1] * 1e-6))


@pytest.mark.parametrize("cdt", ["fp32", "bf16"])
def test_sharded_step_with_a_cut_forward_over_rccl_one_rank(tmp_path, cdt):
    """The sharded update with FOUR pieces (4 layers, one per piece): the forward of the step is cut into three forward-only graphs +
    the one in front of the first backward segment, each waiting for the all-gather of its own piece; the lowest piece stays
    replicated (csrc/comm.h: cut mode).  Over RCCL itself with a one-rank communicator (MB_DP_SHARD_FORCE=1: reduce-scatter and
    all-gather are identities) in deterministic mode: parameters after three steps BIT-IDENTICAL to the plain single-call step."""
    import torch
    out = str(tmp_path / "cut")
    common = dict(OUT=out, KIND="bert", MB_DP_FORCE="1", BACKEND="nccl", STEPS="3", CDT=cdt, MB_DETERMINISTIC="1", LAYERS="4", MB_DP_CHUNK="1",
                  MB_DP_GRAD_DTYPE="fp32")
    _run_engine_workers(tmp_path, 1, dict(common, USE_DP="0"))
    _run_engine_workers(tmp_path, 1, dict(common, USE_DP="1", MB_DP_SHARD_OPT="1", MB_DP_SHARD_FORCE="1"))
    a, b = torch.load(out + ".1.0.0"), torch.load(out + ".1.0.1")
    assert b["fused"] and not a["fused"] and b["slices"] and len(b["slices"]) == 3          # three sharded pieces, the fourth replicated
    print("cut forward, one-rank RCCL (%s): %d collectives, %.1f MB; max |dparam| %.3e" % (cdt, b["stats"][0], b["stats"][1] * 1e-6,
                                                                                          float((a["p"] - b["p"]).abs().max())))
    assert torch.equal(a["p"], b["p"])


@pytest.mark.parametrize("cdt,graph", [("fp32", "1"), ("fp32", "0"), ("bf16", "1")])
def test_single_call_dp_step_over_rccl_one_rank(tmp_path, cdt, graph):
    """the same call over RCCL itself (dlopen'ed and driven from C: ncclCommInitRank from an id made by mb_comm_unique_id,
    ncclAllReduce / ncclAllGather on the comm stream) with a one-rank communicator: every collective is an identity, so in
    deterministic mode the parameters after three steps are BIT-IDENTICAL to the plain single-call step -- graph chain, events,
    row-wise exchange (pack -> gather -> combine) and the split optimizer included.  bf16: the wire format rounds the gradients."""
    import torch
    out = str(tmp_path / "rc")
    common = dict(OUT=out, KIND="bert", MB_DP_FORCE="1", BACKEND="nccl", STEPS="3", CDT=cdt, GRAPH=graph, MB_DETERMINISTIC="1",
                  MB_DP_GRAD_DTYPE=cdt)        # (bf16 wire: the automatic choice of two-GPU groups only, asked for here)
    _run_engine_workers(tmp_path, 1, dict(common, USE_DP="0"))
    _run_engine_workers(tmp_path, 1, dict(common, USE_DP="1"))
    a, b = torch.load(out + ".1.0.0"), torch.load(out + ".1.0.1")
    assert b["fused"] and not a["fused"]
    d = (a["p"] - b["p"]).abs()
    print("RCCL 1-rank single-call DP vs plain (%s): max |dparam| %.3e; %d collectives, %.1f MB" % (cdt, float(d.max()), b["stats"][0], b["stats"][1] * 1e-6))
    if cdt == "bf16":          # bf16 wire: the gradients are rounded once per rank
        assert float(d.max()) <= 3 * 2 * 1e-3 * 1.1 and float(d.mean()) <= 2e-5
    else:
        assert torch.equal(a["p"], b["p"])


def test_row_exchange_kernels_with_scripted_peers():
    """csrc/comm.hip's row-wise exchange of the word-embedding gradient (mark -> pack -> all-gather -> index -> combine) for a
    THREE-rank group played by one process: the all-gather callback of each rank publishes its piece and fills in the pieces the
    other ranks published (first pass: publish; second pass, on restored tables: the real exchange).  Every rank must end with the
    dense sum of the three tables -- repeated ids inside a rank, ids shared between ranks, ids only one rank touched, padding
    positions beyond T -- and all three bit-identical."""
    import ctypes as C
    import torch
    from bert_multimodal_transformer_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    world, vocab, H, cap = 3, 997, 768, 96
    g = torch.Generator().manual_seed(5)
    T = [96, 57, 80]
    ids = [torch.randint(0, 120 if r < 2 else vocab, (T[r],), generator=g) for r in range(world)]      # ranks 0 / 1 collide a lot
    ids[2][:10] = ids[0][:10]
    tables = []
    for r in range(world):
        t = torch.zeros(vocab, H)
        u = ids[r].unique()
        t[u] = torch.randn(len(u), H, generator=g)
        tables.append(t)
    dense = tables[0] + tables[1] + tables[2]
    published = {}
    state = {}

    def make_cb(r):
        def ag(ctx, buf, bpr, stream):
            torch.cuda.synchronize()
            sc = state[r]["scratch"]
            off = buf - sc.data_ptr()
            v = sc[off: off + bpr * world]
            published[(r, bpr)] = v[r * bpr:(r + 1) * bpr].clone()
            for q in range(world):
                if q != r and (q, bpr) in published:
                    v[q * bpr:(q + 1) * bpr].copy_(published[(q, bpr)])
                elif q != r:
                    v[q * bpr:(q + 1) * bpr].view(torch.int32).fill_(-1)      # nothing published yet: "no rows" (ids) / ignored (rows)
            torch.cuda.synchronize()
            return 0

        def ar(ctx, buf, count, dtype, stream):
            return 1004
        return _lib.ALL_REDUCE_CB(ar), _lib.ALL_GATHER_CB(ag)

    for r in range(world):
        cbs = make_cb(r)
        h = C.c_void_p()
        _lib.check(L.mb_comm_create_callbacks(r, world, C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), None, C.byref(h)))
        nb = L.mb_comm_scratch_bytes(world, _lib.DT_F32, 1024, vocab, H, cap)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(L.mb_comm_bind_scratch(h, _lib.ptr(sc), nb, _lib.DT_F32, 1024, vocab, H, cap))
        state[r] = dict(h=h, cbs=cbs, scratch=sc, ids=ids[r].to(dev), stream=L.mb_comm_stream(h))
    torch.cuda.synchronize()
    results = None
    for rounds in range(2):
        results = []
        for r in range(world):
            tab = tables[r].to(dev).contiguous()
            _lib.check(L.mb_comm_exchange_rows(state[r]["h"], _lib.ptr(tab), _lib.ptr(state[r]["ids"]), T[r], state[r]["stream"]))
            torch.cuda.synchronize()
            results.append(tab.cpu())
    for r in range(world):
        err = float((results[r] - dense).abs().max())
        assert err <= 1e-5, (r, err)
        assert torch.equal(results[r], results[0])
    untouched = torch.ones(vocab, dtype=torch.bool)
    untouched[torch.cat(ids).unique()] = False
    assert float(results[0][untouched].abs().max()) == 0.0
    for r in range(world):
        L.mb_comm_destroy(state[r]["h"])


@pytest.mark.parametrize("cdt", ["fp32", "bf16"])
def test_sharded_optimizer_update_equals_replicated(tmp_path, cdt):
    """MB_DP_SHARD_OPT=1 (distributed.OptimizerShards): the gradient pieces of the layers' GEMM weights are reduced to their owner
    rank only, every rank updates ONE shard of that range (plus the replicated rest), and the operands of the next forward come
    back by all-gather -- fp32 masters in parity mode, the bf16 shadow in perf mode.  Two ranks over gloo on one GPU, three steps:
    fp32 -> every parameter on every rank bit-identical to the replicated data-parallel path; bf16 -> each rank's OWN shard of
    masters bit-identical to the replicated run, the stale foreign shards refreshed by gather_masters()."""
    import torch
    out_a, out_b = str(tmp_path / "rep"), str(tmp_path / "shd")
    common = dict(KIND="bert", MB_DP_ENGINE="0", STEPS="3", CDT=cdt, MB_DETERMINISTIC="1")
    _run_engine_workers(tmp_path, 2, dict(common, OUT=out_a, MB_DP_SHARD_OPT="0"), tag="rep")
    _run_engine_workers(tmp_path, 2, dict(common, OUT=out_b, MB_DP_SHARD_OPT="1", GATHER_MASTERS="1"), tag="shd")
    rep = [torch.load(out_a + ".2.%d.1" % r) for r in range(2)]
    shd = [torch.load(out_b + ".2.%d.1" % r) for r in range(2)]
    assert torch.equal(rep[0]["p"], rep[1]["p"])
    assert not shd[0]["fused"] and shd[0]["shards"] is not None
    for r in range(2):
        a, b = shd[r]["shards"][r]
        assert torch.equal(shd[r]["p_before_gather"][a:b], rep[0]["p"][a:b])          # the owner's shard: the replicated update, bit for bit
        lo, hi = shd[r]["shards"][0][0], shd[r]["shards"][-1][1]
        assert torch.equal(shd[r]["p_before_gather"][:lo], rep[0]["p"][:lo]) and torch.equal(shd[r]["p_before_gather"][hi:], rep[0]["p"][hi:])
        assert torch.equal(shd[r]["p"], rep[0]["p"])                                   # after gather_masters(): everything
        if cdt == "fp32":
            assert torch.equal(shd[r]["p_before_gather"], rep[0]["p"])                 # parity mode gathers the masters every step
        else:
            other = shd[r]["shards"][1 - r]
            assert not torch.equal(shd[r]["p_before_gather"][other[0]:other[1]], rep[0]["p"][other[0]:other[1]])      # stale by design
            assert torch.equal(shd[r]["shadow"], rep[0]["shadow"])                     # ... while the bf16 operands are everybody's
    print("sharded optimizer (%s): 2 ranks == replicated path bit for bit; shards %s" % (cdt, shd[0]["shards"]))


def test_bench_py_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it: N ranks started by bench.py itself, rank 0 prints the ONE
    JSON line with n_gpus = N, rccl_ranks = N (an all-reduce of ones through mb_comm_all_reduce), the data-parallel step call and the
    comm statistics.  Two ranks share this GPU over the callback backend (MB_DIST_BACKEND=gloo); asked for two ranks over RCCL with one
    device visible it exits non-zero with a message -- never a silent 1-GPU number."""
    import json
    env = dict(os.environ, MB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--cpu-baseline", "0",
                        "--roofline", "0", "--secondary", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 96 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["step_call"].startswith("mb_bert_train_step_dp") and d["launched_by"].startswith("bench.py itself")
    assert d["comm_collectives_per_step"] >= 5 and d["comm_mbytes_per_step"] > 300 and "comm_exposed_ms" in d
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("MB_DIST_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-baseline", "0", "--roofline", "0"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "device(s) visible" in (r.stdout + r.stderr)
        assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
