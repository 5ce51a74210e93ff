#!/usr/bin/env python
"""bench.py -- train samples/sec of MAG-BERT fine-tuning (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full iteration of train_epoch (/root/reference/multimodal_driver.py:354-388) on one minibatch per GPU: the
batch comes from HOST memory every step (`batch = tuple(t.to(DEVICE) ...)`, :359 -- here one pinned block the step's first
launch gathers across PCIe), forward (embeddings, MAG, 12 encoder layers, pooler, classifier) -> MSE -> backward -> (N>1: RCCL
all-reduce of the flat gradients, overlapped with the backward) -> HF-AdamW -> linear-warmup schedule -> zero_grad.  `value`
is measured with the batch tensors already RESIDENT in HBM when the timed region starts (the bench contract); the same loop fed
from pinned host memory every step -- what rounds 1-5 reported as `value` -- is `value_with_per_step_h2d` (a ~35 us PCIe gather
per step).  Workload = BASELINE.json configs[1]: bert-base-uncased
MAG-BERT, MOSI dims (V=47, A=74), B=48/GPU, L=50, bf16 MFMA with fp32 master weights, dropout ON (0.1/0.1/MAG 0.5), synthetic
batches in prepare_bert_input's layout, random-init weights (no network).  Weak scaling: per-GPU batch fixed, global batch = 48*N.
At N=1 the whole iteration is ONE engine call (mb_bert_train_step: step prologue + one replayed hipGraph).

Prints ONE JSON line (rank 0) with the contract's keys plus
  step_ms_median / p10 / p90 : per-step GPU time from HIP events recorded after every step
  roofline       : the kernel with the largest share of the step -- the per-layer grouped weight-gradient GEMM (MFMA-bound) or the
                   optimizer (HBM-bound) -- timed INSIDE the step with HIP events on the stream it runs on; both are always printed
                   (roofline_mfma, roofline_adamw); HBM-side traffic is REPLAYED from the committed PMC pass (profiles/pmc_traffic.json)
  roofline_gemms : the nine GEMM launches of a layer, back-to-back (warm caches: an upper bound)
  roofline_hbm   : achieved HBM TB/s of the LayerNorm / AdamW row kernels (north_star), HIP events, rotating operands
  instep_kernels : per-kernel in-step table REPLAYED from the committed rocprofv3 kernel trace of this step (profiles/)
  cpu_baseline   : the CPU oracle (oracle/mag_bert_ref.py, kind "port") timed on this box's host cores, same step
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md); fp32 MFMA 157.3
PEAK_F32_TFLOPS = 157.3
# algorithmic FLOPs: forward 8.7234 GFLOP/sample at L=50, V=47; training = 3x
TRAIN_GFLOP_PER_SAMPLE_L50 = 26.170
TRAIN_GFLOP_PER_SAMPLE_C5 = 68.080      # MOSEI V=35, L=128


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--batch", type=int, default=48)
    p.add_argument("--seq", type=int, default=50)
    p.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    p.add_argument("--dataset", choices=["mosi", "mosei"], default="mosi")
    p.add_argument("--model", choices=["bert", "xlnet"], default="bert",
                   help="bert = the headline workload (BASELINE.json configs[1]); xlnet = configs[3] (MAG-XLNet), informational")
    p.add_argument("--cpu-baseline", type=int, default=1)
    p.add_argument("--cpu-steps", type=int, default=5)
    p.add_argument("--roofline", type=int, default=1)
    p.add_argument("--graph", type=int, default=1,
                   help="1 (default): each optimizer step = step prologue + ONE replayed hipGraph (mb_bert_train_step mode 1; MAG-BERT, "
                        "single process); 0: the same single engine call launching the kernels on the stream one by one")
    p.add_argument("--roofline-only", type=int, default=0, help="skip the training loop, print the GEMM table only")
    p.add_argument("--epoch", type=int, default=1,
                   help="1 (default, N = 1, headline workload only): also time one MOSI-sized epoch exactly as the reference's train() runs it "
                        "(27 train steps incl. the ragged 33-sample tail, eval_epoch and test_epoch at B = 128) -> epoch_ms / epoch_train_ms / "
                        "epoch_eval_test_ms; 0: skip")
    p.add_argument("--secondary", type=int, default=1,
                   help="1 (default, N = 1, headline workload only): also measure BASELINE.json configs[3] (MAG-XLNet, B=48, L=50) and the per-GPU "
                        "shape of configs[4] (MAG-BERT MOSEI V=35, B=32, L=128) in the same process and append them as `secondary`")
    return p.parse_args()


def make_batches(n, B, L, V, A, seed, layout="bert"):
    from bert_multimodal_transformer_amd.multimodal_driver import synthetic_dataset
    ds = synthetic_dataset(n * B, L, V, A, seed_=seed, layout=layout)
    out = []
    for i in range(n):
        out.append(tuple(t[i * B:(i + 1) * B].contiguous().pin_memory() for t in ds.tensors))
    return out


def cpu_baseline(B, L, V, A, steps, kind="bert"):
    """The same optimizer step on the host cores with the CPU oracle (pure torch restatement of the reference)."""
    from oracle import mag_bert_ref as R, optim_ref as O
    from oracle import mag_xlnet_ref as X
    from bert_multimodal_transformer_amd.multimodal_driver import synthetic_dataset
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    # a container is entitled to its cgroup CPU quota, not to the machine: more runnable threads than that and the kernel
    # throttles the whole process (round 2: 6.4 s per step on "128 cores").  The baseline runs on what the box grants.
    quota = cpu_quota()
    threads = cores if quota is None else max(1, min(cores, int(quota)))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    if kind == "xlnet":
        model = X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(), X.MultimodalConfig(1.0, 0.5), V, A).train()
    else:
        model = R.MAG_BertForSequenceClassification(R.BertConfigLite(), R.MultimodalConfig(1.0, 0.5), V, A).train()
    opt = O.AdamW(O.grouped_parameters(model), lr=1e-5)
    sch = O.get_linear_schedule_with_warmup(opt, 0.1 * 1040, 1040)
    ds = synthetic_dataset(B * (steps + 1), L, V, A, seed_=7, layout=kind)
    times = []
    for s in range(steps + 1):
        ids, vis, aco, mask, seg, lab = (t[s * B:(s + 1) * B] for t in ds.tensors)
        t0 = time.perf_counter()
        logits = model(ids, vis, aco, mask, seg)[0]
        loss = torch.nn.functional.mse_loss(logits.view(-1), lab.view(-1))
        loss.backward()
        float(loss.item())
        opt.step(); sch.step(); opt.zero_grad()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:])) if steps > 0 else float("nan")
    gflop = {("bert", 50): TRAIN_GFLOP_PER_SAMPLE_L50, ("bert", 128): TRAIN_GFLOP_PER_SAMPLE_C5}.get((kind, L))
    return {"value": B / t, "unit": "samples/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "cgroup_cpu_quota": quota, "physical_cores_of_the_host": cores,
            "achieved_gflops": round(B / t * gflop, 1) if gflop else None,
            "sample": "%d timed optimizer steps (1 warmup) of the same B=%d L=%d MAG-%s step, fp32, dropout on, "
                      "oracle/mag_%s_ref.py + HF-AdamW, median step %.2f s, %d torch threads (cgroup quota %s)"
                      % (steps, B, L, kind.upper(), kind, t, threads, "none" if quota is None else "%.1f CPUs" % quota)}


def cpu_quota():
    """CPUs this process may use according to its cgroup (v2 cpu.max, v1 cfs quota), or None when unlimited / unreadable"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def gemm_roofline(dtype_name, T, reps=30):
    """Time the encoder's GEMM launches through the C ABI with HIP events on the launch stream.  Returns the list of
    (name, shape, avg us, TFLOP/s) and the entry for the dominant kernel (largest share of a training step)."""
    import ctypes as C
    from bert_multimodal_transformer_amd import _lib
    L = _lib.lib()
    dt = _lib.DT_BF16 if dtype_name == "bf16" else _lib.DT_F32
    tdt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    dev = torch.device("cuda", torch.cuda.current_device())
    H, I = 768, 3072
    Tp = (T + 63) // 64 * 64       # the engine zero-pads the token dimension to the GEMM k-tile of the weight gradients:
    st = torch.cuda.current_stream()      # launches use what the engine uses (M = T, wgrad K = Tp), FLOPs count T (algorithmic)
    def g(rows, cols):
        t = (torch.randn(rows, cols, device=dev) * 0.05).to(tdt)
        if rows == Tp and Tp > T:
            t[T:].zero_()
        return t
    x, xi, w_qkv, w_o, w1, w2 = g(Tp, H), g(Tp, I), g(3 * H, H), g(H, H), g(I, H), g(H, I)
    dqkv = g(Tp, 3 * H)
    bias = torch.zeros(3 * I, device=dev)
    out = torch.empty(Tp, I, dtype=tdt, device=dev); out2 = torch.empty(Tp, I, dtype=tdt, device=dev)
    outf = torch.zeros(I, I, dtype=torch.float32, device=dev)
    key = _lib.make_dropkey(1, 1, 17, 0.1)
    NT, NN, TN = _lib.GEMM_NT, _lib.GEMM_NN, _lib.GEMM_TN
    # (name, count per layer per step, layout, epilogue, M, N, K, A, lda, B, ldb, R, splits)
    cases = [
        ("fwd qkv   [T,768]x[2304,768]^T +bias", 1, NT, _lib.EPI_BIAS, T, 3 * H, H, x, H, w_qkv, H, None, 1),
        ("fwd out   [T,768]x[768,768]^T +bias+drop+res", 1, NT, _lib.EPI_BIAS_DROP_RES, T, H, H, x, H, w_o, H, x, 1),
        ("fwd ffn1  [T,768]x[3072,768]^T +bias+gelu", 1, NT, _lib.EPI_BIAS_GELU, T, I, H, x, H, w1, H, None, 1),
        ("fwd ffn2  [T,3072]x[768,3072]^T +bias+drop+res", 1, NT, _lib.EPI_BIAS_DROP_RES, T, H, I, xi, I, w2, I, x, 1),
        ("dgrad ffn2 [T,768]x[768,3072] *gelu'", 1, NN, _lib.EPI_DGELU, T, I, H, x, H, w2, I, xi, 1),
        ("dgrad ffn1 [T,3072]x[3072,768] +res", 1, NN, _lib.EPI_ADD_RES, T, H, I, xi, I, w1, H, x, 1),
        ("dgrad out  [T,768]x[768,768]", 1, NN, _lib.EPI_ADD_RES, T, H, H, x, H, w_o, H, None, 1),
        ("dgrad qkv  [T,2304]x[2304,768] +res", 1, NN, _lib.EPI_ADD_RES, T, H, 3 * H, dqkv, 3 * H, w_qkv, H, x, 1),
        # the layer's four weight gradients are ONE grouped launch in the engine (csrc/gemm_pp.hip gemm_pp_grouped_tn_kernel: 256 x 128 ping-pong tiles)
        ("wgrad x4  grouped [768x3072|3072x768|768x768|2304x768] K=T", 1, "grouped", None, 0, 0, Tp, None, 0, None, 0, None, 1),
    ]
    gshape = [(H, I), (I, H), (H, H), (3 * H, H)]
    gY, gX = [x, xi, x, dqkv], [xi, x, x, x]
    gW = [torch.zeros(m, n, device=dev) for m, n in gshape]
    ia = lambda v: (C.c_int * 4)(*v)
    pa = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
    gM, gN, gpY, gpX, gpW = ia([m for m, n in gshape]), ia([n for m, n in gshape]), pa(gY), pa(gX), pa(gW)
    gtile = int(os.environ.get("MB_GROUP_WGRAD", "256" if dtype_name == "bf16" else "128")) or 128       # the engine's default: csrc/engine.hip group_wgrad
    res = []
    for name, cnt, layout, epi, M, N, K, A, lda, Bm, ldb, R, splits in cases:
        def launch():
            if layout == "grouped":
                _lib.check(L.mb_gemm_grouped_wgrad(dt, 4, gM, gN, K, gpY, gM, gpX, gN, gpW, gN, gtile, st.cuda_stream))
                return
            _lib.check(L.mb_gemm(dt, layout, epi, M, N, K, _lib.ptr(A), lda, _lib.ptr(Bm), ldb, _lib.ptr(out), N,
                                 _lib.ptr(out2), _lib.ptr(outf), _lib.ptr(bias), _lib.ptr(R), N, 1.0, C.byref(key), splits, 0,
                                 st.cuda_stream))
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            launch()
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 2.0 * M * N * K if layout != "grouped" else sum(2.0 * m * n * T for m, n in gshape)     # algorithmic: T tokens, not the padded Tp
        res.append({"kernel": name, "M": M, "N": N, "K": K, "avg_us": round(us, 2), "tflops": round(fl / us * 1e-6, 1),
                    "flop": fl})
    return res


def hbm_roofline(dtype_name, B, L, V, A, reps=20):
    """Achieved HBM GB/s of the row kernels (north_star: "rocprof reports achieved HBM GB/s on the MAG/LayerNorm kernels"):
    each kernel is launched through the C ABI on synthetic operands and timed with HIP events on the launch stream,
    back-to-back with rotating buffers larger than the L2s.  Bytes = algorithmic bytes."""
    import ctypes as C
    from bert_multimodal_transformer_amd import _lib
    Lb = _lib.lib()
    dt = _lib.DT_BF16 if dtype_name == "bf16" else _lib.DT_F32
    tdt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    es = 2 if dtype_name == "bf16" else 4
    dev = torch.device("cuda", torch.cuda.current_device())
    H, T = 768, B * L
    st = torch.cuda.current_stream()
    NB = 8                                   # rotate over 8 operand sets so the inputs are not L2-resident
    xs = [(torch.randn(T, H, device=dev) * 0.5).to(tdt) for _ in range(NB)]
    ys = [torch.empty(T, H, dtype=tdt, device=dev) for _ in range(NB)]
    zs = [torch.empty(T, H, dtype=tdt, device=dev) for _ in range(NB)]
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    mean, rstd = torch.zeros(T, device=dev), torch.ones(T, device=dev)
    dg, db, dbias = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    key = _lib.make_dropkey(1, 1, 17, 0.1)
    nokey = _lib.no_drop()
    out = []

    NPAIR, PER = 9, max(4, reps // 2)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(NPAIR + 1)]       # created once, re-recorded (a host stall between two
    for ev in evs:                                                               # enqueues lands in ONE pair: the median ignores it)
        ev.record(st)
    torch.cuda.synchronize()

    def timed(name, nbytes, fn):
        for i in range(3):
            fn(i)
        k = 0
        evs[0].record(st)
        for pair in range(NPAIR):
            for _ in range(PER):
                fn(k); k += 1
            evs[pair + 1].record(st)
        evs[NPAIR].synchronize()
        spans = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 / PER for i in range(NPAIR))
        us = spans[NPAIR // 2]
        out.append({"kernel": name, "bytes": int(nbytes), "avg_us": round(us, 2), "min_us": round(spans[0], 2), "max_us": round(spans[-1], 2),
                    "tb_per_s": round(nbytes / us * 1e-6, 3), "frac_of_8tbs": round(nbytes / us * 1e-6 / 8.0, 4),
                    "timing": "median of %d HIP-event spans of %d launches" % (NPAIR, PER)})

    timed("ln_fwd (LayerNorm, BertSelfOutput/BertOutput)", 2 * T * H * es, lambda i: _lib.check(Lb.mb_layernorm_forward(
        dt, _lib.ptr(xs[i % NB]), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(ys[i % NB]), _lib.ptr(mean), _lib.ptr(rstd),
        T, H, C.byref(nokey), st.cuda_stream)))
    # (the operator-level launch: column sums by atomics.  Inside the step the partial-sum variant runs -- instep_kernels: ~8 us)
    timed("ln_bwd operator (+dropout backward, dgamma/dbeta/dbias by atomics)", 4 * T * H * es, lambda i: _lib.check(Lb.mb_layernorm_backward(
        dt, _lib.ptr(xs[i % NB]), _lib.ptr(ys[i % NB]), _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(zs[i % NB]),
        _lib.ptr(ys[(i + 1) % NB]), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), T, H, C.byref(nokey), C.byref(key), st.cuda_stream)))
    # MAG forward, the whole operator as the engine runs it (weight pack + modality pack + 3 MFMA GEMMs + the gate / norm-ratio /
    # LayerNorm / dropout row kernel): algorithmic bytes = read e, v, a + write out per token
    mag_ws = torch.empty(Lb.mb_mag_workspace_bytes(dt, T, H, V, A), dtype=torch.uint8, device=dev)
    vis = [torch.randn(T, V, device=dev) for _ in range(NB)]
    aco = [torch.randn(T, A, device=dev) for _ in range(NB)]
    mp = [torch.randn(H, V + H, device=dev) * 0.02, torch.zeros(H, device=dev), torch.randn(H, A + H, device=dev) * 0.02,
          torch.zeros(H, device=dev), torch.randn(H, V, device=dev) * 0.02, torch.zeros(H, device=dev),
          torch.randn(H, A, device=dev) * 0.02, torch.zeros(H, device=dev), gamma, beta]
    mkey = _lib.make_dropkey(1, 1, 1, 0.5)
    timed("mag_forward (modeling.py:25-51: pack + 3 GEMMs + gate/LN/dropout kernel)", T * (2 * H * es + 4 * V + 4 * A),
          lambda i: _lib.check(Lb.mb_mag_forward(dt, _lib.ptr(xs[i % NB]), _lib.ptr(vis[i % NB]), _lib.ptr(aco[i % NB]),
                                                 *[_lib.ptr(t) for t in mp], 1.0, C.byref(mkey), _lib.ptr(ys[i % NB]), _lib.ptr(mag_ws),
                                                 T, H, V, A, st.cuda_stream)))
    n = 110_853_184
    p_, g_, m_, v_ = (torch.zeros(n, device=dev) for _ in range(4))
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    timed("adamw (p,g,m,v read; p,m,v,g=0 + bf16 shadow written)", n * 34, lambda i: _lib.check(Lb.mb_adamw_step(
        p_.data_ptr(), g_.data_ptr(), m_.data_ptr(), v_.data_ptr(), sh.data_ptr(), n, n, 0, n, 1e-5, 0.9, 0.999, 1e-6, 0.01, i + 1, 1,
        1.0, 1, st.cuda_stream)))
    del p_, g_, m_, v_, sh
    return out


def secondary_workload(kind, dataset, B, L, steps=20, warmup=5, dtype="bf16", cpu_steps=0):
    """One of the other single-GPU configurations of BASELINE.json, measured in this process the way the headline is: the same
    train_epoch step as ONE engine call / replayed hipGraph, `warmup` untimed + `steps` timed steps with the batch resident in HBM
    (`value`) and fed from pinned host memory (`value_with_per_step_h2d`); `roofline` from a rocprofv3 kernel trace this run takes
    itself over tools/bin/step_bench with the same shape (every symbol priced: price_trace / pick_roofline, as for the headline);
    cpu_steps > 0: the CPU oracle timed on the same step (`cpu_baseline`)."""
    from bert_multimodal_transformer_amd import (AdamW, BertConfig, MAG_BertForSequenceClassification, MultimodalConfig,
                                                 get_linear_schedule_with_warmup)
    from bert_multimodal_transformer_amd.global_configs import DATASET_DIMS
    from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
    from bert_multimodal_transformer_amd.prefetch import PinnedBatchRing
    V, A = DATASET_DIMS[dataset]["visual_dim"], DATASET_DIMS[dataset]["acoustic_dim"]
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    torch.manual_seed(4321)
    if kind == "xlnet":
        from bert_multimodal_transformer_amd import MAG_XLNetForSequenceClassification, XLNetConfig
        model = MAG_XLNetForSequenceClassification(XLNetConfig(num_labels=1), MultimodalConfig(1.0, 0.5), visual_dim=V, acoustic_dim=A,
                                                   compute_dtype=cdt)
    else:
        model = MAG_BertForSequenceClassification(BertConfig(num_labels=1), MultimodalConfig(1.0, 0.5), visual_dim=V, acoustic_dim=A,
                                                  compute_dtype=cdt)
    opt = AdamW(optimizer_grouped_parameters(model), lr=1e-5)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=0.1 * 1040, num_training_steps=1040)
    model.train()
    nb = 4
    batches = make_batches(nb, B, L, V, A, seed=99, layout=kind)
    dev = torch.device("cuda", torch.cuda.current_device())
    ring = PinnedBatchRing(None, dev)
    resident = [tuple(t.to(dev) for t in b) for b in batches]

    def run(n, start):
        ring.loader = (batches[(start + i) % nb] for i in range(n))
        for ids, vis, aco, mask, seg, lab in ring:
            model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=True)
            sch.step()

    def run_resident(n, start):
        for i in range(n):
            ids, vis, aco, mask, seg, lab = resident[(start + i) % nb]
            model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=True)
            sch.step()

    with model.stream_scope():
        run(3, 0)
        run_resident(warmup, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_resident(steps, warmup)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n2 = max(4, steps // 2)
        t1 = time.perf_counter()
        run(n2, 0)
        torch.cuda.synchronize()
        dt_h2d = (time.perf_counter() - t1) / n2
    n_update = int(model._core.n_update_end)
    mname = "MAG-BERT" if kind == "bert" else "MAG-XLNet"
    out = {"metric": "train samples/sec %s %s seq_len=%d" % (mname, dataset.upper(), L), "value": round(B * steps / dt, 2), "unit": "samples/s",
           "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": dtype,
           "value_with_per_step_h2d": round(B / dt_h2d, 2),
           "config": {"workload": "%s, %s dims (V=%d, A=%d), batch %d, seq_len %d, full optimizer step, inputs resident in HBM, dropout on, "
                                  "synthetic batches, random-init weights" % (mname, dataset.upper(), V, A, B, L),
                      "step_call": "mb_%s_train_step, hipGraph replay" % kind}}
    gflop = TRAIN_GFLOP_PER_SAMPLE_C5 if (kind == "bert" and L == 128 and V == 35) else (TRAIN_GFLOP_PER_SAMPLE_L50 if (kind == "bert" and L == 50 and V == 47) else None)
    if gflop:
        out["step_tflops_algorithmic"] = round(out["value"] * gflop * 1e-3, 1)
    del model, opt, sch, resident, ring
    torch.cuda.empty_cache()
    # the dominant kernel: from this run's own kernel trace of the same step (torch-free driver, ~2 s), every symbol priced
    if os.environ.get("MB_BENCH_TRACE", "1") != "0":
        try:
            doc, roof = instep_trace(B, L, V, dtype, n_update, steps=8, warmup=3, model=kind)
        except Exception as ex:          # noqa: BLE001
            doc, roof = None, "trace failed: %r" % (ex,)
        if doc is not None and roof:
            out["roofline"] = pick_roofline(doc, roof)
            out["roofline_trace"] = roof[:5]
            if "gemm_aggregate" in doc:
                out["gemm_aggregate"] = doc["gemm_aggregate"]
            out["instep_busy_ms_per_step"] = doc["busy_ms_per_step"]
        else:
            out["roofline"] = None
            out["roofline_note"] = str(roof)
    if cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(B, L, V, A, cpu_steps, kind)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


def epoch_mode(dtype):
    """One epoch of the loop the drop-in claim is about -- /root/reference/multimodal_driver.py:483-523 as
    bert_multimodal_transformer_amd.multimodal_driver runs it: train_epoch over MOSI's 1,281 training samples (26 steps of 48 + a ragged
    33-sample step, every batch from host memory), then eval_epoch (229 samples) and test_epoch + metrics (685 samples) at B = 128, on
    synthetic data of those sizes.  Epoch 0 captures the graphs (three train shapes would be a lie: two -- 48 and 33 -- plus the
    evaluation forwards); epochs 1 and 2 are timed, the faster is reported."""
    from torch.utils.data import DataLoader
    from bert_multimodal_transformer_amd import multimodal_driver as D
    D.args = D.parse_args(["--dataset", "mosi", "--compute_dtype", dtype, "--seed", "1234", "--n_epochs", "3"])
    D.set_random_seed(1234)
    V, A = D._dims()
    L = D.args.max_seq_length
    sizes = (1281, 229, 685)                      # MOSI train / dev / test (CMU-MOSI split of the reference's mosi.pkl)
    train = DataLoader(D.synthetic_dataset(sizes[0], L, V, A, 1234), batch_size=D.args.train_batch_size, shuffle=True)
    dev = DataLoader(D.synthetic_dataset(sizes[1], L, V, A, 1235), batch_size=D.args.dev_batch_size, shuffle=True)
    test = DataLoader(D.synthetic_dataset(sizes[2], L, V, A, 1236), batch_size=D.args.test_batch_size, shuffle=True)
    steps = len(train)
    model, opt, sch = D.prep_for_training(steps * 3)
    best = None
    for ep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tl = D.train_epoch(model, train, opt, sch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vl = D.eval_epoch(model, dev, opt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        acc, mae, corr, f1 = D.test_score_model(model, test)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if ep > 0 and (best is None or t3 - t0 < best[0]):
            best = (t3 - t0, t1 - t0, t2 - t1, t3 - t2, tl, vl)
    caps = model._core.graph_stats()
    del model, opt, sch
    torch.cuda.empty_cache()
    return {"epoch_ms": round(best[0] * 1e3, 2), "epoch_train_ms": round(best[1] * 1e3, 2), "epoch_eval_ms": round(best[2] * 1e3, 2),
            "epoch_test_ms": round(best[3] * 1e3, 2), "epoch_eval_test_ms": round((best[2] + best[3]) * 1e3, 2),
            "epoch_train_steps": steps, "epoch_samples": {"train": sizes[0], "dev": sizes[1], "test": sizes[2]},
            "epoch_train_samples_per_s": round(sizes[0] / best[1], 1), "epoch_train_loss": round(float(best[4]), 4), "epoch_valid_loss": round(float(best[5]), 4),
            "epoch_graph_captures_replays": list(caps),
            "epoch_note": "multimodal_driver.train()'s body for one epoch on synthetic MOSI-sized splits: train_epoch (every batch from pinned host "
                          "memory, the last one ragged: 33 samples, its own captured graph) + eval_epoch + test_epoch with metrics (B = 128, "
                          "T = 6,400 tokens per forward); the faster of two timed epochs after one that captures the graphs"}


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start N ranks of this script, one per visible device,
    on a free rendezvous port, and wait for them.  Rank 0 prints the JSON line (stdout is inherited).  Fewer devices than N is an
    error, never a silent 1-GPU number -- except under MB_DIST_BACKEND=gloo, the callback backend of the one-GPU tests, where ranks
    may share a device (the line then says so in `devices_visible`)."""
    import socket
    import subprocess
    n = a.gpus
    have = torch.cuda.device_count()
    backend = os.environ.get("MB_DIST_BACKEND", "nccl")
    if have < 1:
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the product path)")
    if have < n and backend == "nccl":
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible; RCCL needs one device per rank (no silent fallback to fewer GPUs)" % (n, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % have), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MB_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:            # a rank that dies leaves the others inside a collective: stop them (exact PIDs) instead of hanging
        time.sleep(0.2)
        for p_ in list(alive):
            code = p_.poll()
            if code is None:
                continue
            alive.remove(p_)
            if code != 0 and rc == 0:
                rc = code
                for q_ in alive:
                    q_.terminate()
    raise SystemExit(rc)


def kernel_base(name):
    """`mb::<function>` of a kernel symbol, mangled (_ZN2mb<len><function>...) or demangled (mb::<function>(...) / void mb::<function><...>(...))"""
    if name.startswith("_ZN2mb"):
        i = 6
        j = i
        while j < len(name) and name[j].isdigit():
            j += 1
        if j > i:
            n = int(name[i:j])
            return "mb::" + name[j:j + n]
    k = name.split("(")[0].split("<")[0].strip()
    return k[5:] if k.startswith("void ") else k


def kernel_literals(name):
    """`mb::<function><a, b, ...>` for a kernel whose template arguments are all bool / int literals -- from the mangled symbol the library
    logs (_ZN2mb14gemm_pn_kernelILb0ELb0ELi2EEEvNS_8GemmArgsE) or from what rocprofv3 prints for such a kernel (void mb::gemm_pn_kernel<false,
    false, 2>(mb::GemmArgs)): the key the two instantiations of one template are told apart by.  None for anything else."""
    if name.startswith("_ZN2mb"):
        j = 6
        while j < len(name) and name[j].isdigit():
            j += 1
        if j == 6:
            return None
        n = int(name[6:j])
        fn, rest = name[j:j + n], name[j + n:]
        if not rest.startswith("I"):
            return None
        args, i = [], 1
        while i < len(rest) and rest[i] == "L":
            e = rest.find("E", i)
            if e < 0 or rest[i + 1] not in "bijlmxy":
                return None
            v = rest[i + 2:e]
            v = ("-" + v[1:]) if v.startswith("n") else v
            if not v.lstrip("-").isdigit():
                return None
            args.append(("true" if v != "0" else "false") if rest[i + 1] == "b" else v)
            i = e + 1
        if i >= len(rest) or rest[i] != "E" or not args:
            return None
        return "mb::%s<%s>" % (fn, ", ".join(args))
    k = name.split("(")[0].strip()
    k = k[5:] if k.startswith("void ") else k
    if k.startswith("mb::") and k.endswith(">") and "<" in k:
        fn, a = k[:-1].split("<", 1)
        args = [x.strip() for x in a.split(",")]
        if all(x in ("true", "false") or x.lstrip("-").isdigit() for x in args):
            return "%s<%s>" % (fn, ", ".join(args))
    return None


def price_trace(rows, gemm_log, B, L, dtype, n_update, steps, model="bert"):
    """rows: (start ns, end ns, kernel name) of a kernel trace of consecutive training steps (a run of AdamW sweep launches ends a step);
    gemm_log: the library's MB_GEMM_LOG=1 lines of the same run.  -> (per-kernel table over the last `steps` steps, roofline rows: every
    GEMM symbol priced with the FLOPs it was launched with -- weight gradients count T, not the zero-padded Tp -- and the AdamW sweep
    with SURVEY 8(d)'s 28 B/parameter over the parameters ITS launches cover: with riders (csrc/kernels.h AdamRide) part of the update
    runs inside the grouped weight-gradient launches, `[magbert adamw] n=` / `[magbert ride] params=` lines say how much), or
    (None, reason).  Pure function (tests/test_host_cpu.py runs it on a synthetic trace)."""
    rows = sorted(rows)
    is_ad = [("adamw" in x[2] and "tail" not in x[2]) for x in rows]
    ends = [i for i in range(len(rows)) if is_ad[i] and (i + 1 == len(rows) or not is_ad[i + 1])]      # last sweep launch of every step
    if len(ends) < steps + 1:
        return None, "trace too short (%d optimizer sweeps)" % len(ends)
    seg = rows[ends[-steps - 1] + 1: ends[-1] + 1]
    busy, cs, ce = 0, seg[0][0], seg[0][1]
    for s_, e_, _ in seg[1:]:
        if s_ > ce:
            busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    agg = {}
    for s_, e_, k in seg:
        q = agg.setdefault(k, [0, 0])
        q[0] += 1; q[1] += e_ - s_
    # what each GEMM symbol computed (one log line per launch of every enqueue pass)
    flops = {}
    swept, ridden = [], []
    Tt, Tp = B * L, (B * L + 63) // 64 * 64
    for line in gemm_log.splitlines():
        if line.startswith("[magbert adamw] n="):
            swept.append(int(line.split("=")[1]))
            continue
        if line.startswith("[magbert ride] params="):
            ridden.append(int(line.split("=")[1].split()[0]))
            continue
        if not line.startswith("[magbert gemm] "):
            continue
        w = line.split()
        kv = dict(t.split("=") for t in w[3:])
        fl, K = float(kv["flop"]), int(kv["K"])
        if K == Tp and Tp != Tt and "_tn_" in w[2]:
            fl *= Tt / Tp                  # weight gradients run over the zero-padded token rows: algorithmic FLOPs count T
        q = flops.setdefault(w[2], [0, 0.0])
        q[0] += 1; q[1] += fl
    # (rocprofv3 prints non-template kernels demangled, the library logs the mangled symbol: a second index by `mb::<function>`, kept only
    #  where it is unambiguous)
    flops_base, seen = {}, {}
    for name, q in flops.items():
        seen.setdefault(kernel_base(name), []).append(q)
    for b_, qs in seen.items():
        if len(qs) == 1:
            flops_base[b_] = qs[0]
    flops_lit = {kernel_literals(name): q for name, q in flops.items() if kernel_literals(name)}      # (instantiations of one template)
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    ks = sorted(agg.items(), key=lambda kv_: -kv_[1][1])
    table, roof = [], []
    for k, (n, t) in ks[:24]:
        short = k.split("(")[0][:110]
        row = {"kernel": short, "launches_per_step": round(n / steps, 2), "avg_us": round(t / n / 1e3, 2), "ms_per_step": round(t / steps / 1e6, 4)}
        table.append(row)
        us = t / n / 1e3
        f = flops.get(k) or flops.get(k.split("(")[0]) or flops_lit.get(kernel_literals(k)) or flops_base.get(kernel_base(k))
        if f:
            per = f[1] / f[0]
            rr = dict(row, bound="mfma", flop_per_launch=per, achieved=round(per / us * 1e-6, 1), peak=peak, unit="TFLOP/s",
                      frac=round(per / us * 1e-6 / peak, 4), gflop_per_step=round(per * n / steps * 1e-9, 2))
            if ridden and "grouped_tn" in k and n % steps == 0 and n // steps > 1:
                # the layers' grouped weight gradient carries AdamW riders in every launch but the FIRST of a step (nothing is final yet when
                # the top layer's runs): that launch is the kernel's own duration, the average above includes the riders' tail
                mine = [e_ - s_ for s_, e_, kk in seg if kk == k]
                first = [mine[i] for i in range(0, len(mine), n // steps)]
                us0 = sum(first) / len(first) / 1e3
                rr["rider_free_avg_us"] = round(us0, 2)
                rr["rider_free_frac"] = round(per / us0 * 1e-6 / peak, 4)
                rr["rider_note"] = "avg_us / frac include the AdamW riders of 11 of the 12 launches per step; rider_free_* = the step's first launch (top layer: no rider)"
            roof.append(rr)
        elif "adamw" in k and "tail" not in k:
            per_step = int(round(n / steps))                      # sweep launches per step
            n_swept = sum(swept[-per_step:]) if len(swept) >= per_step and per_step > 0 else n_update
            per = 28.0 * n_swept / max(1, per_step)               # read p, g, m, v; write p, m, v -- the step's launches share the sweep
            roof.append(dict(row, bound="hbm", algorithmic_bytes_per_launch=int(per), achieved=round(per / us * 1e-3, 1), peak=8000.0, unit="GB/s",
                             frac=round(per / us * 1e-3 / 8000.0, 4), parameters_swept_per_step=int(n_swept),
                             parameters_updated_by_riders_per_step=int(n_update - n_swept) if n_swept < n_update else 0))
    doc = {"workload": "%s B=%d L=%d %s" % (model, B, L, dtype), "replayed": False,
           "busy_ms_per_step": round(busy / steps / 1e6, 4), "kernels_per_step": round(len(seg) / steps, 1),
           "busy_ms_per_step_is": "the sum of kernel durations UNDER THE PROFILER (a few percent above the untraced step: ms_per_step is the step)",
           "kernels": table}
    if ridden:
        doc["adamw_riders"] = {"launches_logged": len(ridden), "parameters_per_launch": int(sum(ridden) / len(ridden)),
                               "note": "HF-AdamW of already-final layers as extra workgroups of the grouped weight-gradient launches and of the ffn1 / qkv / ffn2 dgrad launches (gemm_pn_ride_kernel, gemm2_ride_kernel); those launches' durations above include it"}
    gemm_ms = sum(x["ms_per_step"] for x in roof if x["bound"] == "mfma")
    gemm_gf = sum(x["gflop_per_step"] for x in roof if x["bound"] == "mfma")
    if gemm_ms > 0:
        doc["gemm_aggregate"] = {"ms_per_step": round(gemm_ms, 4), "gflop_per_step": round(gemm_gf, 1), "tflops": round(gemm_gf / gemm_ms, 1),
                                 "frac": round(gemm_gf / gemm_ms / peak, 4), "note": "every GEMM symbol of the trace, in-step durations"}
    return doc, roof


def pick_roofline(trace_doc, trace_roof):
    """`roofline` from a priced trace: among the symbols within 10 % of the largest time per step, the one FURTHEST BELOW its roof -- the
    conservative pick, and a stable one (the top symbols trade places from box to box; the others ride along in roofline_trace)."""
    lead = [c for c in trace_roof if c["ms_per_step"] >= 0.9 * trace_roof[0]["ms_per_step"]]
    top = dict(min(lead, key=lambda c: c["frac"]))
    top["dominant_by"] = "ms_per_step over all %d symbols of the in-run kernel trace (ties within 10 %% -> the lowest fraction of its roof): " % len(trace_doc["kernels"]) + \
                         ", ".join("%s %.3f ms (%.3f of %s peak)" % (c["kernel"][:48], c["ms_per_step"], c["frac"], c["bound"]) for c in trace_roof[:4])
    top["timing"] = "rocprofv3 kernel trace taken by this run (in-step, graph replay), average over %d launches" % round(top["launches_per_step"] * 10)
    top["traffic"] = None
    return top


def instep_trace(B, L, V, dtype, n_update, steps=10, warmup=3, model="bert"):
    """A rocprofv3 kernel trace of the step, taken by bench.py itself: tools/bin/step_bench (the torch-free driver of the same
    mb_bert_train_step call: prologue + one replayed hipGraph, batch gathered from pinned host memory) runs `steps` traced steps as a
    subprocess; MB_GEMM_LOG=1 makes the library print, per GEMM launch, the kernel symbol and the FLOPs it was launched with, so every
    symbol of the trace is priced against what it computed.  -> (per-kernel table like profiles/instep_kernels.json, roofline rows)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    sb = os.path.join(ROOT, "tools", "bin", "step_bench")
    if not prof or not os.path.exists(sb):
        return None, "rocprofv3 or tools/bin/step_bench missing"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        return None, "this process already runs under a profiler (no nested trace)"
    d = tempfile.mkdtemp(prefix="mb_trace_", dir="/tmp")
    cmd = [prof, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "sb", "--", sb, "--graph", "1", "--h2d", "2", "--steps", str(steps),
           "--warmup", str(warmup), "--batch", str(B), "--seq", str(L), "--visual", str(V), "--dtype", dtype] + (["--model", "xlnet"] if model == "xlnet" else [])
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", MB_GEMM_LOG="1"), capture_output=True, text=True, timeout=150)
    except Exception as ex:
        shutil.rmtree(d, ignore_errors=True)
        return None, "trace failed: %r" % (ex,)
    took = time.perf_counter() - t0
    path = None
    for dp_, _, fs in os.walk(d):
        for f in fs:
            if f.endswith("kernel_trace.csv"):
                path = os.path.join(dp_, f)
    if r.returncode != 0 or path is None:
        shutil.rmtree(d, ignore_errors=True)
        return None, "trace failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-300:])
    rows = []
    with open(path) as fh:
        for x in csv.DictReader(fh):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]))
    shutil.rmtree(d, ignore_errors=True)
    doc, roof = price_trace(rows, r.stderr or "", B, L, dtype, n_update, steps, model)
    if doc is None:
        return None, roof
    doc["source"] = ("rocprofv3 --kernel-trace run BY THIS bench.py invocation over tools/bin/step_bench --graph 1 --h2d 2 (the same "
                     "mb_%s_train_step call, torch-free), last %d steps; %.1f s" % (model, steps, took))
    return doc, roof


def cpu_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    phys = None
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "physical_cores": phys}


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)                     # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local if torch.cuda.device_count() > local else 0)
    import torch.distributed as dist
    force_dp = world == 1 and os.environ.get("MB_DP_FORCE") == "1"     # 1-rank RCCL group: the whole DP code path on one GPU
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("MB_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda", local)} if os.environ.get("MB_DIST_BACKEND", "nccl") == "nccl" else {}))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))

    if a.roofline_only:
        rl = gemm_roofline(a.dtype, a.batch * a.seq)
        for r in rl:
            print("%-52s %8.2f us %8.1f TF/s" % (r["kernel"], r["avg_us"], r["tflops"]))
        print("per-layer GEMM time %.1f us" % sum(r["avg_us"] for r in rl))
        return
    from bert_multimodal_transformer_amd import (AdamW, BertConfig, MAG_BertForSequenceClassification, MultimodalConfig,
                                                 get_linear_schedule_with_warmup)
    from bert_multimodal_transformer_amd.distributed import DataParallel
    from bert_multimodal_transformer_amd.global_configs import DATASET_DIMS
    from bert_multimodal_transformer_amd.multimodal_driver import optimizer_grouped_parameters
    from bert_multimodal_transformer_amd.prefetch import PinnedBatchRing

    V, A = DATASET_DIMS[a.dataset]["visual_dim"], DATASET_DIMS[a.dataset]["acoustic_dim"]
    B, L = a.batch, a.seq
    torch.manual_seed(1234)        # same init on every rank (then broadcast anyway)
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.model == "xlnet":
        from bert_multimodal_transformer_amd import MAG_XLNetForSequenceClassification, XLNetConfig
        model = MAG_XLNetForSequenceClassification(XLNetConfig(num_labels=1), MultimodalConfig(1.0, 0.5), visual_dim=V,
                                                   acoustic_dim=A, compute_dtype=cdt)
    else:
        model = MAG_BertForSequenceClassification(BertConfig(num_labels=1), MultimodalConfig(1.0, 0.5), visual_dim=V,
                                                  acoustic_dim=A, compute_dtype=cdt)
    opt = AdamW(optimizer_grouped_parameters(model), lr=1e-5)
    total_steps = a.steps + a.warmup
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=0.1 * 1040, num_training_steps=1040)
    dp = None
    if world > 1 or force_dp:
        dp = DataParallel(model, opt)
        dp.broadcast_parameters(0)
    model.train()
    nb = 8
    batches = make_batches(nb, B, L, V, A, seed=1234 + rank, layout=a.model)      # pinned host tensors, as a DataLoader yields them
    dev = torch.device("cuda", torch.cuda.current_device())
    dp_call = dp is not None and dp.fused_ready() and opt.flat_step_args(model._core, allow_dp=True) is not None
    single_call = dp_call or (model._core.fused_step_blocker() is None and opt.flat_step_args(model._core) is not None)
    use_graph = None if not single_call else ((None if dp_call else True) if a.graph else "launches")
    graph_on = single_call and bool(a.graph)
    rccl_ranks = None
    if dp_call:                                      # the C-side exchange object is created collectively, at the latest here
        model._core._ensure(B, L)
        if dp.get_comm(B * L) is None:               # (RCCL could not be loaded / initialised: the Python-driven exchange runs)
            dp_call = single_call = graph_on = False
            use_graph = None
    if dp_call:
        # proof that the C-side communicator spans the ranks: an all-reduce(sum) of ones through mb_comm_all_reduce (the call the
        # step's exchange is made of) on the comm stream, in place on the head of the (still all-zero) flat gradient buffer
        from bert_multimodal_transformer_amd import _lib as _l
        g = model._core.grads
        g[:16].fill_(1.0)
        torch.cuda.synchronize()
        _l.check(_l.lib().mb_comm_all_reduce(dp.comm.handle, g.data_ptr(), 16, dp.comm.stream.cuda_stream))
        dp.comm.stream.synchronize()
        rccl_ranks = int(round(float(g[:16].sum().item()) / 16.0))
        g[:16].zero_()
        torch.cuda.synchronize()
        if rccl_ranks != world:
            raise SystemExit("bench.py: the gradient exchange spans %d rank(s), WORLD_SIZE is %d" % (rccl_ranks, world))

    def host_batches(n, start=0):
        for i in range(n):
            yield batches[(start + i) % nb]

    ring = PinnedBatchRing(None, dev)            # one ring (4 pinned blocks) for the whole run

    def run(n, start, events=None):
        """n optimizer steps exactly as train_epoch runs them: the batch comes from the HOST (packed into a pinned block that
        the step's gather launch reads across PCIe: multimodal_driver.py:359), forward + MSE + backward (+ all-reduce) + AdamW +
        schedule + zero_grad."""
        ring.loader = host_batches(n, start)
        for i, batch in enumerate(ring):
            ids, vis, aco, mask, seg, lab = batch
            model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=use_graph)
            sch.step()
            if events is not None:
                events[i + 1].record(torch.cuda.current_stream())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    resident = [tuple(t.to(dev) for t in b) for b in batches]

    def run_resident(n, start, events=None):
        """the same n optimizer steps with the batch tensors already resident in HBM when the step starts (the bench contract's timed region)"""
        for i in range(n):
            ids, vis, aco, mask, seg, lab = resident[(start + i) % nb]
            model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=use_graph)
            sch.step()
            if events is not None:
                events[i + 1].record(torch.cuda.current_stream())

    scope = model.stream_scope()          # the whole loop on one private HIP stream (see _MagBertBase.stream_scope)
    scope.__enter__()
    run(max(2, a.warmup // 2), 0)         # both forms are warmed (each captures its own graph)
    run_resident(a.warmup, 0)
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]      # created (and warmed) before the timed region
    for ev in evs:
        ev.record(torch.cuda.current_stream())
    fence()
    evs[0].record(torch.cuda.current_stream())
    t0 = time.perf_counter()
    run_resident(a.steps, a.warmup, evs)       # TIMED REGION: exactly K steps, inputs resident in HBM
    t_host = time.perf_counter() - t0          # host done enqueueing; the GPU may still be running
    fence()
    dt = time.perf_counter() - t0
    per_step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))
    # secondary figure: the loop as train_epoch runs it -- every batch comes from HOST memory (one pinned block, gathered across PCIe by the
    # step's first launch: multimodal_driver.py:359) -- the PCIe-inclusive rate (rounds 1-5 reported THIS as `value`)
    n2 = max(4, a.steps // 2)
    fence()
    t1 = time.perf_counter()
    run(n2, 0)
    fence()
    dt_h2d = (time.perf_counter() - t1) / n2
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # in-step duration of the two largest kernels of the step -- the per-layer grouped weight-gradient GEMM and the optimizer -- from
    # HIP events the engine records around them on the stream they run on (profiling events cannot live inside a captured graph:
    # these steps run the same kernel sequence launch by launch)
    wgrad_in_step_us = adamw_in_step_us = None
    comm_exposed_ms = None
    n3 = 0
    try:
        import ctypes as C
        from bert_multimodal_transformer_amd import _lib
        core = model._core
        fn = lambda name: getattr(_lib.lib(), "mb_%s_%s" % (core.kind, name))
        _lib.check(fn("set_profiling")(core.handle, 1))
        acc, acc_a = [], []
        for i in range(7):
            ids, vis, aco, mask, seg, lab = resident[i % nb]
            model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=use_graph)
            sch.step()
            torch.cuda.synchronize()
            v = C.c_float()
            _lib.check(fn("profile_wgrad_us")(core.handle, C.byref(v)))
            acc.append(v.value)
            if single_call:
                _lib.check(fn("profile_adamw_us")(core.handle, C.byref(v)))
                acc_a.append(v.value)
        _lib.check(fn("set_profiling")(core.handle, 0))
        wgrad_in_step_us = float(np.median(acc[1:]))
        adamw_in_step_us = float(np.median(acc_a[1:])) if acc_a else None
        n3 = 7
    except Exception as ex:          # MB_GROUP_WGRAD=0 (separate launches): no grouped kernel to time
        print("note: in-step kernel timing unavailable (%s)" % ex, file=sys.stderr)
    # host cost of ONE step call with an empty queue (host_enqueue_ms_per_step above is wall time of the enqueue loop: once the host
    # is a few steps ahead it blocks on the launch queue's depth and the figure approaches the GPU time -- it says nothing about the host)
    host_call = []
    for i in range(6):
        ids, vis, aco, mask, seg, lab = resident[i % nb]
        torch.cuda.synchronize()
        th = time.perf_counter()
        model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=use_graph)
        sch.step()
        host_call.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    host_call_ms = float(np.median(host_call[1:])) * 1e3
    n3 += 6
    comm_stats = None
    if dp is not None:
        if dp_call:          # five more steps with the comm object's timing events on: the compute stream's stalls on the exchange
            dp.comm.set_timing(True)
            ex = []
            for i in range(5):
                ids, vis, aco, mask, seg, lab = resident[i % nb]
                model.train_step(ids, vis, aco, mask, seg, lab, optimizer=opt, graph=use_graph)
                sch.step()
                ex.append(dp.exposed_ms())
            dp.comm.set_timing(False)
            comm_exposed_ms = float(np.median(ex))
            comm_stats = dp.comm.stats()
            n3 += 5
        else:
            comm_exposed_ms = dp.exposed_ms()
    scope.__exit__(None, None, None)
    loss = float(model.loss_running().item()) / max(1, total_steps + n2 + n3)
    value = world * B * a.steps / dt

    out = None
    if rank == 0:
        gflop = TRAIN_GFLOP_PER_SAMPLE_L50 if (L == 50 and V == 47 and a.model == "bert") else None
        if L == 128 and V == 35 and a.model == "bert":
            gflop = TRAIN_GFLOP_PER_SAMPLE_C5
        mname = "MAG-BERT" if a.model == "bert" else "MAG-XLNet"
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        q = lambda f: round(per_step_ms[min(len(per_step_ms) - 1, int(f * len(per_step_ms)))], 3)
        out = {"metric": "train samples/sec %s %s seq_len=%d" % (mname, a.dataset.upper(), L), "value": round(value, 2), "unit": "samples/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": mname + (" bert-base-uncased" if a.model == "bert" else " xlnet-base-cased") + ", %s dims (V=%d, A=%d), batch %d/GPU, seq_len %d, full "
                                      "optimizer step (fwd+MSE+bwd%s+HF-AdamW+schedule+zero_grad; inputs resident in HBM), dropout on, "
                                      "random-init weights" % (a.dataset.upper(), V, A, B, L, "+RCCL all-reduce" if world > 1 else ""),
                          "global_batch": world * B, "seq_len": L, "parallelism": "dp%d" % world,
                          "step_call": ("mb_%s_train_step%s, " % (model._core.kind, "_dp (gradient exchange issued from C between the graphs of the step)" if dp_call else "")
                                        + ("hipGraph replay" if graph_on else "stream launches")) if single_call else "passes driven from Python",
                          "h2d": "none inside the timed region (value_with_per_step_h2d: one pinned host block per batch, gathered across PCIe by the step's first launch)",
                          **({"grad_wire_dtype": "bf16" if dp.reducer.wire_dtype == torch.bfloat16 else "fp32"} if dp is not None else {})},
               "mean_loss": round(loss, 4), "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 3),
               "host_call_ms_per_step": round(host_call_ms, 3),
               "step_ms_median": q(0.5), "step_ms_p10": q(0.1), "step_ms_p90": q(0.9),
               "value_with_per_step_h2d": round(world * B / dt_h2d, 2),
               "value_is": "K timed steps with the batch tensors resident in HBM when the step starts (the bench contract); value_with_per_step_h2d = the same "
                           "steps fed from pinned host memory like train_epoch's `.to(DEVICE)` (what rounds 1-5 reported as `value`)"}
        if dp is not None:
            out["rccl_ranks"] = rccl_ranks
            out["devices_visible"] = torch.cuda.device_count()
            out["launched_by"] = "bench.py itself (one process per rank)" if os.environ.get("MB_BENCH_SPAWNED") == "1" else "external launcher"
        if comm_exposed_ms is not None:
            out["comm_exposed_ms"] = round(comm_exposed_ms, 4)
        if comm_stats is not None:
            out["comm_collectives_per_step"] = comm_stats[0]
            out["comm_mbytes_per_step"] = round(comm_stats[1] * 1e-6, 1)
            out["comm_backend"] = "RCCL called from C (mb_comm)" if dp.comm.backend == "nccl" else "torch.distributed callbacks (%s)" % dp.comm.backend
        if graph_on:
            out["graph_captures_replays"] = list(model._core.graph_stats())
        if gflop:
            out["step_tflops_algorithmic"] = round(value * gflop * 1e-3, 1)
            out["step_mfma_frac"] = round(value * gflop * 1e-3 * 0.984 / (peak * world), 4)
    if a.roofline and rank == 0:
        rl = gemm_roofline(a.dtype, B * L)
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        H_, I_, NL = 768, 3072, 12
        T = B * L
        # ---- the dominant GEMM: the per-layer grouped weight-gradient launch, timed INSIDE the step
        dom = max(rl, key=lambda r: r["avg_us"])
        us, fl = dom["avg_us"], dom["flop"]
        in_step = wgrad_in_step_us is not None and dom["kernel"].startswith("wgrad x4")
        if in_step:
            us = wgrad_in_step_us
            if a.model == "xlnet":      # MAG-XLNet groups seven problems: w2, w1, o, r (K = 2T position rows), q, k, v
                fl = 2.0 * T * (2 * H_ * I_ + 4 * H_ * H_) + 2.0 * (2 * T) * H_ * H_
        ach = fl / us * 1e-6
        mfma = {"bound": "mfma", "kernel": dom["kernel"] if a.model == "bert" else "wgrad x7 grouped (MAG-XLNet layer)",
                "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "avg_us": round(us, 2), "avg_us_standalone": dom["avg_us"] if a.model == "bert" else None, "flop_per_launch": fl,
                "launches_per_step": NL, "ms_per_step": round(us * NL * 1e-3, 4),
                "timing": "HIP events on the launch stream, in-step (median of 6 steps)" if in_step else
                          "HIP events on the launch stream, back-to-back launches"}
        # ---- the optimizer: two launches per step over the flat buffers, HBM-bound
        n_par = int(model._core.n_update_end)
        sh_n = int(model._core.sh_end - model._core.sh_begin) if a.dtype == "bf16" else 0
        hbm = None
        if adamw_in_step_us is not None:
            alg = 28.0 * n_par                       # read p, g, m, v; write p, m, v
            lazy = os.environ.get("MB_ADAMW_KEEP", "1") != "0" and os.environ.get("MB_WGRAD_OVERWRITE", "1") != "0"
            lazy_n = max(0, sh_n - 768 * 768) if lazy else 0      # the layers' GEMM weights: their gradient is not zeroed (pooler excluded)
            moved = alg + 2.0 * sh_n + 4.0 * (n_par - lazy_n)     # + bf16 shadow + the zeros of zero_grad
            ach_h = alg / adamw_in_step_us * 1e-6    # TB/s on the algorithmic bytes
            hbm = {"bound": "hbm", "kernel": "adamw (HF AdamW + zero_grad + bf16 weight shadow, 2 launches)", "achieved": round(ach_h * 1e3, 1),
                   "peak": 8000.0, "unit": "GB/s", "frac": round(ach_h / 8.0, 4), "traffic": None, "avg_us": round(adamw_in_step_us, 2),
                   "launches_per_step": 2, "ms_per_step": round(adamw_in_step_us * 1e-3, 4), "algorithmic_bytes": int(alg),
                   "bytes_moved_by_design": int(moved), "achieved_on_bytes_moved_gbs": round(moved / adamw_in_step_us * 1e-3, 1),
                   "timing": "HIP events on the launch stream, in-step (first launch issued -> second complete, median of 6 steps); the engine's "
                             "profiling mode turns the AdamW riders off, so this is the WHOLE update as the end-of-step sweep (the timed steps of `value` "
                             "run with riders: roofline_trace prices the sweep they leave)"}
        # HBM-side bytes per launch: PMC counters cannot be collected from inside this process; they are REPLAYED from the committed
        # rocprofv3 passes over this same step (profiles/pmc_traffic.json, written by scripts/gpu_artifacts.sh)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pm = json.load(fh)
            if pm.get("workload") == "%s B=%d L=%d %s" % (a.model, B, L, a.dtype):
                for blk, key in ((mfma, "wgrad_grouped"), (hbm, "adamw")):
                    e = pm.get("kernels", {}).get(key)
                    if blk is not None and e:
                        blk["traffic"] = e["fetch_bytes"] + e["write_bytes"]
                        blk["traffic_unit"] = "bytes/launch" if key != "adamw" else "bytes/step (the sweep launches of the committed pass: riders on, %s M parameters swept)" % round(e.get("algorithmic_bytes", 0) / 28e6, 1)
                        blk["traffic_replayed"] = True
                        blk["traffic_source"] = pm["source"]
                        blk.setdefault("algorithmic_bytes", e.get("algorithmic_bytes"))
        except Exception:
            pass
        # `roofline` = the kernel with the largest share of the step, chosen over EVERY symbol of a kernel trace this run takes itself
        # (tools/bin/step_bench under rocprofv3, ~10 s): durations from the trace, FLOPs from the library's launch log, AdamW's bytes
        # from SURVEY 8(d).  Without rocprofv3 / the driver binary: the two candidates timed by engine events above.
        trace_doc = trace_roof = None
        trace_note = "in-run trace only for the headline model (MAG-BERT)"
        if a.model == "bert" and world == 1 and os.environ.get("MB_BENCH_TRACE", "1") != "0":
            try:
                trace_doc, trace_roof = instep_trace(B, L, V, a.dtype, int(model._core.n_update_end))
            except Exception as ex:          # noqa: BLE001
                trace_doc, trace_roof = None, "trace failed: %r" % (ex,)
            if trace_doc is None:
                trace_note, trace_roof = trace_roof, None
        cands = [c for c in (mfma, hbm) if c is not None]
        if trace_roof:
            # three symbols share the top of the trace within a few percent (the 64 x 64 dgrad family, the grouped weight gradient, AdamW) and
            # trade places from box to box: among the symbols within 10 % of the largest time per step, `roofline` is the one FURTHEST BELOW
            # its roof -- the conservative pick, and a stable one (the others ride along in roofline_trace)
            top = pick_roofline(trace_doc, trace_roof)
            # HBM-side bytes per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE / WRITE_SIZE, separately) -- REPLAYED
            # from the committed passes over this same step (profiles/pmc_traffic.json, scripts/gpu_artifacts.sh)
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    pm_ = json.load(fh)
                if pm_.get("workload") == "%s B=%d L=%d %s" % (a.model, B, L, a.dtype):
                    e_ = None
                    if "adamw" in top["kernel"]:
                        e_ = pm_["kernels"].get("adamw")
                        if e_:
                            top["traffic"] = (e_["fetch_bytes"] + e_["write_bytes"]) // int(e_.get("launches_per_step", 2))
                    else:
                        e_ = next((v_ for k_, v_ in pm_.get("by_symbol", {}).items() if top["kernel"].startswith(k_) or k_.startswith(top["kernel"][:90])), None)
                        if e_:
                            top["traffic"] = e_["fetch_bytes"] + e_["write_bytes"]
                    if top["traffic"] is not None:
                        top["traffic_note"] = ("FETCH_SIZE x 2 + WRITE_SIZE: requests of the eight XCD-private L2s at the fabric (Infinity-Cache hits are "
                                               "counted): a GEMM whose tiles spread over 8 XCDs fetches every operand panel once PER XCD that needs it")
                        top["traffic_unit"] = "bytes/launch (mean over the symbol's launches)"
                        top["traffic_replayed"] = True
                        top["traffic_source"] = pm_["source"]
            except Exception:
                pass
            out["roofline"] = top
            out["roofline_trace"] = trace_roof[:8]
        else:
            top = max(cands, key=lambda c: c["ms_per_step"])
            out["roofline"] = dict(top, dominant_by="ms_per_step (launches x in-step duration) of the two engine-timed candidates (%s): " % trace_note +
                                   ", ".join("%s %.3f ms" % (c["kernel"].split(" (")[0], c["ms_per_step"]) for c in cands))
        out["roofline_mfma"] = mfma
        if hbm is not None:
            out["roofline_adamw"] = hbm
        tot_us = sum(r["avg_us"] for r in rl)
        tot_fl = sum(r["flop"] for r in rl)
        out["roofline_gemms"] = {"per_layer_us": round(tot_us, 1), "aggregate_tflops": round(tot_fl / tot_us * 1e-6, 1),
                                 "frac": round(tot_fl / tot_us * 1e-6 / peak, 4), "timing": "back-to-back launches (warm caches: an upper bound)",
                                 "note": "MAG-BERT's nine launches per layer" + ("" if a.model == "bert" else " (the encoder shapes both models share)"),
                                 "kernels": [{k: r[k] for k in ("kernel", "avg_us", "tflops")} for r in rl]}
        out["roofline_hbm"] = {"peak_tb_per_s": 8.0, "timing": "HIP events over rotating operand sets, median of 9 spans per kernel",
                               "kernels": hbm_roofline(a.dtype, B, L, V, A)}
        try:       # in-step per-kernel table: this run's own trace, else REPLAYED from the round's committed profiles
            if trace_doc is not None:
                ik = trace_doc
            else:
                with open(os.path.join(ROOT, "profiles", "instep_kernels.json")) as fh:
                    ik = json.load(fh)
                ik["replayed"] = True
                ik["busy_ms_per_step_is"] = "the sum of kernel durations UNDER THE PROFILER (a few percent above the untraced step: ms_per_step is the step)"
            if ik.get("workload") == "%s B=%d L=%d %s" % (a.model, B, L, a.dtype):
                out["instep_kernels"] = ik
                # achieved HBM rate of the row kernels INSIDE the step (same trace): algorithmic bytes / in-step duration
                es_, TH = (2 if a.dtype == "bf16" else 4), B * L * 768
                # bytes: what the kernel itself has to move (its inputs + outputs once); for the MAG gate that is NOT MAG's algorithmic
                # figure -- the gate re-reads the three GEMM pre-activation panels -- so MAG's forward is reported separately below
                alg = {"ln_fwd_kernel": 2 * TH * es_, "ln_bwd_kernel": 4 * TH * es_, "mag_gate_fwd_kernel": 8 * TH * es_,
                       "mag_gate_bwd_kernel": 15 * TH * es_, "embed_fwd_kernel": TH * (4 + es_), "embed_bwd_kernel": TH * (es_ + 4 + 4)}
                rows = []
                for k in ik.get("kernels", []):
                    for name, nbytes in alg.items():
                        if name in k["kernel"]:
                            rows.append({"kernel": name, "launches_per_step": k["launches_per_step"], "avg_us": k["avg_us"], "bytes": int(nbytes),
                                         "bytes_are": "the kernel's own inputs + outputs" if name.startswith("mag_gate") else "algorithmic",
                                         "tb_per_s": round(nbytes / k["avg_us"] * 1e-6, 3), "frac_of_8tbs": round(nbytes / k["avg_us"] * 1e-6 / 8.0, 4)})
                # MAG forward as a whole on SURVEY 8(d)'s algorithmic bytes: read e (bf16), visual, acoustic (fp32), write the output (bf16)
                # = 3,556 B/token at V = 47, A = 74 -- against the in-step time of everything MAG's forward launches (pack, 3 GEMMs, gate)
                mag_us = sum(k["avg_us"] * k["launches_per_step"] for k in ik.get("kernels", []) if "mag_gate_fwd" in k["kernel"] or "mag_pack" in k["kernel"])
                if mag_us > 0:
                    mag_bytes = B * L * (2 * 768 * es_ + 4 * V + 4 * A)
                    rows.append({"kernel": "MAG forward, gate + weight pack launches only (its three GEMMs are in the GEMM rows)", "avg_us": round(mag_us, 2),
                                 "bytes": int(mag_bytes), "bytes_are": "algorithmic, SURVEY 8(d): %d B/token" % (mag_bytes // (B * L)),
                                 "tb_per_s": round(mag_bytes / mag_us * 1e-6, 3), "frac_of_8tbs": round(mag_bytes / mag_us * 1e-6 / 8.0, 4)})
                out["roofline_hbm"]["instep_replayed" if ik.get("replayed") else "instep"] = rows
        except Exception:
            pass
    if a.secondary and rank == 0 and world == 1 and dp is None and a.model == "bert" and (B, L, a.dataset, a.dtype) == (48, 50, "mosi", "bf16"):
        del resident
        torch.cuda.empty_cache()
        out["secondary"] = []
        for kind, dataset, b2, l2 in (("xlnet", "mosi", 48, 50), ("bert", "mosei", 32, 128)):
            try:
                out["secondary"].append(secondary_workload(kind, dataset, b2, l2, dtype=a.dtype,
                                                           cpu_steps=(2 if (kind == "xlnet" and a.cpu_baseline) else 0)))
            except Exception as ex:
                out["secondary"].append({"metric": "%s %s B=%d L=%d" % (kind, dataset, b2, l2), "error": repr(ex)})
    if a.epoch and rank == 0 and world == 1 and dp is None and a.model == "bert" and (B, L, a.dataset, a.dtype) == (48, 50, "mosi", "bf16"):
        try:
            out.update(epoch_mode(a.dtype))
            # the share of train_epoch that is NOT its 26 full-size steps at the timed rate: the ragged step, the first-batch staging, the one host sync
            out["epoch_train_outside_full_steps_ms"] = round(out["epoch_train_ms"] - 26 * 48.0 / out["value_with_per_step_h2d"] * 1e3, 2)
        except Exception as ex:          # noqa: BLE001
            out["epoch_error"] = repr(ex)
    if a.cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(B, L, V, A, a.cpu_steps, a.model)
        out["cpu_baseline"].update(cpu_info())
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
