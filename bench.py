#!/usr/bin/env python
"""bench.py -- train samples/sec of MAG-BERT fine-tuning (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full optimizer step of /root/reference/multimodal_driver.py:354-388 on one minibatch per GPU:
forward (embeddings, MAG, 12 encoder layers, pooler, classifier) -> MSE -> backward -> (N>1: RCCL all-reduce of the flat
gradients, overlapped with the backward) -> HF-AdamW -> linear-warmup schedule -> zero_grad, with the six batch tensors
already resident in HBM (`value`); the reference's per-step H2D (multimodal_driver.py:359) is timed separately
(`value_with_h2d`).  Workload = BASELINE.json configs[1]: bert-base-uncased MAG-BERT, MOSI dims (V=47, A=74), B=48/GPU, L=50,
bf16 MFMA with fp32 master weights, dropout ON (0.1/0.1/MAG 0.5), synthetic batches in prepare_bert_input's layout,
random-init weights (no network).  Weak scaling: per-GPU batch fixed, global batch = 48*N.

Prints ONE JSON line (rank 0) with the contract's keys plus
  roofline     : the dominant kernel (the per-layer grouped weight-gradient GEMM) timed inside the step with HIP events on the
                 stream it runs on, its HBM-side traffic from the committed PMC pass (profiles/r01_pmc_step.md)
  cpu_baseline : the CPU oracle (oracle/mag_bert_ref.py, kind "port") timed on this box's host cores, same step
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
    p.add_argument("--cpu-steps", type=int, default=2)
    p.add_argument("--roofline", type=int, default=1)
    p.add_argument("--fused-optimizer", type=int, default=0,
                   help="N=1 only: AdamW for the encoder GEMM weights runs in the weight-gradient GEMM epilogue (same arithmetic; "
                        "measured neutral, so the default keeps the same code path at every N)")
    p.add_argument("--pipelined-optimizer", type=int, default=0,
                   help="AdamW runs on the engine's optimizer stream, chunk by chunk, under the next forward (same arithmetic; "
                        "measured neutral: the forward GEMMs are memory-latency-bound and slow down under the HBM-saturating update)")
    p.add_argument("--roofline-only", type=int, default=0, help="skip the training loop, print the GEMM table only")
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
    torch.set_num_threads(cores)
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
    return {"value": B / t, "unit": "samples/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d timed optimizer steps (1 warmup) of the same B=%d L=%d MAG-%s step, fp32, dropout on, "
                      "oracle/mag_%s_ref.py + HF-AdamW, median step %.2f s" % (steps, B, L, kind.upper(), kind, t)}


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
    T = (T + 63) // 64 * 64        # the engine zero-pads the token dimension to the GEMM k-tile
    st = torch.cuda.current_stream()
    g = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(tdt)
    x, xi, w_qkv, w_o, w1, w2 = g(T, H), g(T, I), g(3 * H, H), g(H, H), g(I, H), g(H, I)
    dqkv = g(T, 3 * H)
    bias = torch.zeros(3 * I, device=dev)
    out = torch.empty(T, I, dtype=tdt, device=dev); out2 = torch.empty(T, I, dtype=tdt, device=dev)
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
        # the layer's four weight gradients are ONE grouped launch in the engine (csrc/gemm.hip gemm2_grouped_tn_kernel)
        ("wgrad x4  grouped [768x3072|3072x768|768x768|2304x768] K=T", 1, "grouped", None, 0, 0, T, None, 0, None, 0, None, 1),
    ]
    gshape = [(H, I), (I, H), (H, H), (3 * H, H)]
    gY, gX = [x, xi, x, dqkv], [xi, x, x, x]
    gW = [torch.zeros(m, n, device=dev) for m, n in gshape]
    ia = lambda v: (C.c_int * 4)(*v)
    pa = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
    gM, gN, gpY, gpX, gpW = ia([m for m, n in gshape]), ia([n for m, n in gshape]), pa(gY), pa(gX), pa(gW)
    gtile = int(os.environ.get("MB_GROUP_WGRAD", "128")) or 128
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
        fl = 2.0 * M * N * K if layout != "grouped" else sum(2.0 * m * n * K for m, n in gshape)
        res.append({"kernel": name, "M": M, "N": N, "K": K, "avg_us": round(us, 2), "tflops": round(fl / us * 1e-6, 1),
                    "flop": fl})
    return res


def main():
    a = parse()
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
    if a.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d (using WORLD_SIZE)" % (a.gpus, world), file=sys.stderr)

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
    fused_opt = bool(a.fused_optimizer) and world == 1 and opt.enable_fused_backward(model)
    piped_opt = bool(a.pipelined_optimizer) and not fused_opt and opt.enable_pipelined_step(model)
    model.train()
    nb = 8
    batches = make_batches(nb, B, L, V, A, seed=1234 + rank, layout=a.model)
    dev = torch.device("cuda", torch.cuda.current_device())

    # `value` is quoted with the inputs already resident in HBM; the H2D-inclusive rate (the reference moves every batch
    # inside the loop, multimodal_driver.py:359) is measured by a second, shorter loop and reported as value_with_h2d
    resident = [tuple(t.to(dev) for t in b) for b in batches]

    def step(i, h2d=False):
        if h2d:
            batch = tuple(t.to(dev, non_blocking=True) for t in batches[i % nb])
        else:
            batch = resident[i % nb]
        ids, vis, aco, mask, seg, lab = batch
        model.training_step(ids, vis, aco, mask, seg, lab)
        opt.step(); sch.step(); opt.zero_grad()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    scope = model.stream_scope()          # the whole loop on one private HIP stream (see _MagBertBase.stream_scope)
    scope.__enter__()
    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    t_host = time.perf_counter() - t0          # host done enqueueing; the GPU may still be running
    fence()
    dt = time.perf_counter() - t0
    n2 = max(2, a.steps // 4)
    t1 = time.perf_counter()
    for i in range(n2):
        step(a.warmup + a.steps + i, h2d=True)
    fence()
    dt_h2d = (time.perf_counter() - t1) / n2
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # in-step duration of the dominant kernel (grouped weight-gradient GEMM): HIP events on the engine's side stream
    wgrad_in_step_us = None
    try:
        import ctypes as C
        from bert_multimodal_transformer_amd import _lib
        core = model._core
        if core.kind != "bert":
            raise RuntimeError("the MAG-XLNet engine launches its weight gradients one by one on the main stream")
        _lib.check(_lib.lib().mb_bert_set_profiling(core.handle, 1))
        acc = []
        for i in range(6):
            step(i)
            torch.cuda.synchronize()
            v = C.c_float()
            _lib.check(_lib.lib().mb_bert_profile_wgrad_us(core.handle, C.byref(v)))
            acc.append(v.value)
        _lib.check(_lib.lib().mb_bert_set_profiling(core.handle, 0))
        wgrad_in_step_us = float(np.mean(acc[1:]))
        n2 += 6
    except Exception as ex:          # MB_GROUP_WGRAD=0 (four separate launches): no grouped kernel to time
        print("note: in-step wgrad timing unavailable (%s)" % ex, file=sys.stderr)
    scope.__exit__(None, None, None)
    loss = float(model.loss_running().item()) / max(1, total_steps + n2)
    value = world * B * a.steps / dt

    out = None
    if rank == 0:
        gflop = TRAIN_GFLOP_PER_SAMPLE_L50 if (L == 50 and V == 47 and a.model == "bert") else None
        mname = "MAG-BERT" if a.model == "bert" else "MAG-XLNet"
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        out = {"metric": "train samples/sec %s MOSI seq_len=%d" % (mname, L), "value": round(value, 2), "unit": "samples/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": mname + (" bert-base-uncased" if a.model == "bert" else " xlnet-base-cased") + ", %s dims (V=%d, A=%d), batch %d/GPU, seq_len %d, full "
                                      "optimizer step (fwd+MSE+bwd%s+HF-AdamW%s+schedule; inputs resident in HBM), dropout on, random-init weights"
                                      % (a.dataset.upper(), V, A, B, L, "+RCCL all-reduce" if world > 1 else "",
                                         " (encoder weights updated in the wgrad epilogue)" if fused_opt else
                                         (" (pipelined under the next forward)" if piped_opt else "")),
                          "global_batch": world * B, "seq_len": L, "parallelism": "dp%d" % world,
                          **({"grad_wire_dtype": "bf16" if dp.reducer.wire_dtype == torch.bfloat16 else "fp32"} if dp is not None else {})},
               "mean_loss": round(loss, 4), "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 3),
               "value_with_h2d": round(world * B / dt_h2d, 2)}
        if gflop:
            out["step_tflops_algorithmic"] = round(value * gflop * 1e-3, 1)
            out["step_mfma_frac"] = round(value * gflop * 1e-3 * 0.984 / (peak * world), 4)
    if a.roofline and rank == 0:
        rl = gemm_roofline(a.dtype, B * L)
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        # dominant kernel = the GEMM with the largest time per training step (each runs once per layer per step)
        dom = max(rl, key=lambda r: r["avg_us"])
        us = dom["avg_us"]
        if wgrad_in_step_us is not None and dom["kernel"].startswith("wgrad x4"):
            us = wgrad_in_step_us           # duration inside the training step (runs concurrently with the dgrad chain)
        ach = dom["flop"] / us * 1e-6
        out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(ach, 1), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                           "avg_us": round(us, 2), "avg_us_standalone": dom["avg_us"], "flop_per_launch": dom["flop"],
                           "timing": "HIP events on the launch stream, in-step" if us is not dom["avg_us"] else
                                     "HIP events on the launch stream, back-to-back launches"}
        # HBM-side bytes per launch of that kernel: PMC counters cannot be collected from inside this process; the figure
        # comes from the committed rocprofv3 --pmc passes over this same command (profiles/r01_pmc_step.md), else null
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                pm = json.load(fh)
            if dom["kernel"].startswith(pm["kernel"]) and a.dtype == "bf16" and B == 48 and L == 50 and a.model == "bert":
                out["roofline"]["traffic"] = pm["fetch_bytes"] + pm["write_bytes"]
                out["roofline"]["traffic_unit"] = "bytes/launch"
                out["roofline"]["traffic_source"] = pm["source"]
                out["roofline"]["algorithmic_bytes"] = 116391936
        except Exception:
            pass
        if a.model != "bert":
            out["roofline"]["note"] = ("GEMM table of the encoder shapes both models share, back-to-back; MAG-XLNet's own grouped "
                                       "weight-gradient launch has 7 problems and is not timed separately")
        tot_us = sum(r["avg_us"] for r in rl)
        tot_fl = sum(r["flop"] for r in rl)
        out["roofline_gemms"] = {"per_layer_us": round(tot_us, 1), "aggregate_tflops": round(tot_fl / tot_us * 1e-6, 1),
                                 "frac": round(tot_fl / tot_us * 1e-6 / peak, 4),
                                 "kernels": [{k: r[k] for k in ("kernel", "avg_us", "tflops")} for r in rl]}
    if a.cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(B, L, V, A, a.cpu_steps, a.model)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
