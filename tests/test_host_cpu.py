"""CPU: host logic -- C-ABI symbols, parameter layout vs the reference's state dict, dropout-key twin,
feature conversion (G7), metrics (G9), data-parallel reducer over gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bert_multimodal_transformer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "magbert_hip.h")).read()
    declared = set(re.findall(r"\b(mb_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    h = _lib.lib()                      # loads the .so and binds every prototype (AttributeError if one is missing)
    assert h.mb_version() >= 100
    assert b"shape" in h.mb_error_string(1001)


def test_dropkey_twin_matches_library():
    from bert_multimodal_transformer_amd import _lib, rng
    for seed, step, site, p in ((0, 1, 0, 0.1), (1234, 77, 18, 0.5), (2**40 + 5, 3, 2, 0.1), (9, 9, 61, 0.0)):
        k = _lib.make_dropkey(seed, step, site, p)
        assert (k.k0, k.k1, k.thresh) == rng.make_key(seed, step, site, p)[:3]
        assert abs(k.scale - rng.make_key(seed, step, site, p)[3]) < 1e-7
    m = rng.keep_mult(200000, rng.make_key(5, 1, 1, 0.5))
    assert abs(float((m == 0).mean()) - 0.5) < 0.01 and set(np.unique(m)) == {0.0, 2.0}
    m = rng.keep_mult(200000, rng.make_key(5, 1, 16, 0.1))
    assert abs(float((m == 0).mean()) - 0.1) < 0.005


def _engine_table(V=47, dtype=0):
    from bert_multimodal_transformer_amd import _lib
    L = _lib.lib()
    cfg = _lib.BertEngineConfig(30522, 768, 12, 12, 3072, 512, 2, 1, V, 74, 0, 1e-12, 1e-5, 1.0, 0.1, 0.1, 0.5, dtype, 4, 50)
    h = C.c_void_p()
    _lib.check(L.mb_bert_create(C.byref(cfg), C.byref(h)))
    name = C.create_string_buffer(160)
    off, numel, ndim, decay = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
    shape = (C.c_int64 * 4)()
    rows = []
    for i in range(L.mb_bert_num_tensors(h)):
        _lib.check(L.mb_bert_tensor_info(h, i, name, 160, C.byref(off), C.byref(numel), C.byref(ndim), shape, C.byref(decay)))
        rows.append((name.value.decode(), off.value, numel.value, tuple(shape[k] for k in range(ndim.value)), decay.value))
    info = dict(n=L.mb_bert_param_count(h), n_decay=L.mb_bert_decay_count(h), ws=L.mb_bert_workspace_bytes(h))
    ranges = []
    offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
    for s in range(14):
        k = L.mb_bert_stage_grad_ranges(h, s, offs, lens, 8)
        ranges += [(offs[i], lens[i]) for i in range(k)]
    L.mb_bert_destroy(h)
    return rows, info, ranges


@pytest.mark.parametrize("V", [47, 35])
def test_flat_layout_matches_reference_state_dict(V):
    from oracle import mag_bert_ref as R
    rows, info, ranges = _engine_table(V)
    ref = R.MAG_BertForSequenceClassification(R.BertConfigLite(), R.MultimodalConfig(1.0, 0.5), V, 74)
    want = {n: tuple(p.shape) for n, p in ref.named_parameters()}
    got = {r[0]: r[3] for r in rows}
    assert got == want                                     # 211 tensors, reference names + shapes
    assert sum(r[2] for r in rows) == sum(p.numel() for p in ref.parameters())     # 110,853,121 @ V=47
    # decay flag == the driver's substring rule (multimodal_driver.py:329-343)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    for name, off, numel, shape, decay in rows:
        assert bool(decay) == (not any(nd in name for nd in no_decay)), name
        assert off % 64 == 0
        assert (off < info["n_decay"]) == bool(decay)
    # no overlap, q/k/v contiguous (fused QKV operand)
    spans = sorted((r[1], r[1] + r[2]) for r in rows)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    byname = {r[0]: r for r in rows}
    for l in (0, 11):
        q, k, v = (byname["bert.encoder.layer.%d.attention.self.%s.weight" % (l, t)] for t in ("query", "key", "value"))
        assert k[1] == q[1] + q[2] and v[1] == k[1] + k[2]
    # data-parallel stage ranges tile the flat buffer exactly once
    cover = np.zeros(info["n"], np.int8)
    for off, n in ranges:
        cover[off:off + n] += 1
    assert cover.min() == 1 and cover.max() == 1


def test_xlnet_flat_layout_matches_reference_state_dict():
    """MAG-XLNet engine layout (host-side, no GPU): reference names / shapes, the driver's decay rule (XLNet's `layer_norm.weight`
    IS decayed, `r_*_bias` is not), `mask_emb` frozen, q|k|v contiguous, stage ranges tile the trainable part exactly once."""
    from bert_multimodal_transformer_amd import _lib
    from oracle import mag_xlnet_ref as X
    L = _lib.lib()
    cfg = _lib.XlnetEngineConfig(32000, 768, 12, 12, 3072, 1, 47, 74, 1, 1e-12, 1e-5, 1.0, 0.1, 0.1, 0.5, _lib.DT_BF16, 48, 50)
    h = C.c_void_p()
    _lib.check(L.mb_xlnet_create(C.byref(cfg), C.byref(h)))
    name = C.create_string_buffer(160)
    off, numel, ndim, decay = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
    shape = (C.c_int64 * 4)()
    rows = []
    for i in range(L.mb_xlnet_num_tensors(h)):
        _lib.check(L.mb_xlnet_tensor_info(h, i, name, 160, C.byref(off), C.byref(numel), C.byref(ndim), shape, C.byref(decay)))
        rows.append((name.value.decode(), off.value, numel.value, tuple(shape[k] for k in range(ndim.value)), decay.value))
    n_params, n_decay = L.mb_xlnet_param_count(h), L.mb_xlnet_decay_count(h)
    ranges = []
    for s in range(12 + 2):
        offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
        k = L.mb_xlnet_stage_grad_ranges(h, s, offs, lens, 8)
        assert 0 <= k <= 8
        ranges += [(offs[i], lens[i]) for i in range(k)]
    L.mb_xlnet_destroy(h)
    ref = X.MAG_XLNetForSequenceClassification(X.XLNetConfigLite(), X.MultimodalConfig(1.0, 0.5), 47, 74)
    assert {r[0]: r[3] for r in rows} == {n: tuple(p.shape) for n, p in ref.named_parameters()}
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    byname = {r[0]: r for r in rows}
    for nme, o, n, shp, d in rows:
        assert o % 64 == 0
        if nme == "transformer.mask_emb":
            assert d == 2                                     # never receives a gradient (xlnet.py:29): frozen slot
            continue
        assert bool(d) == (not any(nd in nme for nd in no_decay)), nme
        assert (o < n_decay) == bool(d), nme
    assert byname["transformer.layer.3.rel_attn.layer_norm.weight"][4] == 1 and byname["transformer.layer.3.rel_attn.r_r_bias"][4] == 0
    for l in (0, 11):
        q, k, v = (byname["transformer.layer.%d.rel_attn.%s" % (l, t)] for t in ("q", "k", "v"))
        assert k[1] == q[1] + q[2] and v[1] == k[1] + k[2]
    spans = sorted((r[1], r[1] + r[2]) for r in rows)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= n_params
    cover = np.zeros(n_params, np.int8)
    for o, n in ranges:
        cover[o:o + n] += 1
    for nme, o, n, shp, d in rows:
        if d != 2:
            assert cover[o:o + n].min() == 1 and cover[o:o + n].max() == 1, nme     # every trainable tensor reduced exactly once
    assert cover.max() == 1


def test_feature_conversion_matches_reference(golden):
    g = golden["g7g9_features_metrics"]
    from oracle import weights
    from oracle.make_golden import FakeTokenizer
    from bert_multimodal_transformer_amd import multimodal_driver as D
    words_sets = [["hello", "world"], ["a"] * 3, ["abcdefgh"] * 30, ["xy"] * 48, ["pq"] * 49]
    for kind, model_name in (("bert", "bert-base-uncased"), ("xlnet", "xlnet-base-cased")):
        D.args = D.parse_args(["--model", model_name, "--max_seq_length", "50", "--seed", "1"])
        examples = []
        for i, words in enumerate(words_sets):
            n = len(words)
            examples.append(((words, weights.uniform("feat.v%d" % i, (n, 47)), weights.uniform("feat.a%d" % i, (n, 74))),
                             float(i) - 1.5, "seg%d" % i))
        feats = D.convert_to_features(examples, 50, FakeTokenizer(kind))
        assert np.array_equal(np.array([f.input_ids for f in feats], np.int64), g[kind + "/input_ids"])      # bit-exact
        assert np.array_equal(np.array([f.input_mask for f in feats], np.int64), g[kind + "/input_mask"])
        assert np.array_equal(np.array([f.segment_ids for f in feats], np.int64), g[kind + "/segment_ids"])
        assert np.array_equal(np.array([f.visual for f in feats], np.float32), g[kind + "/visual"])
        assert np.array_equal(np.array([f.acoustic for f in feats], np.float32), g[kind + "/acoustic"])
        ds = D.features_to_dataset(feats)
        assert [t.dtype for t in ds.tensors] == [torch.long, torch.float, torch.float, torch.long, torch.long, torch.float]


def test_metrics_match_reference(golden):
    g = golden["g7g9_features_metrics"]
    from bert_multimodal_transformer_amd import multimodal_driver as D
    for uz in (False, True):
        got = D.score_predictions(g["score/preds"].copy(), g["score/labels"].copy(), use_zero=uz)
        np.testing.assert_allclose(np.array(got, np.float64), g["score/use_zero_%d" % int(uz)], rtol=1e-12)


def test_synthetic_dataset_layout():
    from bert_multimodal_transformer_amd import multimodal_driver as D
    ds = D.synthetic_dataset(64, 50, 47, 74)
    ids, vis, aco, mask, seg, lab = ds.tensors
    n = mask.sum(1)
    assert (ids[:, 0] == 101).all() and (ids[torch.arange(64), n - 1] == 102).all()
    assert ((ids == 0) == (mask == 0)).all() and (seg == 0).all()
    word = (mask == 1) & (ids != 101) & (ids != 102)
    assert (vis[~word] == 0).all() and (aco[~word] == 0).all() and (vis[word].abs().sum(-1) > 0).all()


def test_synthetic_dataset_xlnet_layout_matches_prepare_xlnet_input():
    """the synthetic XLNet samples have exactly the integer layout prepare_xlnet_input produces (multimodal_driver.py:176-205)"""
    from bert_multimodal_transformer_amd import multimodal_driver as D
    L = 50
    ds = D.synthetic_dataset(32, L, 47, 74, layout="xlnet")
    ids, vis, aco, mask, seg, lab = ds.tensors
    n = mask.sum(1)                                         # tokens + <sep> + <cls>
    for b in range(32):
        pad = L - int(n[b])
        assert (ids[b, :pad] == 5).all() and (mask[b, :pad] == 0).all() and (seg[b, :pad] == 3).all()
        assert ids[b, L - 2] == 4 and ids[b, L - 1] == 3 and seg[b, L - 1] == 2 and (seg[b, pad:L - 1] == 0).all()
        assert (vis[b, :pad] == 0).all() and (vis[b, L - 2:] == 0).all() and (aco[b, :pad] == 0).all() and (aco[b, L - 2:] == 0).all()
        assert (vis[b, pad:L - 2].abs().sum(-1) > 0).all()


def test_shard_indices_cover_dataset_once():
    from bert_multimodal_transformer_amd.distributed import shard_indices
    n, world, bs = 1281, 8, 48
    per_rank = [shard_indices(n, r, world, bs, seed=3, epoch=0) for r in range(world)]
    assert len({len(p) for p in per_rank}) == 1                     # same number of steps on every rank
    flat = [i for p in per_rank for b in p for i in b]
    assert sorted(flat) == list(range(n))                           # each sample exactly once
    assert all(len(b) == bs for p in per_rank for b in p[:-1])


def test_sharded_batch_sampler_epochs_and_coverage():
    """the driver's per-rank sampler (torchrun launch): all ranks see the same number of batches, together they cover the
    dataset exactly once per epoch, and the shuffle changes from epoch to epoch"""
    from bert_multimodal_transformer_amd.multimodal_driver import ShardedBatchSampler
    n, world, bs = 1000, 4, 48
    samplers = [ShardedBatchSampler(n, r, world, bs, 7) for r in range(world)]
    assert len({len(s) for s in samplers}) == 1
    e0 = [list(s) for s in samplers]
    e1 = [list(s) for s in samplers]
    for ep in (e0, e1):
        flat = [i for per_rank in ep for b in per_rank for i in b]
        assert sorted(flat) == list(range(n))
        assert len({len(per_rank) for per_rank in ep}) == 1
    assert e0[0][0] != e1[0][0]


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from bert_multimodal_transformer_amd.distributed import GradReducer
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
n = 10000
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
red = GradReducer(g)
red.reduce_ranges([(0, 4096), (4096, 1000)])      # two "stages"
red.reduce_ranges([(5096, n - 5096)])             # the tail piece
red.wait()
want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(g, want), (g[:4], want[:4])
# averaging is folded into the optimizer: grad_scale = 1/world
assert abs((g * (1.0 / world))[10].item() - 10 * 1.5) < 1e-6
dist.destroy_process_group()
print("OK", rank)
'''


def test_grad_reducer_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_DDP_WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REPO_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()[-2000:]


def _stub_core(kind):
    """engine layout through the C ABI (host-only calls) wrapped like bert._Core, for distributed.stage_plan"""
    from bert_multimodal_transformer_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    if kind == "bert":
        cfg = _lib.BertEngineConfig(30522, 768, 12, 12, 3072, 512, 2, 1, 47, 74, 0, 1e-12, 1e-5, 1.0, 0.1, 0.1, 0.5, _lib.DT_BF16, 4, 50)
        _lib.check(L.mb_bert_create(C.byref(cfg), C.byref(h)))
    else:
        cfg = _lib.XlnetEngineConfig(32000, 768, 12, 12, 3072, 1, 47, 74, 1, 1e-12, 1e-5, 1.0, 0.1, 0.1, 0.5, _lib.DT_BF16, 4, 50)
        _lib.check(L.mb_xlnet_create(C.byref(cfg), C.byref(h)))
    fn = lambda n: getattr(L, "mb_%s_%s" % (kind, n))

    class Core(object):
        n_layers = 12
        n_params = fn("param_count")(h)
        n_decay = fn("decay_count")(h)

        def stage_ranges(self, stage):
            offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
            k = fn("stage_grad_ranges")(h, stage, offs, lens, 8)
            return [(offs[i], lens[i]) for i in range(k)]

    name = C.create_string_buffer(160)
    off, numel, ndim, decay = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
    shape = (C.c_int64 * 4)()
    rows = []
    for i in range(fn("num_tensors")(h)):
        _lib.check(fn("tensor_info")(h, i, name, 160, C.byref(off), C.byref(numel), C.byref(ndim), shape, C.byref(decay)))
        rows.append((name.value.decode(), off.value, numel.value, decay.value))
    return Core(), rows


@pytest.mark.parametrize("kind", ["bert", "xlnet"])
def test_dp_plan_reduces_every_trainable_element_exactly_once(kind):
    """distributed.stage_plan: the per-stage large ranges plus the remainder cover every trainable element ONCE for both
    engines (MAG-XLNet reports its whole no-decay block as one large range of its last stage: round 1 reduced it twice)."""
    from bert_multimodal_transformer_amd.distributed import stage_plan
    core, rows = _stub_core(kind)
    plan, tail = stage_plan(core)
    assert len(plan) == 14
    cover = np.zeros(core.n_params, np.int16)
    for rng_ in [r for big in plan for r in big] + list(tail):
        cover[rng_[0]: rng_[0] + rng_[1]] += 1
    for name, off, numel, decay in rows:
        want_min = 0 if decay == 2 else 1               # frozen slots (XLNet mask_emb) may or may not travel; never twice
        assert cover[off: off + numel].min() >= want_min and cover[off: off + numel].max() <= 1, name
    assert cover.max() == 1


_ROWS_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from bert_multimodal_transformer_amd.distributed import exchange_embedding_rows
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
V, H, cap = 997, 24, 64
for trial in range(4):
    g = torch.Generator().manual_seed(100 * trial + rank)
    n = cap if trial % 2 == 0 else cap - 7 - rank           # also fewer ids than the agreed capacity (ragged last batch)
    ids = torch.randint(0, 40 if trial < 2 else V, (n,), generator=g)      # many repeats / almost none
    table = torch.zeros(V, H)
    table.index_add_(0, ids, torch.randn(n, H, generator=g))                # what the embedding backward leaves: rows of ids only
    dense = table.clone()
    dist.all_reduce(dense)
    got = exchange_embedding_rows(table.clone(), ids, cap)
    assert torch.allclose(got, dense, rtol=0, atol=1e-6), float((got - dense).abs().max())
    both = [torch.empty_like(got) for _ in range(world)]
    dist.all_gather(both, got)
    assert torch.equal(both[0], both[1])                                    # replicas bit-identical
dist.destroy_process_group()
print("OK", rank)
"""


def test_embedding_row_exchange_equals_dense_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "rows.py"
    script.write_text(_ROWS_WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REPO_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()[-2000:]


def test_no_cpu_fallback():
    from bert_multimodal_transformer_amd import MAG, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = MAG(768, 1.0, 0.5)
    with pytest.raises(_lib.MagbertError):
        m(torch.zeros(1, 2, 768), torch.zeros(1, 2, 47), torch.zeros(1, 2, 74))


def _innermost_mfma_loops(disassembly):
    """(function, body lines) of every innermost backward-branch loop of the gfx950 disassembly that holds an MFMA."""
    fn, fn_addr, insts, loops = None, 0, [], {}
    for line in disassembly.split("\n"):
        m = re.match(r"^([0-9a-f]{16}) <(\w+)>:", line)
        if m:
            fn, fn_addr = m.group(2), int(m.group(1), 16)
            loops[fn] = ([], [])
            continue
        m = re.search(r"//\s*([0-9A-F]{12}):", line)
        if not m or fn is None:
            continue
        addr = int(m.group(1), 16)
        body, back = loops[fn]
        body.append((addr, line))
        t = re.search(r"s_c?branch\w*\s.*<\w+\+0x([0-9a-f]+)>", line)
        if t and fn_addr + int(t.group(1), 16) <= addr:
            back.append((fn_addr + int(t.group(1), 16), addr))
    out = []
    for fn, (body, back) in loops.items():
        inner = [a for a in back if not any(b != a and b[0] >= a[0] and b[1] <= a[1] for b in back)]
        for lo, hi in inner:
            text = [l for a, l in body if lo <= a <= hi]
            if any("v_mfma" in l for l in text):
                out.append((fn, text))
    return out


def test_gemm_k_loops_hold_no_scalar_memory_reads(tmp_path):
    """gemm.hip counts LDS reads with partial `s_waitcnt lgkmcnt(N)` inside its k loops (lds_wait<N>): only valid while no scalar
    load (same counter, returns out of order) is in flight there.  Checked on the ISA of the object build() produced."""
    import shutil
    from bert_multimodal_transformer_amd import build as mb_build
    mb_build.build(verbose=False)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    obj = shutil.copy(os.path.join(mb_build.LIBDIR, "obj", "gemm.o"), tmp_path / "gemm.o")
    subprocess.run([objdump, "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    dev = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(dev) == 1, os.listdir(tmp_path)
    dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(tmp_path / dev[0])], check=True, capture_output=True,
                         text=True).stdout
    loops = _innermost_mfma_loops(dis)
    assert len(loops) > 100, len(loops)                  # every instantiation of gemm2_kernel / gemm2_grouped_tn_kernel has some
    bad = [(fn, [l.strip() for l in text if re.search(r"\bs_(buffer_)?load_|s_memtime|s_memrealtime", l)][:2])
           for fn, text in loops if any(re.search(r"\bs_(buffer_)?load_|s_memtime|s_memrealtime", l) for l in text)]
    assert not bad, bad[:4]


def test_prefetch_loads_do_not_delay_their_host_kernel(tmp_path):
    """common.h Prefetch: the LayerNorm forward touches the next GEMMs' weights with four 16-byte loads per thread.  Loads return in
    order, so the property that makes this free is a placement: the four loads sit BEHIND every load of the kernel's own and nothing
    waits for memory between them and the output stores (the wave waits at its end).  Checked on the ISA of the object build()
    produced: the compiler is free to reorder independent loads, the sched_barriers / the empty asm in common.h are what stops it."""
    import shutil
    from bert_multimodal_transformer_amd import build as mb_build
    mb_build.build(verbose=False)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    obj = shutil.copy(os.path.join(mb_build.LIBDIR, "obj", "rowops.o"), tmp_path / "rowops.o")
    subprocess.run([objdump, "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    dev = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
    m = re.search(r"<_ZN2mb13ln_fwd_kernelIDF16bLi3E\w*>:\n(.*?)s_endpgm", dis, re.S)
    assert m, "bf16 ln_fwd_kernel<3> not found"
    ops = [l.split("//")[0].strip() for l in m.group(1).split("\n") if re.search(r"global_(load|store)|s_waitcnt vmcnt", l)]
    # the prefetch loads are the only 16-byte loads through a 64-bit VGPR address (gamma / beta come through an SGPR base)
    pf = [i for i, o in enumerate(ops) if re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off", o)]
    assert len(pf) == 4, ops
    own = [i for i, o in enumerate(ops) if o.startswith("global_load") and i not in pf]
    stores = [i for i, o in enumerate(ops) if o.startswith("global_store")]
    assert max(own) < pf[0], "a load of the kernel's own behind the prefetch would wait for it: %s" % ops
    # vmcnt(N) returns once at most N of the youngest memory operations are outstanding: up to the row's last output store no wait
    # may reach back to the first prefetch load (the trailing stores are the row statistics and the never-taken sink)
    last_out = max(i for i in stores if "dwordx2" in ops[i] or "dwordx4" in ops[i])
    for i in range(pf[0], last_out):
        m2 = re.match(r"s_waitcnt vmcnt\((\d+)\)", ops[i])
        if m2:
            since = sum(1 for o in ops[pf[0]:i] if o.startswith("global_"))
            assert int(m2.group(1)) >= since, "this wait includes the prefetch loads: %s" % ops[pf[0]:i + 1]


def test_ragged_tail_is_split_evenly_and_weighted_by_sample_count():
    """data parallel, ragged last step (distributed.shard_indices / shard_step_sizes, ShardedBatchSampler.loss_scale): every rank
    gets a batch whenever the tail has >= world samples (5 samples on 4 ranks used to leave the last rank empty), sizes differ by
    at most one, and each rank's loss weight B_rank * world / B_global makes the 1/world average the mean over the global batch"""
    from bert_multimodal_transformer_amd.distributed import shard_indices, shard_step_sizes, tail_sizes
    from bert_multimodal_transformer_amd.multimodal_driver import ShardedBatchSampler
    assert tail_sizes(5, 4) == [2, 1, 1, 1] and tail_sizes(8, 4) == [2, 2, 2, 2] and sum(tail_sizes(1281 % 384, 8)) == 1281 % 384
    for n, world, bs in ((101, 4, 24), (1281, 8, 48), (53, 4, 12), (50, 4, 12)):
        per_rank = [shard_indices(n, r, world, bs, seed=1, epoch=0) for r in range(world)]
        sizes = shard_step_sizes(n, world, bs)
        assert len({len(p) for p in per_rank}) == 1 == len({len(p) for p in per_rank + [sizes]})
        for step, row in enumerate(sizes):
            assert [len(per_rank[r][step]) for r in range(world)] == row and min(row) >= 1 and max(row) - min(row) <= 1
        flat = sorted(i for p in per_rank for b in p for i in b)
        dropped = n % (world * bs) if 0 < n % (world * bs) < world else 0
        assert len(flat) == n - dropped and len(set(flat)) == len(flat)
        samplers = [ShardedBatchSampler(n, r, world, bs, 1) for r in range(world)]
        last = len(sizes) - 1
        w = [s.loss_scale(last) for s in samplers]
        assert abs(sum(w) / world - 1.0) < 1e-12 and all(abs(s.loss_scale(0) - 1.0) < 1e-12 for s in samplers)
        assert all(abs(w[r] - sizes[last][r] * world / float(sum(sizes[last]))) < 1e-12 for r in range(world))


def test_optimizer_shards_split_covers_every_element_once():
    """distributed.OptimizerShards.split: flat gradient ranges -> pieces reduced to ONE owner (inside the sharded range
    [sh_begin, sh_end), cut at the shard boundaries) + pieces every rank needs (outside it): together they cover every element of
    the input exactly once, owned pieces lie inside their owner's shard, shard boundaries sit on 256-element marks"""
    from bert_multimodal_transformer_amd.distributed import OptimizerShards

    class FakeCore(object):
        sh_begin, sh_end = 0, 85_524_480
    for world in (2, 3, 8):
        shards = [OptimizerShards(FakeCore(), r, world) for r in range(world)]
        b = shards[0].bounds
        assert b[0][0] == 0 and b[-1][1] == FakeCore.sh_end and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert all(x % 256 == 0 for x, _ in b) and all(s.bounds == b for s in shards)
        ranges = [(0, 7_077_888), (7_077_888 * 3, 7_077_888), (84_934_656, 589_824 + 1000), (90_000_000, 12345)]
        owned, shared = shards[1].split(ranges)
        cover = sorted([(o, n) for o, n, _ in owned] + shared)
        want = sum(n for _, n in ranges)
        assert sum(n for _, n in cover) == want
        for (o1, n1), (o2, _) in zip(cover, cover[1:]):
            assert o1 + n1 <= o2
        for o, n, r in owned:
            assert b[r][0] <= o and o + n <= b[r][1]
        for o, n in shared:
            assert o >= FakeCore.sh_end or o + n <= FakeCore.sh_begin


def test_bench_prices_a_kernel_trace():
    """bench.py's `roofline` comes from a kernel trace it takes itself (price_trace): steps are delimited by the run of AdamW launches that ends each, every
    GEMM symbol is priced with the FLOPs the library logged for it (weight gradients: T, not the padded Tp), AdamW with 28 B/parameter."""
    import bench
    B, L, T, Tp = 48, 50, 2400, 2432
    gemm_a, gemm_w, adam, ln = "_ZN2mb12gemm2_kernelIA", "_ZN2mb23gemm2_grouped_tn_kernelIW", "void mb::adamw_var_kernel<true, 2, true>(float*)", "ln_fwd"
    rows, t = [], 0
    for step in range(4):                          # one untraced-looking warm-up step + 3 timed ones
        for name, dur, n in ((gemm_a, 10_000, 3), (ln, 2_000, 2), (gemm_w, 40_000, 1), (adam, 250_000, 2)):
            for _ in range(n):
                rows.append((t, t + dur, name)); t += dur + 500
    log = "\n".join(["[magbert gemm] %s problems=1 flop=%d M=%d N=768 K=768" % (gemm_a, 2 * T * 768 * 768, T)] * 3 +
                    ["[magbert gemm] %s problems=4 flop=%d M=768 N=3072 K=%d" % (gemm_w, 2 * Tp * 768 * 3072 * 4, Tp), "noise line"])
    doc, roof = bench.price_trace(rows[::-1], log, B, L, "bf16", 110_853_121, steps=3)
    assert doc["kernels_per_step"] == 8.0 and abs(doc["busy_ms_per_step"] - (3 * 10 + 2 * 2 + 40 + 2 * 250) * 1e-3) < 1e-6
    by = {r["kernel"][:20]: r for r in roof}
    a, w, ad = by[gemm_a[:20]], by[gemm_w[:20]], by[adam[:20]]
    assert a["launches_per_step"] == 3 and abs(a["achieved"] - 2 * T * 768 * 768 / 10.0 * 1e-6) < 0.1 and a["bound"] == "mfma"
    assert abs(w["flop_per_launch"] - 2 * T * 768 * 3072 * 4) < 1.0            # the padded K = Tp scaled back to T
    assert ad["bound"] == "hbm" and abs(ad["achieved"] - 14 * 110_853_121 / 250.0 * 1e-3) < 0.5
    assert doc["gemm_aggregate"]["gflop_per_step"] > 0 and not doc["replayed"]
    assert bench.price_trace(rows[:5], log, B, L, "bf16", 1, steps=3)[0] is None          # too short: a reason, not a crash
    # two instantiations of one template with literal arguments only (csrc/gemm_pp.hip gemm_pn_kernel): rocprofv3 prints them demangled,
    # the library logs the mangled symbols -- told apart by their template arguments
    pn_f, pn_d = "void mb::gemm_pn_kernel<false, false, 2>(mb::GemmArgs)", "void mb::gemm_pn_kernel<false, true, 3>(mb::GemmArgs)"
    rows3, t = [], 0
    for step in range(4):
        for name, dur, n in ((pn_f, 20_000, 2), (pn_d, 10_000, 1), (adam, 100_000, 1)):
            for _ in range(n):
                rows3.append((t, t + dur, name)); t += dur + 500
    log3 = "\n".join(["[magbert gemm] _ZN2mb14gemm_pn_kernelILb0ELb0ELi2EEEvNS_8GemmArgsE problems=1 flop=%d M=%d N=768 K=3072" % (2 * T * 768 * 3072, T)] * 2 +
                     ["[magbert gemm] _ZN2mb14gemm_pn_kernelILb0ELb1ELi3EEEvNS_8GemmArgsE problems=1 flop=%d M=%d N=768 K=768" % (2 * T * 768 * 768, T)])
    doc3, roof3 = bench.price_trace(rows3, log3, B, L, "bf16", 110_853_121, steps=3)
    by3 = {r["kernel"]: r for r in roof3}
    assert abs(by3[pn_f.split("(")[0]]["flop_per_launch"] - 2 * T * 768 * 3072) < 1 and abs(by3[pn_d.split("(")[0]]["flop_per_launch"] - 2 * T * 768 * 768) < 1
    # riders (csrc/kernels.h AdamRide): part of the update runs inside the weight-gradient launches -- the sweep launches are priced with
    # the parameters THEY cover (the library logs them), a step may end with three sweep launches instead of two
    rows2, t = [], 0
    for step in range(4):
        for name, dur, n in ((gemm_a, 10_000, 3), (gemm_w, 40_000, 1), (adam, 100_000, 3)):
            for _ in range(n):
                rows2.append((t, t + dur, name)); t += dur + 500
    log2 = log + "\n" + "\n".join(["[magbert ride] params=2500000 blocks=40", "[magbert adamw] n=50000000", "[magbert adamw] n=20000000", "[magbert adamw] n=10000000"])
    doc2, roof2 = bench.price_trace(rows2, log2, B, L, "bf16", 110_853_121, steps=3)
    ad2 = {r["kernel"][:20]: r for r in roof2}[adam[:20]]
    assert doc2["kernels_per_step"] == 7.0 and ad2["launches_per_step"] == 3 and ad2["parameters_swept_per_step"] == 80_000_000
    assert abs(ad2["achieved"] - 28 * 80_000_000 / 3 / 100.0 * 1e-3) < 0.5 and doc2["adamw_riders"]["parameters_per_launch"] == 2_500_000
    top = bench.pick_roofline(doc2, roof2)
    assert top["kernel"] and "dominant_by" in top
    # ... and the grouped launch's own duration is the step's FIRST launch (the top layer carries no rider), reported next to the average
    rows3, t = [], 0
    for step in range(4):
        for name, dur, n in ((gemm_w, 40_000, 1), (gemm_a, 10_000, 1), (gemm_w, 50_000, 2), (adam, 100_000, 3)):
            for _ in range(n):
                rows3.append((t, t + dur, name)); t += dur + 500
    log3 = "\n".join(["[magbert gemm] %s problems=4 flop=%d M=768 N=3072 K=%d" % (gemm_w, 2 * Tp * 768 * 3072 * 4, Tp)] * 3 +
                     ["[magbert gemm] %s problems=1 flop=%d M=%d N=768 K=768" % (gemm_a, 2 * T * 768 * 768, T), "[magbert ride] params=2500000 blocks=40",
                      "[magbert adamw] n=50000000", "[magbert adamw] n=20000000", "[magbert adamw] n=10000000"])
    _, roof3 = bench.price_trace(rows3, log3, B, L, "bf16", 110_853_121, steps=3)
    w3 = {r["kernel"][:20]: r for r in roof3}[gemm_w[:20]]
    assert w3["launches_per_step"] == 3 and abs(w3["avg_us"] - 140.0 / 3) < 0.01 and w3["rider_free_avg_us"] == 40.0
    assert abs(w3["rider_free_frac"] / w3["frac"] - (140.0 / 3) / 40.0) < 1e-2
    assert bench.kernel_base("_ZN2mb25gemm_pp_grouped_tn_kernelENS_15GroupedGemmArgsE") == bench.kernel_base("mb::gemm_pp_grouped_tn_kernel(mb::GroupedGemmArgs)")
    assert bench.kernel_base("void mb::adamw_var_kernel<true, 2, true>(float*)") == "mb::adamw_var_kernel"


def test_bench_refuses_to_run_fewer_ranks_than_asked(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts its own N ranks -- or exits non-zero with a message when the box has fewer
    devices (RCCL needs one per rank), never a silent 1-GPU number.  Host logic only: the device count is patched."""
    import types
    import torch
    import bench
    a = types.SimpleNamespace(gpus=8)
    monkeypatch.delenv("MB_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(SystemExit) as ex:
        bench.spawn_ranks(a)
    assert "needs a ROCm GPU" in str(ex.value)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit) as ex:
        bench.spawn_ranks(a)
    assert "only 4 device(s) visible" in str(ex.value) and ex.value.code not in (0, None)


def test_rider_branches_do_not_cost_a_gemm_kernel_its_residency(tmp_path):
    """kernels.h AdamRide: the rider branch (an AdamW update with several quads in flight) sits inside GEMM kernels, and registers are
    allocated for the larger of a kernel's branches.  Round 6 found it had silently taken the 64 x 64 dgrad kernel from three blocks per
    CU to two (200 registers: a second round of tiles, +3.7 us per launch) and MAG's 64 x 64 grouped kernels from 48-72 registers to
    ~200.  Checked on the code-object metadata of the objects build() produced: blocks per CU by registers >= blocks per CU by LDS."""
    import shutil
    from bert_multimodal_transformer_amd import build as mb_build
    mb_build.build(verbose=False)
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not in this image")
    obj = shutil.copy(os.path.join(mb_build.LIBDIR, "obj", "gemm.o"), tmp_path / "gemm.o")
    subprocess.run([objdump, "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    dev = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    notes = subprocess.run([readelf, "--notes", str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(name|vgpr_count|vgpr_spill_count|group_segment_fixed_size|wavefront_size):\s+(\S+)", line)
        if m:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "wavefront_size":
                kernels[cur["name"]] = (int(cur["vgpr_count"]), int(cur["group_segment_fixed_size"]), int(cur.get("vgpr_spill_count", 0)))
                cur = {}
    by_reg = lambda v: 512 // ((v + 7) // 8 * 8)           # waves per SIMD = 256-thread blocks per CU
    by_lds = lambda l: (160 * 1024) // l
    checked = 0
    for name, (vgpr, lds, spill) in kernels.items():
        if "gemm2_ride_kernelIDF16b" in name or "gemm2_grouped_tn_kernelIDF16bLi64ELi64E" in name:
            assert spill == 0 and by_reg(vgpr) >= min(by_lds(lds), 8), (name, vgpr, lds, spill)
            checked += 1
    assert checked >= 3, sorted(kernels)[:5]
    # the layers' grouped kernels carry riders and stay at two blocks per CU (64 KB of LDS each)
    g128 = [v for n, v in kernels.items() if "gemm2_grouped_tn_kernelIDF16bLi128ELi128ELi2ELi128E" in n]
    assert g128 and by_reg(g128[0][0]) >= 2 and g128[0][2] == 0, g128
