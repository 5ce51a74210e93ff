"""Train / eval / test loop -- drop-in for /root/reference/multimodal_driver.py, on the HIP path.

Same function names, argument lists and return values as the reference:
    convert_to_features, prepare_bert_input, prepare_xlnet_input, get_appropriate_dataset, set_up_data_loader,
    set_random_seed, prep_for_training, train_epoch, eval_epoch, test_epoch, test_score_model, train, main
and the same CLI flags / defaults (multimodal_driver.py:35-57).  `args` is a module global like in the reference
(parsed lazily by main(); tests assign driver.args = parse_args([...])).

Differences (all additive):
  * train_epoch runs the fused forward+MSE+backward step and accumulates the loss ON DEVICE; the per-step
    `loss.item()` host sync of multimodal_driver.py:380 is gone (one sync per epoch).  `--reference_loop true`
    runs the literal reference sequence (model(...) -> MSELoss -> loss.backward() -> optimizer.step()) instead.
  * data parallel: launched under torchrun, each rank takes its shard of every global batch and gradients are
    all-reduced over RCCL while the backward runs (distributed.py).
  * `--synthetic N` builds an N-sample dataset in prepare_bert_input's exact layout (no mosi.pkl offline);
    VISUAL_DIM follows --dataset (47 mosi / 35 mosei) instead of a hand-edited module constant.
  * `--seed 7` works (the reference's argparse `seed` type rejects every integer string, argparse_utils.py:18-31).
"""
from __future__ import absolute_import, division, print_function

import argparse
import os
import pickle
import random
import sys
import time

import numpy as np
import torch
import torch.nn as nn
from torch.nn import MSELoss
from torch.utils.data import DataLoader, TensorDataset

from .bert import BertConfig, MAG_BertForSequenceClassification
from .global_configs import DATASET_DIMS
from .optimization import AdamW, get_linear_schedule_with_warmup

args = None
DEVICE = None      # resolved in main() / _device(): the current ROCm device (reference: cuda:0, global_configs.py:7)


def _device():
    global DEVICE
    if DEVICE is None:
        DEVICE = torch.device("cuda", torch.cuda.current_device())
    return DEVICE


def str2bool(s):
    if isinstance(s, bool):
        return s
    if s.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if s.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected. Recieved {0}".format(s))


def seed(s):
    """argparse_utils.py:18-31, fixed to accept integer strings."""
    if isinstance(s, str) and s == "random":
        return random.randint(0, 9999)
    try:
        v = int(s)
    except (TypeError, ValueError):
        raise argparse.ArgumentTypeError("Integer value is expected. Recieved {0}".format(s))
    if 0 <= v <= 9999:
        return v
    raise argparse.ArgumentTypeError("Seed must be between 0 and 9999. Received {0}".format(s))


def get_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", type=str, choices=["mosi", "mosei"], default="mosi")
    parser.add_argument("--max_seq_length", type=int, default=50)
    parser.add_argument("--train_batch_size", type=int, default=48)
    parser.add_argument("--dev_batch_size", type=int, default=128)
    parser.add_argument("--test_batch_size", type=int, default=128)
    parser.add_argument("--n_epochs", type=int, default=40)
    parser.add_argument("--beta_shift", type=float, default=1.0)
    parser.add_argument("--dropout_prob", type=float, default=0.5)
    parser.add_argument("--model", type=str, choices=["bert-base-uncased", "xlnet-base-cased"], default="bert-base-uncased")
    parser.add_argument("--learning_rate", type=float, default=1e-5)
    parser.add_argument("--gradient_accumulation_step", type=int, default=1)
    parser.add_argument("--warmup_proportion", type=float, default=0.1)
    parser.add_argument("--seed", type=seed, default="random")
    # additions
    parser.add_argument("--synthetic", type=int, default=0, help="use N synthetic training samples (no .pkl needed)")
    parser.add_argument("--compute_dtype", choices=["bf16", "fp32"], default="bf16")
    parser.add_argument("--reference_loop", type=str2bool, default=False)
    parser.add_argument("--step_graph", type=str2bool, default=True,
                        help="fused loop, single process: each optimizer step = step prologue + one replayed hipGraph "
                             "(mb_bert_train_step); false = the same engine call launching the kernels one by one")
    parser.add_argument("--prefetch", type=str2bool, default=True,
                        help="pack every batch into a pinned host block the GPU reads in place (prefetch.PinnedBatchRing) instead of "
                             "six t.to(DEVICE) copies per step")
    parser.add_argument("--pretrained", type=str, default="", help="local checkpoint dir/file (offline)")
    return parser


def parse_args(argv=None):
    return get_parser().parse_args(argv)


class InputFeatures(object):
    """A single set of features of data (multimodal_driver.py:64-73)."""

    def __init__(self, input_ids, visual, acoustic, input_mask, segment_ids, label_id):
        self.input_ids = input_ids
        self.visual = visual
        self.acoustic = acoustic
        self.input_mask = input_mask
        self.segment_ids = segment_ids
        self.label_id = label_id


class MultimodalConfig(object):
    """multimodal_driver.py:76-79."""

    def __init__(self, beta_shift, dropout_prob):
        self.beta_shift = beta_shift
        self.dropout_prob = dropout_prob


def _dims():
    d = DATASET_DIMS[args.dataset]
    return d["visual_dim"], d["acoustic_dim"]


# ------------------------------------------------------------------------------------------------- features
def _wordpieces(words, tokenizer):
    """wordpieces of a word list + for every piece the index of the word it came from"""
    pieces, owner = [], []
    for w, word in enumerate(words):
        sub = tokenizer.tokenize(word)
        pieces += sub
        owner += [w] * len(sub)
    return pieces, np.asarray(owner, dtype=np.int64)


def convert_to_features(examples, max_seq_length, tokenizer):
    """Feature conversion with the semantics of multimodal_driver.py:82-140: every wordpiece inherits the visual / acoustic
    row of the word it belongs to (a gather by owner index), the sequence is cut to max_seq_length - 2 pieces, and the
    model-specific layout (special tokens, padding side, segment ids) is applied by prepare_bert_input / prepare_xlnet_input."""
    layout = {"bert-base-uncased": prepare_bert_input, "xlnet-base-cased": prepare_xlnet_input}[args.model]
    room = max_seq_length - 2
    out = []
    for (words, visual, acoustic), label_id, _segment in examples:
        pieces, owner = _wordpieces(words, tokenizer)
        visual, acoustic = np.asarray(visual), np.asarray(acoustic)
        owner = owner[:room]
        vis = visual[owner] if owner.size else np.zeros((0, visual.shape[1]))
        aco = acoustic[owner] if owner.size else np.zeros((0, acoustic.shape[1]))
        input_ids, vis, aco, input_mask, segment_ids = layout(pieces[:room], vis, aco, tokenizer)
        if not (len(input_ids) == len(input_mask) == len(segment_ids) == vis.shape[0] == aco.shape[0] == args.max_seq_length):
            raise AssertionError("feature rows must all have max_seq_length = %d entries" % args.max_seq_length)
        out.append(InputFeatures(input_ids=input_ids, visual=vis, acoustic=aco, input_mask=input_mask, segment_ids=segment_ids,
                                 label_id=label_id))
    return out


def prepare_bert_input(tokens, visual, acoustic, tokenizer):
    """multimodal_driver.py:143-173: [CLS] x [SEP] then right-pad; zero modality rows on special/pad slots."""
    L = args.max_seq_length
    A, V = acoustic.shape[1], visual.shape[1]
    tokens = [tokenizer.cls_token] + tokens + [tokenizer.sep_token]
    n = len(tokens)
    aco = np.zeros((L, A)); aco[1:n - 1] = acoustic
    vis = np.zeros((L, V)); vis[1:n - 1] = visual
    input_ids = tokenizer.convert_tokens_to_ids(tokens) + [0] * (L - n)
    input_mask = [1] * n + [0] * (L - n)
    segment_ids = [0] * L
    return input_ids, vis, aco, input_mask, segment_ids


def prepare_xlnet_input(tokens, visual, acoustic, tokenizer):
    """multimodal_driver.py:176-205: x [SEP] [CLS], LEFT-padded; segment ids 0.. / 2 for CLS / 3 for pad."""
    L = args.max_seq_length
    A, V = acoustic.shape[1], visual.shape[1]
    tokens = tokens + [tokenizer.sep_token] + [tokenizer.cls_token]
    n = len(tokens)
    pad = L - n
    aco = np.zeros((L, A)); aco[pad:pad + n - 2] = acoustic
    vis = np.zeros((L, V)); vis[pad:pad + n - 2] = visual
    input_ids = [tokenizer.pad_token_id] * pad + tokenizer.convert_tokens_to_ids(tokens)
    input_mask = [0] * pad + [1] * n
    segment_ids = [3] * pad + [0] * (n - 1) + [2]
    return input_ids, vis, aco, input_mask, segment_ids


def get_tokenizer(model):
    """multimodal_driver.py:208-218.  Needs the vocab files in the local HF cache (no network)."""
    try:
        from transformers import BertTokenizer, XLNetTokenizer
    except Exception as e:      # pragma: no cover
        raise RuntimeError("transformers tokenizers unavailable: %s" % e)
    if model == "bert-base-uncased":
        return BertTokenizer.from_pretrained(model)
    elif model == "xlnet-base-cased":
        return XLNetTokenizer.from_pretrained(model)
    raise ValueError("Expected 'bert-base-uncased' or 'xlnet-base-cased, but received {}".format(model))


def features_to_dataset(features):
    """multimodal_driver.py:226-246: six tensors, this order."""
    all_input_ids = torch.tensor(np.array([f.input_ids for f in features]), dtype=torch.long)
    all_input_mask = torch.tensor(np.array([f.input_mask for f in features]), dtype=torch.long)
    all_segment_ids = torch.tensor(np.array([f.segment_ids for f in features]), dtype=torch.long)
    all_visual = torch.tensor(np.array([f.visual for f in features]), dtype=torch.float)
    all_acoustic = torch.tensor(np.array([f.acoustic for f in features]), dtype=torch.float)
    all_label_ids = torch.tensor(np.array([f.label_id for f in features]), dtype=torch.float)
    return TensorDataset(all_input_ids, all_visual, all_acoustic, all_input_mask, all_segment_ids, all_label_ids)


def get_appropriate_dataset(data, tokenizer=None):
    tokenizer = tokenizer or get_tokenizer(args.model)
    return features_to_dataset(convert_to_features(data, args.max_seq_length, tokenizer))


def synthetic_dataset(n, L, V, A, seed_=1234, vocab=30522, layout="bert"):
    """n samples in prepare_bert_input's layout: ids [101, tokens, 102, 0...], mask, seg 0,
    modality rows N(0,1) on word rows and EXACT zeros on [CLS]/[SEP]/pad rows, labels U(-3, 3).
    layout="xlnet": prepare_xlnet_input's layout (multimodal_driver.py:176-205) -- LEFT padded, ids [5.., tokens, 4, 3],
    mask 0 on pads, segment ids 3 (pad) / 0 (tokens, <sep>) / 2 (<cls>)."""
    rs = np.random.RandomState(seed_)
    lens = rs.randint(5, L - 2 + 1, size=n)
    if layout == "xlnet":
        vocab = min(vocab, 32000)
        pos = np.arange(L)[None, :]
        pad = (L - 2 - lens)[:, None]
        word = (pos >= pad) & (pos < L - 2)
        ids = np.full((n, L), 5, np.int64)
        tok = rs.randint(10, vocab, size=(n, L))
        ids[word] = tok[word]
        ids[:, L - 2], ids[:, L - 1] = 4, 3
        mask = (pos >= pad).astype(np.int64)
        seg = np.where(pos >= pad, 0, 3).astype(np.int64)
        seg[:, L - 1] = 2
        vis = rs.randn(n, L, V).astype(np.float32)
        aco = rs.randn(n, L, A).astype(np.float32)
        vis[~word] = 0.0
        aco[~word] = 0.0
        label = rs.uniform(-3, 3, size=n).astype(np.float32)
        t = torch.from_numpy
        return TensorDataset(t(ids), t(vis), t(aco), t(mask), t(seg), t(label))
    ids = np.zeros((n, L), np.int64)
    mask = np.zeros((n, L), np.int64)
    seg = np.zeros((n, L), np.int64)
    vis = rs.randn(n, L, V).astype(np.float32)
    aco = rs.randn(n, L, A).astype(np.float32)
    tok = rs.randint(1000, vocab, size=(n, L))
    pos = np.arange(L)[None, :]
    word = (pos >= 1) & (pos <= lens[:, None])
    ids[word] = tok[word]
    ids[:, 0] = 101
    ids[np.arange(n), lens + 1] = 102
    mask[pos.repeat(n, 0) <= (lens[:, None] + 1)] = 1
    vis[~word] = 0.0
    aco[~word] = 0.0
    label = rs.uniform(-3, 3, size=n).astype(np.float32)
    t = torch.from_numpy
    return TensorDataset(t(ids), t(vis), t(aco), t(mask), t(seg), t(label))


def _dist():
    """(rank, world) of a torchrun launch; initialises the process group (RCCL) on first use."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch.distributed as dist
    if not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MB_DIST_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world, **kw)
    return dist.get_rank(), world


class ShardedBatchSampler(object):
    """Per-rank minibatches of the data-parallel run (distributed.shard_indices): every global batch of
    world * train_batch_size shuffled samples is cut into one contiguous slice per rank; reshuffled every epoch."""

    def __init__(self, n, rank, world, batch_size, seed_):
        self.n, self.rank, self.world, self.batch_size, self.seed, self.epoch = n, rank, world, batch_size, seed_, 0

    def _batches(self):
        from .distributed import shard_indices
        return shard_indices(self.n, self.rank, self.world, self.batch_size, self.seed, self.epoch)

    def __iter__(self):
        b = self._batches()
        self.epoch += 1
        return iter(b)

    def loss_scale(self, step):
        """weight of this rank's mean loss in step `step`: B_rank * world / B_global (1.0 except in a ragged last step), so that the
        1/world average of the ranks' gradients is the mean over the global batch"""
        from .distributed import shard_step_sizes
        sizes = shard_step_sizes(self.n, self.world, self.batch_size)
        if step >= len(sizes):
            return 1.0
        return sizes[step][self.rank] * self.world / float(sum(sizes[step]))

    def __len__(self):
        return len(self._batches())


def set_up_data_loader():
    """multimodal_driver.py:249-286 (+ synthetic mode, + per-rank sharding under torchrun)."""
    V, A = _dims()
    rank, world = _dist()
    if args.synthetic:
        n = args.synthetic
        lay = "xlnet" if args.model == "xlnet-base-cased" else "bert"
        train_dataset = synthetic_dataset(n, args.max_seq_length, V, A, 1234, layout=lay)
        dev_dataset = synthetic_dataset(max(8, n // 6), args.max_seq_length, V, A, 1235, layout=lay)
        test_dataset = synthetic_dataset(max(8, n // 2), args.max_seq_length, V, A, 1236, layout=lay)
    else:
        with open(f"datasets/{args.dataset}.pkl", "rb") as handle:
            data = pickle.load(handle)
        tok = get_tokenizer(args.model)
        train_dataset = get_appropriate_dataset(data["train"], tok)
        dev_dataset = get_appropriate_dataset(data["dev"], tok)
        test_dataset = get_appropriate_dataset(data["test"], tok)
    num_train_optimization_steps = (
        int(len(train_dataset) / (args.train_batch_size * world) / args.gradient_accumulation_step) * args.n_epochs)
    if world > 1:      # weak scaling: train_batch_size stays the PER-GPU batch, the global batch is world times larger
        sampler = ShardedBatchSampler(len(train_dataset), rank, world, args.train_batch_size, args.seed if isinstance(args.seed, int) else 0)
        train_dataloader = DataLoader(train_dataset, batch_sampler=sampler)
    else:
        train_dataloader = DataLoader(train_dataset, batch_size=args.train_batch_size, shuffle=True)
    # under DP the dev / test batches are dealt round-robin to the ranks (eval_epoch / test_epoch), so every rank must see the
    # same batch order: no shuffle there (the metrics do not depend on the order)
    dev_dataloader = DataLoader(dev_dataset, batch_size=args.dev_batch_size, shuffle=world == 1)
    test_dataloader = DataLoader(test_dataset, batch_size=args.test_batch_size, shuffle=world == 1)
    return train_dataloader, dev_dataloader, test_dataloader, num_train_optimization_steps


def set_random_seed(seed: int):
    """multimodal_driver.py:289-308."""
    print("Seed: {}".format(seed))
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def optimizer_grouped_parameters(model, weight_decay=0.01):
    """multimodal_driver.py:328-343."""
    param_optimizer = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    return [
        {"params": [p for n, p in param_optimizer if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in param_optimizer if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]


def prep_for_training(num_train_optimization_steps: int):
    """multimodal_driver.py:311-351."""
    multimodal_config = MultimodalConfig(beta_shift=args.beta_shift, dropout_prob=args.dropout_prob)
    V, A = _dims()
    dt = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32
    if args.model == "bert-base-uncased":
        if args.pretrained:
            model = MAG_BertForSequenceClassification.from_pretrained(
                args.pretrained, multimodal_config=multimodal_config, num_labels=1, visual_dim=V, acoustic_dim=A,
                compute_dtype=dt)
        else:       # offline: fresh init by the reference's init law
            model = MAG_BertForSequenceClassification(BertConfig(num_labels=1), multimodal_config, visual_dim=V,
                                                      acoustic_dim=A, compute_dtype=dt)
    elif args.model == "xlnet-base-cased":
        from .xlnet import MAG_XLNetForSequenceClassification, XLNetConfig
        if args.pretrained:
            model = MAG_XLNetForSequenceClassification.from_pretrained(
                args.pretrained, multimodal_config=multimodal_config, num_labels=1, visual_dim=V, acoustic_dim=A,
                compute_dtype=dt)
        else:
            model = MAG_XLNetForSequenceClassification(XLNetConfig(num_labels=1), multimodal_config, visual_dim=V,
                                                       acoustic_dim=A, compute_dtype=dt)
    model.to(_device())
    optimizer = AdamW(optimizer_grouped_parameters(model), lr=args.learning_rate)
    scheduler = get_linear_schedule_with_warmup(
        optimizer, num_warmup_steps=args.warmup_proportion * num_train_optimization_steps,
        num_training_steps=num_train_optimization_steps)
    rank, world = _dist()
    if world > 1:
        from .distributed import DataParallel
        # the gradient exchange: issued from C inside the single-call step (distributed.Comm), by stage hooks on accumulation steps
        model._dp = DataParallel(model, optimizer, row_capacity=args.train_batch_size * args.max_seq_length)
        model._dp.broadcast_parameters(0)
    return model, optimizer, scheduler


def _unpack(batch):
    batch = tuple(t.to(_device(), non_blocking=True) for t in batch)
    input_ids, visual, acoustic, input_mask, segment_ids, label_ids = batch
    visual = torch.squeeze(visual, 1)
    acoustic = torch.squeeze(acoustic, 1)
    return input_ids, visual, acoustic, input_mask, segment_ids, label_ids


def _batches(dataloader):
    """The loop's `batch = tuple(t.to(DEVICE) for t in batch)` (multimodal_driver.py:359-362): by default every batch is
    packed into a pinned host block that the engine's gather launch reads in place (prefetch.PinnedBatchRing); with
    --prefetch false the six copies of the reference."""
    if getattr(args, "prefetch", True):
        from .prefetch import PinnedBatchRing
        ring = _RINGS.get(str(_device()))
        if ring is None:                       # one ring (four pinned blocks) per device for the whole run
            ring = _RINGS[str(_device())] = PinnedBatchRing(None, _device())
        ring.loader = dataloader
        return ring
    return (_unpack(b) for b in dataloader)


_RINGS = {}


def train_epoch(model: nn.Module, train_dataloader: DataLoader, optimizer, scheduler):
    """multimodal_driver.py:354-388.  Returns the mean training loss of the epoch."""
    model.train()
    accum = args.gradient_accumulation_step
    nb_tr_steps = 0
    dp = getattr(model, "_dp", None)
    if args.reference_loop:
        tr_loss = 0
        for step, batch in enumerate(train_dataloader):
            input_ids, visual, acoustic, input_mask, segment_ids, label_ids = _unpack(batch)
            outputs = model(input_ids, visual, acoustic, token_type_ids=segment_ids, attention_mask=input_mask, labels=None)
            logits = outputs[0]
            loss = MSELoss()(logits.view(-1), label_ids.view(-1))
            if accum > 1:
                loss = loss / accum
            if dp is not None:
                dp.sync = (step + 1) % accum == 0
            loss.backward()
            tr_loss += loss.item()
            nb_tr_steps += 1
            if (step + 1) % accum == 0:
                optimizer.step()
                scheduler.step()
                optimizer.zero_grad()
        return tr_loss / max(1, nb_tr_steps)
    with model.stream_scope():                        # one private HIP stream for the whole epoch (no NULL-stream hops)
        model.loss_running(reset=True)
        batches = _batches(train_dataloader)
        use_graph = None if getattr(args, "step_graph", True) else "launches"
        sampler = getattr(train_dataloader, "batch_sampler", None)
        for step, batch in enumerate(batches):
            input_ids, visual, acoustic, input_mask, segment_ids, label_ids = batch
            update = (step + 1) % accum == 0
            share = 1.0
            if dp is not None:
                dp.sync = update                          # all-reduce only on the micro-step that is followed by step()
                if hasattr(sampler, "loss_scale"):
                    share = sampler.loss_scale(step)      # ragged last step: ranks weigh in by their sample counts
            # forward + MSE + backward (+ optimizer.step() + zero_grad()): one replayed hipGraph where the engine can
            model.train_step(input_ids, visual, acoustic, input_mask, segment_ids, label_ids,
                             optimizer=optimizer if update else None, loss_scale=share / accum, graph=use_graph)
            nb_tr_steps += 1
            if update:
                scheduler.step()
        running = model.loss_running(reset=True)
    stats = torch.stack([running.double() / accum, torch.tensor(float(nb_tr_steps), dtype=torch.float64, device=running.device)])
    if dp is not None and _dist()[1] > 1:
        import torch.distributed as dist
        dist.all_reduce(stats)                        # the epoch's loss covers every rank's shard, like valid_loss
    stats = stats.cpu()                               # the only host sync of the epoch
    return float(stats[0]) / max(1.0, float(stats[1]))


def eval_epoch(model: nn.Module, dev_dataloader: DataLoader, optimizer):
    """multimodal_driver.py:391-421: mean over the dev batches of MSELoss(logits, labels) (divided by the accumulation
    factor like the reference).  The per-batch loss is the engine's fused MSE, summed on the device: one host sync."""
    model.eval()
    nb_dev_steps = 0
    rank, world = _dist()
    with torch.no_grad(), model.stream_scope():
        model.loss_running(reset=True)
        for step, batch in enumerate(_batches(dev_dataloader)):
            if step % world != rank:          # data parallel: each rank evaluates its share of the batches
                continue
            input_ids, visual, acoustic, input_mask, segment_ids, label_ids = batch
            model.eval_step(input_ids, visual, acoustic, input_mask, segment_ids, label_ids)
            nb_dev_steps += 1
        dev_loss = model.loss_running(reset=True)
    if args.gradient_accumulation_step > 1:
        dev_loss = dev_loss / args.gradient_accumulation_step
    if world > 1:
        import torch.distributed as dist
        acc = torch.stack([dev_loss.float(), torch.tensor(float(nb_dev_steps), device=dev_loss.device)])
        dist.all_reduce(acc)                  # sum of the batch losses, number of batches
        return float(acc[0].item()) / max(1.0, float(acc[1].item()))
    return float(dev_loss.item()) / max(1, nb_dev_steps)


def test_epoch(model: nn.Module, test_dataloader: DataLoader):
    """multimodal_driver.py:424-459."""
    model.eval()
    preds, labels = [], []
    rank, world = _dist()
    with torch.no_grad(), model.stream_scope():
        for step, batch in enumerate(_batches(test_dataloader)):
            if step % world != rank:
                continue
            input_ids, visual, acoustic, input_mask, segment_ids, label_ids = batch
            outputs = model(input_ids, visual, acoustic, token_type_ids=segment_ids, attention_mask=input_mask, labels=None)
            preds.append(outputs[0].detach().view(-1))
            labels.append(label_ids.detach().view(-1).cpu().clone())      # (the ring recycles its pinned blocks)
        preds = torch.cat(preds).cpu().numpy() if preds else np.zeros((0,), np.float32)
    labels = torch.cat(labels).numpy() if labels else np.zeros((0,), np.float32)
    if world > 1:                             # every rank gets every prediction (order = by rank; the metrics are order-free)
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, (preds, labels))
        preds = np.concatenate([p for p, _ in parts])
        labels = np.concatenate([l for _, l in parts])
    return preds, labels


def score_predictions(preds, y_test, use_zero=False):
    """Metrics of test_score_model (multimodal_driver.py:465-480) on the samples whose label is non-zero (all of them with
    use_zero): binary accuracy and weighted F1 of the sign (>= 0), mean absolute error, Pearson correlation."""
    from sklearn.metrics import accuracy_score, f1_score
    preds, y_test = np.asarray(preds), np.asarray(y_test)
    keep = np.flatnonzero(np.ones_like(y_test, dtype=bool) if use_zero else (y_test != 0))
    p, y = preds[keep], y_test[keep]
    mae = np.abs(p - y).mean()
    corr = np.corrcoef(p, y)[0, 1]
    p_pos, y_pos = p >= 0, y >= 0
    return accuracy_score(y_pos, p_pos), mae, corr, f1_score(y_pos, p_pos, average="weighted")


def test_score_model(model: nn.Module, test_dataloader: DataLoader, use_zero=False):
    """multimodal_driver.py:462-480."""
    preds, y_test = test_epoch(model, test_dataloader)
    return score_predictions(preds, y_test, use_zero)


def train(model, train_dataloader, validation_dataloader, test_data_loader, optimizer, scheduler):
    """multimodal_driver.py:483-523 (wandb.log replaced by JSON lines on stdout)."""
    import json
    valid_losses, test_accuracies = [], []
    for epoch_i in range(int(args.n_epochs)):
        t0 = time.time()
        train_loss = train_epoch(model, train_dataloader, optimizer, scheduler)
        torch.cuda.synchronize()
        dt = time.time() - t0
        valid_loss = eval_epoch(model, validation_dataloader, optimizer)
        test_acc, test_mae, test_corr, test_f_score = test_score_model(model, test_data_loader)
        if _dist()[0] != 0:
            continue
        print("epoch:{}, train_loss:{}, valid_loss:{}, test_acc:{}".format(epoch_i, train_loss, valid_loss, test_acc))
        valid_losses.append(valid_loss)
        test_accuracies.append(test_acc)
        print(json.dumps({k: float(v) for k, v in {
            "train_loss": train_loss, "valid_loss": valid_loss, "test_acc": test_acc, "test_mae": test_mae,
            "test_corr": test_corr, "test_f_score": test_f_score, "best_valid_loss": min(valid_losses),
            "best_test_acc": max(test_accuracies), "train_samples_per_sec": len(train_dataloader.dataset) / dt}.items()}))


def main(argv=None):
    global args
    args = parse_args(argv)
    rank, world = _dist()
    if world > 1:
        # `--seed random` (the default) draws in every process: the ranks must agree, the sharded sampler cuts ONE global
        # permutation (distributed.shard_indices) and the replicas must start from the same weights
        import torch.distributed as dist
        box = [args.seed]
        dist.broadcast_object_list(box, src=0)
        args.seed = int(box[0])
    set_random_seed(args.seed)
    train_data_loader, dev_data_loader, test_data_loader, num_train_optimization_steps = set_up_data_loader()
    model, optimizer, scheduler = prep_for_training(num_train_optimization_steps)
    train(model, train_data_loader, dev_data_loader, test_data_loader, optimizer, scheduler)


if __name__ == "__main__":
    main()
