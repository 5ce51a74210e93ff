"""Asynchronous host-to-device staging of DataLoader batches.

The reference moves every batch inside the training loop, six synchronous-looking copies on the compute stream
(`batch = tuple(t.to(DEVICE) for t in batch)`, /root/reference/multimodal_driver.py:359, :396, :429): one host round trip
per tensor per step.  Here a batch is packed into ONE pinned host block, crosses PCIe as ONE copy on a dedicated HIP
stream while the previous step is still computing, and is handed to the step as six device views of a landing slot;
the compute stream only waits for the copy's event.  New relative to the reference (which has no overlap at all); the
tensors the model sees are bit-identical to `t.to(DEVICE)`.
"""
import torch


class _Slot(object):
    def __init__(self, nbytes, device):
        self.nbytes = nbytes
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.landed = torch.cuda.Event()      # the H2D copy into .dev finished (recorded on the copy stream)
        self.used = False


class DevicePrefetcher(object):
    """Iterates `loader`, yielding each batch as a tuple of device tensors (views of a landing slot).

    depth batches are in flight ahead of the consumer (default 1: batch i+1 crosses PCIe while step i computes).  A slot is
    recycled `slots` batches later; before its device buffer is overwritten the copy stream waits for everything the compute
    stream had enqueued at that moment, which includes every reader of the slot's previous contents.  `squeeze_dim1` applies
    the reference's `torch.squeeze(visual, 1)` / `torch.squeeze(acoustic, 1)` (multimodal_driver.py:361-362) to tensors 1, 2."""

    def __init__(self, loader, device=None, depth=1, squeeze_dim1=True):
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePrefetcher needs a ROCm device (no CPU fallback)")
        self.loader = loader
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.depth = max(1, int(depth))
        self.nslots = self.depth + 2
        self.squeeze_dim1 = squeeze_dim1
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = []
        self._next = 0

    def __len__(self):
        return len(self.loader)

    @staticmethod
    def _layout(batch):
        offs, off = [], 0
        for t in batch:
            offs.append(off)
            off = (off + t.numel() * t.element_size() + 255) // 256 * 256
        return offs, max(off, 256)

    def _stage(self, batch):
        batch = tuple(batch)
        offs, nbytes = self._layout(batch)
        k = self._next
        self._next = (self._next + 1) % self.nslots
        while len(self.slots) <= k:
            self.slots.append(None)
        slot = self.slots[k]
        if slot is None or slot.nbytes < nbytes:
            slot = self.slots[k] = _Slot(nbytes, self.device)
        if slot.used:
            slot.landed.synchronize()          # the previous copy out of this pinned block is long finished: returns at once
        views = []
        for t, off in zip(batch, offs):
            n = t.numel() * t.element_size()
            src = t.detach()
            if src.is_cuda:
                src = src.cpu()
            slot.host[off: off + n].view(src.dtype).view(src.shape).copy_(src)
            views.append(slot.dev[off: off + n].view(src.dtype).view(src.shape))
        compute = torch.cuda.current_stream(self.device)
        gate = torch.cuda.Event()
        gate.record(compute)                    # readers of this slot's previous contents are in front of this point
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(gate)
            slot.dev[:nbytes].copy_(slot.host[:nbytes], non_blocking=True)
            slot.landed.record(self.copy_stream)
        slot.used = True
        if self.squeeze_dim1:
            for i in (1, 2):
                if i < len(views) and views[i].dim() > 1 and views[i].shape[1] == 1:
                    views[i] = views[i].squeeze(1)
        return slot, tuple(views)

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        done = False
        while True:
            while not done and len(queue) < self.depth + 1:
                try:
                    queue.append(self._stage(next(it)))
                except StopIteration:
                    done = True
            if not queue:
                return
            slot, views = queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(slot.landed)
            yield views
