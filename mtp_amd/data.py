"""Input side of the path: host uint8 batches -> device, double-buffered.

The reference moves each collated batch to the GPU inside `MTP_DataPreprocessor.forward` (`cast_data`,
Multi-Task_Pretrain/preprocessing.py:145-187) -- a synchronous pageable copy in front of every step -- and then normalises it
there.  Here the raw (B, H, W, 3) uint8 batch is staged in pinned memory and copied on a side HIP stream while the previous
step is still computing; the normalise / flip / pad / im2col happens on the device inside `mtp_preprocess_patchify`
(`ViT_Win_RVSA_V3_WSZ7.set_data_preprocessor`).  uint8 HWC is 4x fewer PCIe bytes than the f32 NCHW tensor the reference
ships to the device (9.6 MB instead of 38.5 MB for 64 x 224^2).

PyTorch is plumbing here (pinned allocations, streams, events); there is no kernel in this file.
"""
import torch


class HostBatchPrefetcher:
    """Iterate device-resident uint8 batches from an iterable of host batches (torch CPU tensors or numpy arrays, all of one
    shape), `depth` batches in flight.  Each `next()` returns a tensor that is valid for work enqueued on the CURRENT stream
    until the following `next()`; its slot is recycled only after that work (an event on the consumer's stream orders it).

    device="cpu" keeps the same slot / ordering logic with plain copies (used by the CPU tests)."""

    def __init__(self, batches, device="cuda", depth=2):
        assert depth >= 2
        self.it = iter(batches)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.depth = depth
        self.slots = None           # [(pinned host buffer, device buffer)] * depth, allocated on the first batch
        self.ready = [None] * depth     # event: H2D copy of the slot finished (side stream)
        self.consumed = [None] * depth  # event: consumer's work on the slot enqueued before this point (consumer stream)
        self.filled = [False] * depth
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.head = 0               # next slot to hand out
        self.tail = 0               # next slot to fill
        self.last = None            # slot handed out by the previous next()
        self.exhausted = False
        self.bytes_copied = 0

    def _alloc(self, like):
        self.slots = []
        for _ in range(self.depth):
            host = torch.empty(like.shape, dtype=like.dtype, pin_memory=self.cuda)
            dev = torch.empty(like.shape, dtype=like.dtype, device=self.device)
            self.slots.append((host, dev))

    def _fill_one(self):
        """stage the next host batch into slot `tail` and start its copy; False when the source is exhausted"""
        if self.exhausted or self.filled[self.tail]:
            return False
        try:
            b = next(self.it)
        except StopIteration:
            self.exhausted = True
            return False
        b = torch.as_tensor(b)
        if b.device.type != "cpu":
            raise ValueError("HostBatchPrefetcher takes host batches")
        if self.slots is None:
            self._alloc(b)
        host, dev = self.slots[self.tail]
        if b.shape != host.shape or b.dtype != host.dtype:
            raise ValueError("all batches must share one shape / dtype: got %s %s, expected %s %s" % (tuple(b.shape), b.dtype, tuple(host.shape), host.dtype))
        if self.cuda:
            if self.ready[self.tail] is not None:
                self.ready[self.tail].synchronize()      # the pinned buffer's previous H2D copy has been read out
            host.copy_(b)                                # pageable -> pinned (host memcpy)
            with torch.cuda.stream(self.stream):
                if self.consumed[self.tail] is not None:
                    self.stream.wait_event(self.consumed[self.tail])   # the device buffer's last consumer is done
                dev.copy_(host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.ready[self.tail] = ev
        else:
            host.copy_(b)
            dev.copy_(host)
        self.bytes_copied += b.numel() * b.element_size()
        self.filled[self.tail] = True
        self.tail = (self.tail + 1) % self.depth
        return True

    def __iter__(self):
        return self

    def __next__(self):
        if self.last is not None:      # everything the caller enqueued on the previous batch is ordered before its slot is reused
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.consumed[self.last] = ev
            self.filled[self.last] = False
            self.last = None
        while self._fill_one():
            pass
        if not self.filled[self.head]:
            raise StopIteration
        slot = self.head
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.ready[slot])
        self.head = (self.head + 1) % self.depth
        self.last = slot
        return self.slots[slot][1]
