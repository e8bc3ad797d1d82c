"""Input on-ramp of the hot path: the crop-to-min batching of AudioFileDataset.__iter__
(audio8/data.py:409-426) and the `batch.to(device)` of the training loops (audio8/pretrain.py:177), as a pinned,
double-buffered, asynchronous upload.

`crop_to_min_batches` reproduces the reference's batching rule exactly -- including its quirk that the sample which
arrives once the token budget is reached is dropped (the reference appends only while the budget is open and
discards the sample that triggers the yield).  `BatchOnRamp` takes such a list of ragged waveforms, crops and stacks
them straight into pinned host memory and copies them to the device on its own HIP stream; `get()` makes the
compute stream wait for that copy only, so the upload of batch n+1 overlaps the step on batch n."""
import collections

import numpy as np
import torch


def crop_to_min_batches(samples, max_length, target_tokens_per_batch):
    """Generator over lists of samples cropped to their common minimum length (audio8/data.py:409-426).

    samples: iterable of 1-D arrays.  Yields np.float32 arrays [B, min_len]."""
    min_length = max_length
    num_tokens_predicted = 0
    pending = []
    for sample in samples:
        if num_tokens_predicted < target_tokens_per_batch:
            min_length = min(min_length, len(sample))
            pending.append(sample)
            num_tokens_predicted = len(pending) * min_length
        else:
            batch = [s[:min_length] for s in pending]
            pending = []
            num_tokens_predicted = 0                    # (min_length is NOT reset, and `sample` is dropped: reference behaviour)
            yield batch


class BatchOnRamp:
    """Pinned, double-buffered host->device upload of cropped batches.

        ramp = BatchOnRamp(depth=2)
        ramp.submit(next_list_of_waveforms)             # crop + stack into pinned memory, async H2D on the copy stream
        x = ramp.get()                                   # [B, min_len] f32 on the device; compute stream waits for its copy
    """

    def __init__(self, depth=2, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("a8hip: BatchOnRamp needs a GPU (there is no CPU path)")
        self.device = device
        self.depth = depth
        self.copy_stream = torch.cuda.Stream()
        self.queue = collections.deque()
        self._pinned = {}                                # (slot) -> pinned tensor, grown on demand
        self._slot = 0

    def _staging(self, n):
        buf = self._pinned.get(self._slot)
        if buf is None or buf.numel() < n:
            buf = torch.empty(n, dtype=torch.float32, pin_memory=True)
            self._pinned[self._slot] = buf
        return buf[:n]

    def submit(self, samples, min_length=None):
        """samples: list of 1-D float arrays (any lengths >= min_length); crops to the common minimum."""
        if len(self.queue) >= self.depth:
            raise RuntimeError("a8hip: BatchOnRamp is full (call get() before submitting more)")
        if min_length is None:
            min_length = min(len(s) for s in samples)
        B = len(samples)
        # the staging slot is reused every `depth` submits: its previous copy must have drained
        prev = getattr(self, "_slot_events", {}).get(self._slot)
        if prev is not None:
            prev.synchronize()
        stage = self._staging(B * min_length).view(B, min_length)
        dst = stage.numpy()
        for i, s in enumerate(samples):
            dst[i, :] = np.asarray(s[:min_length], dtype=np.float32)
        with torch.cuda.stream(self.copy_stream):
            x = stage.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self.__dict__.setdefault("_slot_events", {})[self._slot] = done
        self._slot = (self._slot + 1) % self.depth
        self.queue.append((x, done))
        return x.shape

    def get(self):
        if not self.queue:
            raise RuntimeError("a8hip: BatchOnRamp.get() without a submitted batch")
        x, done = self.queue.popleft()
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        x.record_stream(cur)
        return x
