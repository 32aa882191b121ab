"""Input prefetcher: the reference's `DataPrefetcher` (solver/clip_solver.py:30-63, switched on by `data.train.prefetch`) with the
parts it left commented out actually done, so that at > 10^4 pairs/s per GPU the host side of a batch is never on the step's
critical path:

  * a background thread pulls the next host batch from the loader, tokenises string captions with the host-thread BPE of the
    C-ABI library (`bpe.NativeTokenizer`; the ctypes call releases the GIL) and pins the tensors;
  * the host->device copies of batch i+1 are enqueued on a copy stream while step i computes; `next()` makes the caller's
    stream wait for that copy's event (no host synchronisation) and hands the device batch over;
  * images may stay uint8 HWC (the vision tower normalises bytes on the GPU, `dh_image_prep_u8`): 4x less PCIe traffic;
  * a batch may also carry decoded images at their SOURCE size on one uint8 canvas plus crop boxes (`image_boxes`, int32 [b, 8] or
    [b, views, 8] from declip_amd.augment; optional `image_flip`): RandomResizedCrop / Resize + CenterCrop, mirror, ToTensor and
    Normalize then run on the GPU right behind the copy, on the copy stream (`dh_image_resized_crop_u8`) -- the multi-view models
    get all their views from ONE upload of the decoded image, and no CPU core resizes pixels.

`next()` returns None when the loader is exhausted, as the reference's does.  Works without a GPU (device "cpu": same
ordering / tokenisation logic, plain tensors) so the host logic is covered by the CPU tests.
"""
import queue
import threading

import torch

from . import bpe, ops

__all__ = ["DataPrefetcher", "crops_on_device"]


def crops_on_device(batch, out_hw):
    """`images` uint8 [b, Hs, Ws, 3] + `image_boxes` int32 [b, 8] | [b, V, 8] (+ `image_flip` [b] | [b, V]) on the device ->
    `images` fp32 [b, 3 * V, H, W] (channel-stacked views, data/transforms.py:38-54); the box / flip entries are consumed."""
    boxes = batch.get("image_boxes")
    if boxes is None:
        return batch
    out = dict(batch)
    src = out["images"]
    boxes = out.pop("image_boxes")
    flip = out.pop("image_flip", None)
    if boxes.dim() == 2:
        boxes = boxes[:, None, :]
        flip = None if flip is None else flip[:, None]
    b, V = boxes.shape[0], boxes.shape[1]
    H, W = out_hw
    images = torch.empty(b, 3 * V, H, W, device=src.device, dtype=torch.float32)
    for v in range(V):
        ops.image_resized_crop_u8(src, boxes[:, v].contiguous(), (H, W), flip=None if flip is None else flip[:, v].contiguous(),
                                  out=images, c0=3 * v)
    out["images"] = images
    return out

_END = object()


HOST_KEYS = ("mlm_labels",)       # batch entries the step consumes on the host (never uploaded by the prefetcher)

class DataPrefetcher(object):
    def __init__(self, loader, device="cuda", tokenizer=None, context_length=77, depth=2, image_size=224, text_prep=None, rows_sync=None):
        self.device = torch.device(device)
        self.image_hw = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        self.tokenizer, self.context_length = tokenizer, context_length
        # text_prep(captions) -> dict of batch entries: everything a model does to caption STRINGS before its text tower, moved to
        # this worker thread (DECLIP.prepare_captions: caption sampling, EDA augmentation, BPE, MLM masking -- declip.py:203-230
        # runs them inside forward(), a per-caption Python loop on the critical path)
        self.text_prep = text_prep
        # rows_sync(rows) -> padded packed row count of the batch that EVERY rank of a data-parallel job will use (dist.RowsSync: MAX
        # over the ranks of each rank's own padded count, a host-side collective on this worker thread, one batch ahead of the step):
        # the key of the captured step (engine.packed_key) is then the same on all ranks, so they capture and replay in lock-step
        self.rows_sync = rows_sync
        self._it = iter(loader)
        self._q = queue.Queue(maxsize=max(1, depth))
        self._cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._staged, self._event = None, None
        self._stop = False
        self._thread = threading.Thread(target=self._work, name="declip-prefetch", daemon=True)
        self._thread.start()
        self._stage()

    # ---- worker thread: host-side preparation -------------------------------------------------------------------------
    def _prepare(self, batch):
        out = dict(batch)
        caps = out.get("captions")
        if self.text_prep is not None and caps is not None and not torch.is_tensor(caps):
            out.update(self.text_prep(caps))
        caps = out.get("captions")
        if self.tokenizer is not None and caps is not None and not torch.is_tensor(caps):
            # the reference's batches carry a list of captions per sample and use the first one (clip.py:110-111)
            texts = [c if isinstance(c, str) else c[0] for c in caps]
            out["captions"] = bpe.tokenize(self.tokenizer, texts, self.context_length)
        caps = out.get("captions")
        if torch.is_tensor(caps) and caps.dtype == torch.int64 and not caps.is_cuda:
            # packed captions (DH_TEXT_PACKED, engine.PackedCaptions) size their buffers by the number of caption rows up to
            # <|endoftext|>: counted HERE, on the host copy, so that the step never reads it back from the device
            # (numpy, NOT torch: a torch CPU reduction wakes an OpenMP team sized by the node's visible cores, whose spin-waiting
            # burns a container's CPU quota and gets the enqueueing thread throttled -- hostinfo.py)
            out["_caption_rows"] = int((caps.numpy().argmax(axis=-1) + 1).sum())
            if self.rows_sync is not None:
                out["_caption_rows_pad"] = int(self.rows_sync(out["_caption_rows"]))
        if self._cuda:
            for k, v in out.items():
                if torch.is_tensor(v) and not v.is_cuda and not v.is_pinned():
                    out[k] = v.pin_memory()
        return out

    @staticmethod
    def _tag_rows(batch):
        rows = batch.pop("_caption_rows", None)
        rows_pad = batch.pop("_caption_rows_pad", None)
        caps = batch.get("captions")
        if rows is not None and torch.is_tensor(caps):
            caps._dh_rows = (caps._version, rows)       # all captions of the tensor ([b, ctx], or every variant of [b, k, ctx])
            if rows_pad is not None:
                caps._dh_rows_pad = (caps._version, rows_pad)   # the job-wide padded row count (engine.PackedCaptions pads up to it)
        return batch

    def _work(self):
        try:
            for batch in self._it:
                item = self._prepare(batch)
                while not self._stop:                # (a bounded put: close() must be able to end a worker that is ahead of the step)
                    try:
                        self._q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        pass
                if self._stop:
                    return
            self._q.put(_END)
        except BaseException as e:       # surfaces in the training thread at the next next()
            self._q.put(e)

    def close(self):
        """Stop the worker thread (an endless loader never ends it) and drop what it prepared; the prefetcher yields nothing afterwards."""
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)
        self._staged = None

    # ---- training thread ---------------------------------------------------------------------------------------------
    def _stage(self):
        """Take the next prepared host batch and enqueue its copies on the copy stream."""
        item = self._q.get()
        if item is _END:
            self._staged = None
            self._q.put(_END)            # stays exhausted
            return
        if isinstance(item, BaseException):
            self._staged = item
            return
        if not self._cuda:
            self._staged = self._tag_rows(item)
            return
        with torch.cuda.stream(self.stream):
            # `mlm_labels` stay on the HOST: the masked-LM head takes its row selection from them with index arithmetic on the
            # labels' own device (heads._mlm_selection) -- on a device tensor that is a nonzero() read-back, i.e. a host stall
            # on the whole queue in every DeCLIP / DeFILIP step
            dev = {k: (v.to(self.device, non_blocking=True) if (torch.is_tensor(v) and k not in HOST_KEYS) else v) for k, v in item.items()}
            dev = self._tag_rows(dev)
            dev = crops_on_device(dev, self.image_hw)      # resize / mirror / normalise behind the copy, on the copy stream
            self._event = torch.cuda.Event()
            self._event.record(self.stream)
        self._staged = dev

    def next(self):
        batch = self._staged
        if batch is None:
            return None
        if isinstance(batch, BaseException):
            raise batch
        if self._cuda:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._event)
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)          # allocated under the copy stream, consumed on this one
        self._stage()                             # batch i+1's copies overlap step i
        return batch

    def __iter__(self):
        return self

    def __next__(self):
        b = self.next()
        if b is None:
            raise StopIteration
        return b
