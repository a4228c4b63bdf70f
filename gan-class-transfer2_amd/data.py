"""Input pipeline of the reference (train.py:285-297, 313-321): list_files -> shuffle(1000) -> repeat -> decode_file
(decode, random crop to size x size, random left-right flip, value/128 - 1, returned as the pair (image, image)) -> batch ->
prefetch.

Decoding is host work (PIL, like tf.image.decode_jpeg it accepts JPEG and PNG and always yields 3 channels); everything
after it runs on the GPU in one launch per batch (gct2_image_prepare): the decoded bytes of a batch are uploaded once, the crop
origins and flip flags are drawn on the host from a seeded numpy generator.  A background thread keeps `prefetch` batches
ready (tf.data.AUTOTUNE in the reference)."""
from __future__ import annotations

import glob
import queue
import threading
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import call


def shuffle_repeat(items: Sequence, buffer_size: int, rng: np.random.Generator) -> Iterator:
    """tf.data's .shuffle(buffer_size).repeat(): a buffer of `buffer_size` elements is kept filled from the (endlessly repeated,
    per-epoch reshuffled like list_files) source and a uniformly random slot is emitted and refilled."""
    if not len(items):
        raise ValueError("empty dataset")

    def source():
        while True:
            order = rng.permutation(len(items))          # list_files shuffles the file order every epoch
            for i in order:
                yield items[i]

    src = source()
    buf = [next(src) for _ in range(min(buffer_size, len(items)))]
    while True:
        j = int(rng.integers(len(buf)))
        out, buf[j] = buf[j], next(src)
        yield out


def decode_rgb(path: str) -> np.ndarray:
    """tf.image.decode_jpeg(file, 3) (train.py:287): uint8 [H, W, 3]."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class ImageDataset:
    """iterable of (image, image) batches, fp32 [B, size, size, 3] in [-1, 1) on the HIP device (train.py:292-293, 316-320).

    source: a glob pattern (train.py:5, 313) or a list of decoded uint8 HWC arrays (synthetic / in-memory data)."""

    def __init__(self, source, size: int, batch_size: int, device: Optional[torch.device] = None, seed: int = 0,
                 shuffle_buffer: int = 1000, prefetch: int = 2, crop: bool = True, decoder: Callable[[str], np.ndarray] = decode_rgb):
        self.items = sorted(glob.glob(source)) if isinstance(source, str) else list(source)
        if not self.items:
            raise ValueError(f"no files match {source!r}")
        self.size, self.batch_size, self.crop, self.decoder = size, batch_size, crop, decoder
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.rng = np.random.default_rng(seed)
        self.shuffle_buffer, self.prefetch = shuffle_buffer, prefetch

    # ---- host side: pick, decode, draw the augmentation --------------------------------------------------------------------
    def host_batches(self) -> Iterator[Tuple[List[np.ndarray], np.ndarray]]:
        stream = shuffle_repeat(self.items, self.shuffle_buffer, self.rng)
        while True:
            imgs, dims = [], np.zeros((self.batch_size, 5), dtype=np.int32)
            for b in range(self.batch_size):
                it = next(stream)
                im = self.decoder(it) if isinstance(it, str) else np.ascontiguousarray(it, dtype=np.uint8)
                H0, W0 = im.shape[:2]
                if im.ndim != 3 or im.shape[2] != 3 or H0 < self.size or W0 < self.size:
                    raise ValueError(f"image of shape {im.shape} cannot be cropped to {self.size}x{self.size}x3 (tf.image.random_crop "
                                     "raises here, train.py:289)")
                if not self.crop and (H0, W0) != (self.size, self.size):
                    raise ValueError("crop=False needs size x size images (tf.broadcast_to, train.py:290)")
                oy = int(self.rng.integers(H0 - self.size + 1)) if self.crop else 0
                ox = int(self.rng.integers(W0 - self.size + 1)) if self.crop else 0
                dims[b] = (H0, W0, oy, ox, int(self.rng.integers(2)))       # random_flip_left_right: p = 1/2
                imgs.append(im)
            yield imgs, dims

    # ---- device side ------------------------------------------------------------------------------------------------------
    def to_device(self, imgs: List[np.ndarray], dims: np.ndarray, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        sizes = np.array([im.size for im in imgs], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        packed = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs]))
        st = stream or torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            src = packed.to(self.device, non_blocking=True)
            off_d = torch.from_numpy(offsets).to(self.device, non_blocking=True)
            dims_d = torch.from_numpy(np.ascontiguousarray(dims)).to(self.device, non_blocking=True)
            out = torch.empty(len(imgs), self.size, self.size, 3, dtype=torch.float32, device=self.device)
            call("gct2_image_prepare", src.data_ptr(), off_d.data_ptr(), dims_d.data_ptr(), out.data_ptr(), len(imgs), self.size,
                 st.cuda_stream)
            for t in (src, off_d, dims_d):
                t.record_stream(st)
        return out

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        q: "queue.Queue" = queue.Queue(maxsize=max(1, self.prefetch))
        stop = threading.Event()
        # the loader's own stream, picked like the engine's (engine.distinct_stream: one that shares a hardware queue neither with the
        # consumer's stream nor with the streams a train step of this consumer already runs on)
        from .engine import distinct_stream
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        side = distinct_stream(dev, "data", torch.cuda.current_stream(dev))

        def worker():
            try:
                for imgs, dims in self.host_batches():
                    if stop.is_set():
                        return
                    x = self.to_device(imgs, dims, side)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    q.put((x, ev))
            except BaseException as e:          # surface decoder / shape errors in the consumer
                q.put(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                x, ev = item
                torch.cuda.current_stream(self.device).wait_event(ev)
                x.record_stream(torch.cuda.current_stream(self.device))
                yield x, x                      # labels = the image itself (train.py:293)
        finally:
            stop.set()
