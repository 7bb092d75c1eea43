"""Overlap-average tiling of inputs larger than `chop_size` (reference: utils/util_image.py:889-979 `ImageSpliterTh`,
driven by sampler.py:186-208).

Same iteration protocol as the reference class — `for patches, index_infos in splitter: splitter.update(out, index_infos)`,
then `splitter.gather()` — but the accumulation / normalisation runs in the engine's tile kernels
(`rs_tile_accumulate`, `rs_tile_finalize`) on NCHW fp32 device tensors.  Tiles are independent units, so `extra_bs` tiles
travel through the sampler as one batch.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _lib


def extract_starts(length: int, pch_size: int, stride: int) -> List[int]:
    """Tile origins along one axis: every `stride`, the last one pulled back so that it ends at the border
    (util_image.py:922-931)."""
    if length <= pch_size:
        return [0]
    starts = list(range(0, length, stride))
    starts = [s if s + pch_size <= length else length - pch_size for s in starts]
    out: List[int] = []
    for s in starts:  # de-duplicate, keep first occurrence order
        if s not in out:
            out.append(s)
    return out


class TileSplitter:
    def __init__(self, im: torch.Tensor, pch_size: int, stride: int, sf: int = 1, extra_bs: int = 1):
        assert stride <= pch_size
        self.lib = _lib.load()
        self.pch_size, self.stride, self.sf, self.extra_bs = pch_size, stride, sf, extra_bs
        bs, chn, height, width = im.shape
        self.true_bs = bs
        self.starts: List[Tuple[int, int]] = [(i, j) for i in extract_starts(height, pch_size, stride)
                                              for j in extract_starts(width, pch_size, stride)]
        self.count_pchs = 0
        # the crops read the image through rs_window_copy (device, contiguous fp32): convert ONCE, not per tile
        if not im.is_cuda:
            raise RuntimeError("TileSplitter needs a device tensor: the host mirror does not fall back to CPU arithmetic")
        self.im_ori = im.detach().to(torch.float32).contiguous()
        self.out_shape = None  # allocated on the first update (the output channel count is the sampler's business)
        self.im_res = None
        self.pixel_count = None

    def __len__(self) -> int:
        return len(self.starts)

    def __iter__(self):
        return self

    def __next__(self):
        if self.count_pchs >= len(self.starts):
            raise StopIteration()
        cur = self.starts[self.count_pchs:self.count_pchs + self.extra_bs]
        self.count_pchs += len(cur)
        ps, sf = self.pch_size, self.sf
        H, W = self.im_ori.shape[2:]
        # tile crops on the device (rs_window_copy), stacked along the batch axis like the reference's torch.cat of slices
        B0, C0 = self.im_ori.shape[:2]
        th, tw = min(ps, H), min(ps, W)
        pch = torch.empty(len(cur) * B0, C0, th, tw, device=self.im_ori.device, dtype=torch.float32)
        for k, (h0, w0) in enumerate(cur):
            _lib.window_copy(self.im_ori, h0, w0, th, tw, out=pch[k * B0:(k + 1) * B0])
        # a side that is <= pch_size yields a shorter tile (the slice clamps, as the reference's does, util_image.py:946-952):
        # the canvas window is clamped likewise (the reference's slice ASSIGNMENT clamps it, :962-968)
        index_infos = [[h0 * sf, min(h0 + ps, H) * sf, w0 * sf, min(w0 + ps, W) * sf] for h0, w0 in cur]
        return pch, index_infos

    def update(self, pch_res: torch.Tensor, index_infos) -> None:
        """pch_res: (len(index_infos) * true_bs) x c x (pch*sf) x (pch*sf) fp32 device tensor."""
        assert pch_res.shape[0] == self.true_bs * len(index_infos)
        pch_res = pch_res.detach().to(torch.float32).contiguous()
        if self.im_res is None:
            _, _, H, W = self.im_ori.shape
            self.im_res = torch.zeros(self.true_bs, pch_res.shape[1], H * self.sf, W * self.sf, device=pch_res.device, dtype=torch.float32)
            self.pixel_count = torch.zeros(H * self.sf, W * self.sf, device=pch_res.device, dtype=torch.float32)
        B, Cc, H, W = self.im_res.shape
        st = _lib.current_stream_ptr()
        th, tw = pch_res.shape[2:]
        for k, (h0, h1, w0, w1) in enumerate(index_infos):
            if (h1 - h0, w1 - w0) != (th, tw):   # the kernel reads the tile with row pitch tw
                raise ValueError(f"tile result is {th}x{tw} but its canvas window is {h1 - h0}x{w1 - w0}")
            tile = pch_res[k * self.true_bs:(k + 1) * self.true_bs]
            rc = self.lib.rs_tile_accumulate(self.im_res.data_ptr(), self.pixel_count.data_ptr(), tile.data_ptr(), B, Cc, H, W, h0, w0,
                                             h1 - h0, w1 - w0, st)
            _lib.check(rc, "rs_tile_accumulate")

    def gather(self) -> torch.Tensor:
        B, Cc, H, W = self.im_res.shape
        _lib.check(self.lib.rs_tile_finalize(self.im_res.data_ptr(), self.pixel_count.data_ptr(), B, Cc, H, W, _lib.current_stream_ptr()),
                   "rs_tile_finalize")
        return self.im_res
