"""Device-side input pipeline: what the reference's Dataset / collate code does on the host in fp32 runs here on
the GPU from the raw uint8 bytes, so the H2D copy carries 1 byte per pixel instead of 4 and no host core touches the pixels.

Same names, argument meaning and results as the reference functions; every function takes/returns device tensors and has no CPU path:

* ``transform_image``   ocrs_models/datasets/util.py:27-35        uint8 -> float in [-0.5, 0.5]
* ``resize_line``       ocrs_models/datasets/hiertext.py:288-294  antialiased resize of a line crop to ``output_height`` rows, width by aspect
                                                                  ratio clamped to [10, 800]
* ``collate_samples``   ocrs_models/train_rec.py:248-304          bucketed right-padding (``round_up`` quirk, pad value 0.0, infeasible-sample
                                                                  drop rule, same keys/dtypes); images may be raw uint8 (transform fused)
"""
from __future__ import annotations

import torch

from ._lib import lib, ptr
from .text import ctc_input_and_target_compatible, round_up

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a tensor on the GPU (this pipeline has no CPU path; use ocrs_models_amd.text for host code)")


def transform_image(img: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """8-bit greyscale CHW (uint8, on the GPU) -> float CHW with values in [-0.5, 0.5]."""
    _need_cuda(img, "transform_image")
    if img.dtype != torch.uint8:
        raise RuntimeError("transform_image: expected a uint8 image")
    img = img.contiguous()
    out = torch.empty(img.shape, dtype=dtype, device=img.device)
    lib().transform_image_u8(ptr(img), ptr(out), img.numel(), _DT[dtype])
    return out


def line_output_width(line_height: int, line_width: int, output_height: int = 64) -> int:
    """Width rule of hiertext.py:288-292: scale with the height, at least 10 (never zero-width), at most 800 (bounds batch memory)."""
    return min(800, max(10, int(output_height * (line_width / line_height))))


def resize(img: torch.Tensor, size) -> torch.Tensor:
    """``torchvision.transforms.functional.resize(img, [h, w], antialias=True)`` for a float (C,H,W) or (N,C,H,W) GPU tensor."""
    _need_cuda(img, "resize")
    if img.dtype != torch.float32:
        raise RuntimeError("resize: expected a float32 image")
    oh, ow = int(size[0]), int(size[1])
    h, w = img.shape[-2:]
    planes = img.numel() // (h * w)
    src = img.contiguous()
    out = torch.empty(*img.shape[:-2], oh, ow, dtype=torch.float32, device=img.device)
    ws = torch.empty(lib().resize_aa_ws_floats(planes, h, ow), dtype=torch.float32, device=img.device)
    lib().resize_aa(ptr(src), ptr(ws), ptr(out), planes, h, w, oh, ow)
    return out


def resize_line(line_img: torch.Tensor, output_height: int = 64) -> torch.Tensor:
    """(1,h,w) float line crop -> (1, output_height, line_output_width(h, w)) (hiertext.py:288-294)."""
    _, h, w = line_img.shape
    return resize(line_img, [output_height, line_output_width(h, w, output_height)])


def collate_samples(samples: list[dict], device, dtype: torch.dtype = torch.float32) -> dict:
    """list of {'image': (1,H,w) uint8 or float32 HOST tensor, 'text_seq': (L,) int32} -> padded batch dict of train_rec.py:248-304 with
    ``image`` on ``device``.

    uint8 images are raw pixels (the kernel applies ``transform_image`` while it pads); float32 images are the reference's
    already-transformed samples.  One packed H2D copy + one kernel, instead of B strided host copies into a 4-byte-per-pixel batch.
    """
    wmax = round_up(max(s["image"].shape[-1] for s in samples), 256)
    lmax = round_up(max(s["text_seq"].shape[0] for s in samples), 64)
    keep = [s for s in samples if ctc_input_and_target_compatible(s["image"].shape[-1] // 4, s["text_seq"])]
    n = len(keep)
    h = keep[0]["image"].shape[1] if keep else 64
    text = torch.zeros(n, lmax, dtype=torch.int32)
    tl = torch.zeros(n, dtype=torch.int64)
    iw = torch.zeros(n, dtype=torch.int64)
    for i, s in enumerate(keep):
        L = s["text_seq"].shape[0]
        text[i, :L] = s["text_seq"]
        tl[i], iw[i] = L, s["image"].shape[-1]
    image = torch.empty(n, 1, h, wmax, dtype=dtype, device=device)
    if n:
        kinds = {s["image"].dtype for s in keep}
        if kinds not in ({torch.uint8}, {torch.float32}):
            raise RuntimeError(f"collate_samples: images must be all uint8 or all float32, got {kinds}")
        if any(s["image"].shape[0] != 1 or s["image"].shape[1] != h for s in keep):
            raise RuntimeError("collate_samples: every image must be (1, H, w) with the same H")
        packed = torch.cat([s["image"].reshape(-1) for s in keep])
        sizes = torch.tensor([s["image"].numel() for s in keep], dtype=torch.int64)
        offs = torch.cumsum(sizes, 0) - sizes
        packed_d = packed.to(device, non_blocking=True)
        offs_d = offs.to(device, non_blocking=True)
        widths_d = iw.to(torch.int32).to(device, non_blocking=True)
        lib().collate_pad(ptr(packed_d), ptr(offs_d), ptr(widths_d), ptr(image), n, h, wmax, 0 if torch.uint8 in kinds else 1, _DT[dtype])
    return {"image": image, "text_seq": text, "text_len": tl, "image_width": iw}
