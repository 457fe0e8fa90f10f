"""Multi-GPU image-space sharding: one process per GPU, volume replicated, each rank renders
the 64x64 tiles t with t % world == rank into its own tile-major slab; the framebuffer is
assembled only when an image is needed, with ONE collective: all_gather of the equal-sized
slabs over RCCL/xGMI (backend "nccl"), then a de-tiling kernel.
The reference has no multi-device notion; seeds depend only on the global pixel index and the
frame (fragment.frag:143), so the gathered image is bit-identical to a 1-GPU render.
"""
from __future__ import annotations

import numpy as np

from . import tiles


class _DevPtr:
    """__cuda_array_interface__ view of a raw device pointer (zero copy into torch)."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False),
                                         "version": 2, "strides": None}


def slab_tensor(renderer):
    """torch view of the renderer's device slab (no copy)."""
    import torch
    n, _ = renderer.slab_info()
    return torch.as_tensor(_DevPtr(renderer.slab_device_ptr(), n), device=f"cuda:{torch.cuda.current_device()}")


def gather_image(slab, width: int, height: int, renderer=None, group=None, perm=None):
    """all_gather the per-rank slabs and de-tile.  `slab` is a 1-D float tensor (CUDA for the
    product path; CPU tensors are accepted so the partition/gather logic can be exercised with
    the gloo backend; `perm` = the tile dealing order then, see tiles.balanced_order).  Returns an
    [H, W, 4] tensor on the slab's device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        gathered = torch.empty(world * slab.numel(), dtype=slab.dtype, device=slab.device)
        dist.all_gather_into_tensor(gathered, slab.contiguous(), group=group)
    else:
        gathered = slab
    if slab.is_cuda:
        if renderer is None:
            raise ValueError("a renderer is needed to de-tile on the device")
        image = torch.empty(height * width * 4, dtype=torch.float32, device=slab.device)
        torch.cuda.current_stream().synchronize()
        renderer.detile(gathered.data_ptr(), image.data_ptr())
        renderer.finish()
        return image.view(height, width, 4)
    img = tiles.detile_numpy(gathered.numpy().reshape(world, -1, 4), width, height, world, perm)
    return torch.from_numpy(img)
