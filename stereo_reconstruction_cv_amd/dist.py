"""Frame sharding across the GPUs of one node: one process per GPU, stereo pairs
are independent, so the data path needs no collective.  The only communication is optional
ingestion/egress when a single rank owns the frames: point-to-point sends of u8 frames out and
int16 disparities (and, if wanted, float XYZ) back, over torch.distributed ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  The reference has no counterpart (it is a
single-process notebook, main.ipynb:780-797).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames for `rank`; the first n % world ranks get one extra frame."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(max(n_frames, 0), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def scatter_frames(frames: Optional[torch.Tensor], src: int = 0, device: Optional[torch.device] = None,
                   group=None) -> torch.Tensor:
    """Rank `src` holds frames [N, ...]; every rank returns its shard [n_r, ...] (possibly empty).

    Shards may be ragged (N not a multiple of the world size), hence point-to-point sends rather
    than dist.scatter.  Shapes and dtype are announced with one small broadcast."""
    rank, world = _world(group)
    if world == 1:
        return frames if device is None else frames.to(device)
    meta = [None]
    if rank == src:
        meta = [(tuple(frames.shape), str(frames.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(meta, src=src, group=group)
    shape, dtype_name = meta[0]
    dtype = getattr(torch, dtype_name)
    n = shape[0]
    lo, hi = shard_range(n, rank, world)
    dev = device if device is not None else (frames.device if frames is not None else torch.device("cpu"))
    if rank == src:
        reqs = []
        for r in range(world):
            if r == src:
                continue
            a, b = shard_range(n, r, world)
            if b > a:
                reqs.append(dist.isend(frames[a:b].contiguous().to(dev), dst=r, group=group))
        mine = frames[lo:hi].to(dev)
        for q in reqs:
            q.wait()
        return mine
    out = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=dev)
    if hi > lo:
        dist.recv(out, src=src, group=group)
    return out


def gather_results(local: torch.Tensor, n_frames: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames: rank `dst` returns [n_frames, ...], the others None."""
    rank, world = _world(group)
    if world == 1:
        return local
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local.contiguous(), dst=dst, group=group)
        return None
    out = torch.empty((n_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        a, b = shard_range(n_frames, r, world)
        if b <= a:
            continue
        if r == dst:
            out[a:b] = local
        else:
            dist.recv(out[a:b], src=r, group=group)
    return out


def run_sharded(compute: Callable, lefts: Optional[torch.Tensor], rights: Optional[torch.Tensor], src: int = 0,
                device: Optional[torch.device] = None, group=None):
    """scatter -> compute(local_lefts, local_rights) -> gather.  `compute` maps [n, H, W] u8 pairs
    to [n, ...] results on the same device (on the GPU box: the HIP engine's batch entry); it may
    return one tensor or a tuple of tensors (disparity, XYZ), each gathered to `src` in turn."""
    rank, world = _world(group)
    n_meta = [int(lefts.shape[0]) if rank == src and lefts is not None else 0]
    if world > 1:
        dist.broadcast_object_list(n_meta, src=src, group=group)
    l = scatter_frames(lefts, src, device, group)
    r = scatter_frames(rights, src, device, group)
    res = compute(l, r)
    if isinstance(res, (tuple, list)):
        outs = [gather_results(t, n_meta[0], src, group) for t in res]
        return tuple(outs) if rank == src or world == 1 else None
    return gather_results(res, n_meta[0], src, group)


def hip_batch_compute(params: dict, Q=None, want_float: bool = False) -> Callable:
    """compute() for run_sharded backed by the HIP engine on this rank's GPU: device tensors
    [n, H, W] u8 in, int16 disparities [n, H, W] out.  With Q (4x4) the whole driver cell runs per
    frame (sgm_pipeline_device: compute -> float scaling -> reprojectImageTo3D,
    main.ipynb:780-797) and compute returns (disparity int16, XYZ float32 [n, H, W, 3])
    -- plus the float disparity [n, H, W] in between when want_float is set."""
    import numpy as np

    from . import stereo as _cv

    Qm = None if Q is None else np.ascontiguousarray(np.asarray(Q, dtype=np.float64).reshape(4, 4))

    def compute(lefts: torch.Tensor, rights: torch.Tensor):
        if not lefts.is_cuda or not rights.is_cuda:
            raise _cv.error("hip_batch_compute needs device tensors (there is no CPU fallback)")
        if lefts.shape != rights.shape or lefts.dim() != 3 or lefts.dtype != torch.uint8 or rights.dtype != torch.uint8:
            raise _cv.error("hip_batch_compute: lefts/rights must be uint8 [n, H, W] of equal shape")
        lefts, rights = lefts.contiguous(), rights.contiguous()
        n, H, W = lefts.shape
        dev = lefts.device
        eng = _cv.get_engine(params, dev.index or 0)
        disp = torch.empty((n, H, W), dtype=torch.int16, device=dev)
        dispf = torch.empty((n, H, W), dtype=torch.float32, device=dev) if Qm is not None else None
        xyz = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev) if Qm is not None else None
        torch.cuda.current_stream(dev).synchronize()  # the engine runs on its own stream
        for i in range(n):
            if Qm is None:
                eng.compute_device(lefts[i].data_ptr(), rights[i].data_ptr(), H, W, W, disp[i].data_ptr())
            else:
                eng.pipeline_device(lefts[i].data_ptr(), rights[i].data_ptr(), H, W, W, Qm, disp[i].data_ptr(),
                                    dispf[i].data_ptr(), xyz[i].data_ptr())
        eng.synchronize()
        if Qm is None:
            return disp
        return (disp, dispf, xyz) if want_float else (disp, xyz)

    return compute
