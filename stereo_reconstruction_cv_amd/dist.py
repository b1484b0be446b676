"""Frame sharding across the GPUs of one node: one process per GPU, stereo pairs
are independent, so the data path needs no collective.  The only communication is optional
ingestion/egress when a single rank owns the frames: point-to-point transfers of u8 frames out and
int16 disparities (and, if wanted, float XYZ or the compacted point list) back, over
torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The reference
has no counterpart (it is a single-process notebook, main.ipynb:780-797).

xGMI is point to point: rank 0 has one link to each of its seven peers (about 153 GB/s each).  A
scatter or gather therefore posts ALL its transfers as ONE batch (dist.batch_isend_irecv: a
coalesced group on RCCL), so that the seven links carry their shards at the same time instead of one
after the other; and `IngestPipeline` keeps two batches in flight, so that the gather of batch k and
the scatter of batch k + 2 run beside the compute of batch k + 1.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

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


class _Batch:
    """One group of point-to-point transfers, posted together (all links at once)."""

    def __init__(self, group=None):
        self.group = group
        self.ops: List[dist.P2POp] = []
        self.reqs = []

    def send(self, t: torch.Tensor, dst: int):
        if t.numel():
            self.ops.append(dist.P2POp(dist.isend, t, dst, self.group))

    def recv(self, t: torch.Tensor, src: int):
        if t.numel():
            self.ops.append(dist.P2POp(dist.irecv, t, src, self.group))

    def post(self):
        if self.ops:
            self.reqs = dist.batch_isend_irecv(self.ops)
        return self

    def wait(self):
        # NCCL: makes the CURRENT stream wait (the host does not block); gloo: blocks until done
        for q in self.reqs:
            q.wait()
        self.reqs = []


def _announce(frames: Optional[torch.Tensor], src: int, group=None):
    """(shape, dtype) of the tensor rank `src` holds, on every rank (one small broadcast)."""
    rank, _ = _world(group)
    meta = [None]
    if rank == src:
        meta = [(tuple(frames.shape), str(frames.dtype).replace("torch.", ""))]
    dist.broadcast_object_list(meta, src=src, group=group)
    shape, dtype_name = meta[0]
    return tuple(shape), getattr(torch, dtype_name)


def post_scatter(frames: Optional[torch.Tensor], shape, dtype, src: int, device, group=None,
                 out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, _Batch]:
    """Post the scatter of frames [N, ...] (held by `src`) as one batch; returns (this rank's shard, the batch
    to wait for).  Shards may be ragged (N not a multiple of the world size), hence point to point."""
    rank, world = _world(group)
    n = shape[0]
    lo, hi = shard_range(n, rank, world)
    b = _Batch(group)
    if rank == src:
        for r in range(world):
            if r == src:
                continue
            a, e = shard_range(n, r, world)
            if e > a:
                b.send(frames[a:e].contiguous().to(device), r)
        mine = frames[lo:hi].to(device)
        return mine, b.post()
    if out is None or tuple(out.shape) != (hi - lo,) + tuple(shape[1:]) or out.dtype != dtype:
        out = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=device)
    b.recv(out, src)
    return out, b.post()


def scatter_frames(frames: Optional[torch.Tensor], src: int = 0, device: Optional[torch.device] = None,
                   group=None) -> torch.Tensor:
    """Rank `src` holds frames [N, ...]; every rank returns its shard [n_r, ...] (possibly empty).
    Shapes and dtype are announced with one small broadcast; all shards travel as one batch."""
    rank, world = _world(group)
    if world == 1:
        return frames if device is None else frames.to(device)
    shape, dtype = _announce(frames, src, group)
    dev = device if device is not None else (frames.device if frames is not None else torch.device("cpu"))
    out, b = post_scatter(frames, shape, dtype, src, dev, group)
    b.wait()
    return out


def post_gather(local: torch.Tensor, n_frames: int, dst: int, group=None,
                out: Optional[torch.Tensor] = None) -> Tuple[Optional[torch.Tensor], _Batch]:
    """Inverse of post_scatter: rank `dst` posts every receive as one batch (all links at once)."""
    rank, world = _world(group)
    b = _Batch(group)
    if rank != dst:
        if local.shape[0] > 0:
            b.send(local.contiguous(), dst)
        return None, b.post()
    want = (n_frames,) + tuple(local.shape[1:])
    if out is None or tuple(out.shape) != want or out.dtype != local.dtype:
        out = torch.empty(want, dtype=local.dtype, device=local.device)
    for r in range(world):
        a, e = shard_range(n_frames, r, world)
        if e <= a:
            continue
        if r == dst:
            out[a:e] = local
        else:
            b.recv(out[a:e], r)
    return out, b.post()


def gather_results(local: torch.Tensor, n_frames: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames: rank `dst` returns [n_frames, ...], the others None."""
    rank, world = _world(group)
    if world == 1:
        return local
    out, b = post_gather(local, n_frames, dst, group)
    b.wait()
    return out


def run_sharded(compute: Callable, lefts: Optional[torch.Tensor], rights: Optional[torch.Tensor], src: int = 0,
                device: Optional[torch.device] = None, group=None):
    """scatter -> compute(local_lefts, local_rights) -> gather, one batch, nothing overlapped (IngestPipeline
    overlaps).  `compute` maps [n, H, W] u8 pairs to [n, ...] results on the same device (on the GPU box: the
    HIP engine's batch entry); it may return one tensor or a tuple of tensors (disparity, XYZ), each gathered to
    `src` -- all of them in one batch of transfers."""
    rank, world = _world(group)
    if world == 1:
        l = lefts if device is None else lefts.to(device)
        r = rights if device is None else rights.to(device)
        return compute(l, r)
    shape, dtype = _announce(lefts, src, group)
    dev = device if device is not None else (lefts.device if lefts is not None else torch.device("cpu"))
    l, bl = post_scatter(lefts, shape, dtype, src, dev, group)
    r, br = post_scatter(rights, shape, dtype, src, dev, group)
    bl.wait()
    br.wait()
    res = compute(l, r)
    many = isinstance(res, (tuple, list))
    posted = [post_gather(t, shape[0], src, group) for t in (res if many else [res])]
    for _, b in posted:
        b.wait()
    if rank != src:
        return None
    outs = tuple(o for o, _ in posted)
    return outs if many else outs[0]


class _Streams:
    """Stream plumbing of the pipeline; every method is a no-op for CPU tensors (gloo tests)."""

    def __init__(self, device: torch.device, compute_stream=None):
        self.cuda = device.type == "cuda"
        self.device = device
        self.comm = torch.cuda.Stream(device) if self.cuda else None
        self.compute = compute_stream

    def on_comm(self):
        import contextlib
        return torch.cuda.stream(self.comm) if self.cuda else contextlib.nullcontext()

    def event_after_compute(self):
        if not self.cuda or self.compute is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.compute)
        return ev

    def comm_waits(self, ev):
        if ev is not None:
            self.comm.wait_event(ev)

    def event_on_comm(self):
        """marks "everything the communication stream has been told to wait for so far has happened" (called right after
        a scatter's wait: its frames have landed)"""
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(self.comm)
        return ev

    def compute_waits(self, ev):
        # an EVENT, not wait_stream(comm): the communication stream also carries the later gathers and scatters,
        # which the kernels of this batch must not wait for
        if ev is not None and self.compute is not None:
            self.compute.wait_event(ev)


class IngestPipeline:
    """Rank `src` owns the frames of every batch; results come back to it.  Two batches in flight:

        step k:   post gather(k - 2)   |  post scatter(k)   |  compute(k - 1) enqueued behind scatter(k - 1)

    so the transfers of batches k - 2 and k run beside the kernels of batch k - 1.  On the GPU box transfers are
    enqueued on a communication stream of their own and ordered against the compute stream with events only (the
    host never blocks inside a step); `compute(l, r)` must enqueue its work on `compute_stream` and return its
    result tensors without synchronising.  Every rank must call `step` the same number of times with batches of
    the same frame count in the same order (`src` passes the tensors, the others None) and finish with `drain`.

    Frame shape and dtype are announced once (`first batch`); later batches must keep the shape.
    """

    def __init__(self, compute: Callable, src: int = 0, device: Optional[torch.device] = None, group=None,
                 compute_stream=None):
        self.compute, self.src, self.group = compute, src, group
        self.rank, self.world = _world(group)
        self.device = device if device is not None else torch.device("cpu")
        self.s = _Streams(self.device, compute_stream)
        self.shape = self.dtype = None
        self.k = 0
        self.scattered = {}    # k -> (l, r, batch_l, batch_r)
        self.computed = {}     # k -> (results tuple, many?, event)
        self.gathering = {}    # k -> (outs, batches)
        self.done: List = []   # on src: results in batch order
        self.in_bufs = [[None, None], [None, None], [None, None]]   # receive buffers, 3 deep (scatter k + 2 beside compute k + 1 beside gather k)

    def _post_scatter(self, k, lefts, rights):
        if self.world == 1:
            self.scattered[k] = (lefts.to(self.device), rights.to(self.device), None, None)
            return
        if self.shape is None:
            self.shape, self.dtype = _announce(lefts, self.src, self.group)
        with self.s.on_comm():
            buf = self.in_bufs[k % 3]
            l, bl = post_scatter(lefts, self.shape, self.dtype, self.src, self.device, self.group, out=buf[0])
            r, br = post_scatter(rights, self.shape, self.dtype, self.src, self.device, self.group, out=buf[1])
            if self.rank != self.src:
                buf[0], buf[1] = l, r
            bl.wait()     # (NCCL: the communication stream waits, not the host)
            br.wait()
            landed = self.s.event_on_comm()
        self.scattered[k] = (l, r, landed, None)

    def _compute(self, k):
        l, r, landed, _ = self.scattered.pop(k)
        self.s.compute_waits(landed)
        res = self.compute(l, r)
        many = isinstance(res, (tuple, list))
        self.computed[k] = (tuple(res) if many else (res,), many, self.s.event_after_compute())

    def _post_gather(self, k):
        res, many, ev = self.computed.pop(k)
        if self.world == 1:
            self.done.append(res if many else res[0])
            return
        n = self.shape[0]
        with self.s.on_comm():
            self.s.comm_waits(ev)
            posted = [post_gather(t, n, self.src, self.group) for t in res]
        self.gathering[k] = (posted, many)

    def _finish_gather(self, k):
        if k not in self.gathering:    # (a single process gathers nothing)
            return
        posted, many = self.gathering.pop(k)
        with self.s.on_comm():
            for _, b in posted:
                b.wait()
        if self.rank == self.src:
            outs = tuple(o for o, _ in posted)
            self.done.append(outs if many else outs[0])

    def step(self, lefts: Optional[torch.Tensor], rights: Optional[torch.Tensor]):
        """Feed batch k.  Order inside a step (the same on every rank, so that transfers pair up): finish gather(k - 3),
        post gather(k - 2), post scatter(k), enqueue compute(k - 1)."""
        k = self.k
        if k - 3 in self.gathering:
            self._finish_gather(k - 3)
        if k - 2 in self.computed:
            self._post_gather(k - 2)
        self._post_scatter(k, lefts, rights)
        if k - 1 in self.scattered:
            self._compute(k - 1)
        self.k += 1

    def drain(self):
        """Finish everything in flight; returns (on `src`) the list of per-batch results, else an empty list."""
        k = self.k
        for j in (k - 3, k - 2, k - 1):
            if j in self.gathering:
                self._finish_gather(j)
        if k - 2 in self.computed:
            self._post_gather(k - 2)
        if k - 1 in self.scattered:
            self._compute(k - 1)
        if k - 2 in self.gathering:
            self._finish_gather(k - 2)
        if k - 1 in self.computed:
            self._post_gather(k - 1)
            self._finish_gather(k - 1)
        if self.s.cuda:
            self.s.comm.synchronize()
            if self.s.compute is not None:
                self.s.compute.synchronize()
        out, self.done = self.done, []
        self.k = 0
        return out


def hip_batch_compute(params: dict, Q=None, want_float: bool = False, schedule: Optional[int] = None,
                      stream=None, synchronize: bool = True, compact: bool = False) -> Callable:
    """compute() for run_sharded / IngestPipeline backed by the HIP engine on this rank's GPU: device tensors
    [n, H, W] u8 in, int16 disparities [n, H, W] out.  With Q (4x4) the whole driver cell runs per
    frame (compute -> float scaling -> reprojectImageTo3D, main.ipynb:780-797) and compute returns
    (disparity int16, XYZ float32 [n, H, W, 3]) -- plus the float disparity [n, H, W] in between when
    want_float is set.

    schedule: SGM_OPT_SCHEDULE of the engine (2 = throughput mode: the pairs of a call share one chained sweep
    launch per pass, sgm_pipeline_batch_device).  stream: a torch.cuda.Stream the engine works on (default: a
    stream of its own); synchronize=False returns as soon as the work is enqueued (IngestPipeline orders it with
    events).  compact=True (needs Q): instead of the dense XYZ image -- 12 bytes per pixel, 99.5 MB per 4K pair --
    the call returns (disparity, points [n, H*W, 3] float32, counts [n] int64): the valid points of every frame
    (main.ipynb:726-737: finite X and disparity > 0) packed to the front of its row, the rest of the row zero;
    a consumer that only wants the cloud gathers counts first and then counts[i] points per frame
    (gather_compacted)."""
    import numpy as np

    from . import _lib as _l
    from . import stereo as _cv

    Qm = None if Q is None else np.ascontiguousarray(np.asarray(Q, dtype=np.float64).reshape(4, 4))
    if compact and Qm is None:
        raise _cv.error("hip_batch_compute(compact=True) needs Q")
    own = {}

    def engine(dev):
        key = dev.index or 0
        if key not in own:
            if stream is None and schedule is None:
                own[key] = _cv.get_engine(params, key)
            else:
                e = _cv.Engine(params, key, stream=stream.cuda_stream if stream is not None else None)
                if schedule is not None:
                    e.set_option(_l.SGM_OPT_SCHEDULE, schedule)
                own[key] = e
        return own[key]

    def compute(lefts: torch.Tensor, rights: torch.Tensor):
        if not lefts.is_cuda or not rights.is_cuda:
            raise _cv.error("hip_batch_compute needs device tensors (there is no CPU fallback)")
        if lefts.shape != rights.shape or lefts.dim() != 3 or lefts.dtype != torch.uint8 or rights.dtype != torch.uint8:
            raise _cv.error("hip_batch_compute: lefts/rights must be uint8 [n, H, W] of equal shape")
        lefts, rights = lefts.contiguous(), rights.contiguous()
        n, H, W = lefts.shape
        dev = lefts.device
        eng = engine(dev)
        disp = torch.empty((n, H, W), dtype=torch.int16, device=dev)
        dispf = torch.empty((n, H, W), dtype=torch.float32, device=dev) if Qm is not None else None
        xyz = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev) if Qm is not None else None
        if stream is None:
            torch.cuda.current_stream(dev).synchronize()  # the engine runs on its own stream
        else:
            # the engine reads / writes these on `stream`, which the caching allocator does not know about: without this
            # a tensor the caller drops right after the (asynchronous) call is handed out again while kernels still use it
            for t in (lefts, rights, disp, dispf, xyz):
                if t is not None:
                    t.record_stream(stream)
        if n:
            ptrs = lambda t: [t[i].data_ptr() for i in range(n)]
            eng.pipeline_batch_device(ptrs(lefts), ptrs(rights), H, W, W, Qm, ptrs(disp),
                                      ptrs(dispf) if Qm is not None else None, ptrs(xyz) if Qm is not None else None)
        if compact:
            pts = torch.zeros((n, H * W, 3), dtype=torch.float32, device=dev)
            counts = torch.zeros((n,), dtype=torch.int64)
            for i in range(n):     # (synchronises per frame: the count comes back to the host)
                counts[i] = eng.compact_points_device(xyz[i].data_ptr(), dispf[i].data_ptr(), None, H * W, pts[i].data_ptr(), None)
            return disp, pts, counts.to(dev)
        if synchronize:
            eng.synchronize()
        if Qm is None:
            return disp
        return (disp, dispf, xyz) if want_float else (disp, xyz)

    return compute


def gather_compacted(points: torch.Tensor, counts: torch.Tensor, n_frames: int, dst: int = 0, group=None):
    """Gather of compacted point lists (hip_batch_compute(compact=True)): counts first (8 bytes per frame), then
    exactly counts[i] points of every frame -- for a 4K pair with a third of its pixels valid 33 MB instead of the
    99.5 MB of the dense XYZ image.  Rank `dst` returns (list of [counts[i], 3] tensors, counts), the others None."""
    rank, world = _world(group)
    allc = gather_results(counts, n_frames, dst, group)
    if world == 1:
        return [points[i, :int(counts[i])] for i in range(points.shape[0])], counts
    b = _Batch(group)
    if rank != dst:
        for i in range(points.shape[0]):
            b.send(points[i, :int(counts[i])].contiguous(), dst)
        b.post().wait()
        return None
    outs: List[Optional[torch.Tensor]] = [None] * n_frames
    for r in range(world):
        a, e = shard_range(n_frames, r, world)
        for i in range(a, e):
            c = int(allc[i])
            if r == dst:
                outs[i] = points[i - a, :c]
            else:
                outs[i] = torch.empty((c, 3), dtype=points.dtype, device=points.device)
                b.recv(outs[i], r)
    b.post().wait()
    return outs, allc
