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

    def used_by_comm(self, *tensors):
        """tensors allocated on another stream that the communication stream is about to read or write"""
        if self.cuda:
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self.comm)

    def used_by_compute(self, *tensors):
        if self.cuda and self.compute is not None:
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self.compute)

    def compute_waits(self, ev):
        # an EVENT, not wait_stream(comm): the communication stream also carries the later gathers and scatters,
        # which the kernels of this batch must not wait for
        if ev is not None and self.compute is not None:
            self.compute.wait_event(ev)


class IngestPipeline:
    """Rank `src` owns the frames of every batch; results come back to it.  Two batches in flight:

        step k:   finish gather(k - 3)  |  post gather(k - 2)  |  post scatter(k)  |  compute(k - 1) enqueued behind scatter(k - 1)

    so the transfers of batches k - 2 and k run beside the kernels of batch k - 1.  On the GPU box transfers are
    enqueued on a communication stream of their own and ordered against the compute stream with events only (the
    host never blocks inside a step); `compute(l, r)` must enqueue its work on `compute_stream` and return its
    result tensors without synchronising.  Every rank must call `step` the same number of times with batches of
    the same frame count in the same order (`src` passes the tensors, the others None) and finish with `drain`.

    Tensors cross streams here, and torch's caching allocator knows only the stream a tensor was allocated on: every
    tensor that another stream reads or writes is announced with `record_stream` -- a result tensor (allocated on the
    compute stream) before the communication stream sends or copies it, a scattered shard (allocated on the
    communication stream) before the compute stream reads it -- so a block is never handed out again while a
    transfer or a kernel of the other stream is still pending on it, whenever its last reference dies.

    compact=True: `compute` returns (..., points [n, M, 3], counts [n] int64) -- the valid points of every frame packed
    to the front of its row and their number (hip_batch_compute(compact=True); main.ipynb:726-737) -- and instead of
    rows of M points only counts[i] points per frame travel.  A transfer needs its size on both sides when it is
    posted, so the points follow the counts one step later:

        step k:   finish points(k - 4)  |  finish gather(k - 3) [dense results + counts], post points(k - 3)  |  ... as above

    Posting points(k - 3) reads counts on the host, i.e. waits until compute(k - 3) has finished -- while compute(k - 2),
    enqueued a step earlier, keeps the GPU busy: one host wait per step, none per frame.  `src` gets, per batch,
    (dense results ..., [points of frame i: (counts[i], 3)], counts [N]).

    Frame shape and dtype are announced once (`first batch`); later batches must keep the shape.
    """

    def __init__(self, compute: Callable, src: int = 0, device: Optional[torch.device] = None, group=None,
                 compute_stream=None, compact: bool = False):
        self.compute, self.src, self.group = compute, src, group
        self.rank, self.world = _world(group)
        self.device = device if device is not None else torch.device("cpu")
        self.s = _Streams(self.device, compute_stream)
        self.compact = compact
        self.shape = self.dtype = None
        self.k = 0
        self.scattered = {}    # k -> (l, r, event behind the scatter)
        self.computed = {}     # k -> (results tuple, many?, event behind the kernels)
        self.gathering = {}    # k -> (posted, many?, ragged)
        self.gathering2 = {}   # k -> (dense outs, point tensors, counts, batch)       (compact only)
        self.done: List = []   # on src: results in batch order
        self.in_bufs = [[None, None], [None, None], [None, None]]   # receive buffers, 3 deep (scatter k + 2 beside compute k + 1 beside gather k)

    def _post_scatter(self, k, lefts, rights):
        if self.world == 1:
            self.scattered[k] = (lefts.to(self.device), rights.to(self.device), None)
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
        self.scattered[k] = (l, r, landed)

    def _compute(self, k):
        l, r, landed = self.scattered.pop(k)
        self.s.compute_waits(landed)
        self.s.used_by_compute(l, r)
        res = self.compute(l, r)
        many = isinstance(res, (tuple, list))
        self.computed[k] = (tuple(res) if many else (res,), many, self.s.event_after_compute())

    def _post_gather(self, k):
        res, many, ev = self.computed.pop(k)
        ragged = None
        if self.compact:
            if len(res) < 2:
                raise ValueError("IngestPipeline(compact=True): compute must return (..., points, counts)")
            ragged = (res[-2], res[-1], ev)
            res = res[:-2] + (res[-1],)          # the counts travel with the dense results
        if self.world == 1:
            if self.compact:
                self.gathering[k] = ([(t, None) for t in res], many, ragged)
            else:
                self.done.append(res if many else res[0])
            return
        n = self.shape[0]
        with self.s.on_comm():
            self.s.comm_waits(ev)
            self.s.used_by_comm(*res)
            posted = [post_gather(t, n, self.src, self.group) for t in res]
        self.gathering[k] = (posted, many, ragged)

    def _finish_gather(self, k):
        if k not in self.gathering:    # (a single process gathers nothing)
            return
        posted, many, ragged = self.gathering.pop(k)
        with self.s.on_comm():
            for _, b in posted:
                if b is not None:
                    b.wait()
        outs = tuple(o for o, _ in posted)
        if ragged is None:
            if self.rank == self.src:
                self.done.append(outs if many else outs[0])
            return
        # ---- compact: the counts are here (on src: everybody's); now the points, exactly counts[i] per frame
        points, counts, ev = ragged
        if self.s.cuda and ev is not None:
            ev.synchronize()                 # (compute(k) is done; compute(k + 1) was enqueued a step ago)
        mine = [int(c) for c in counts.cpu().tolist()]
        b = _Batch(self.group)
        if self.world == 1:
            self.gathering2[k] = (outs[:-1], [points[i, :c] for i, c in enumerate(mine)], counts, b)
            return
        n = self.shape[0]
        with self.s.on_comm():
            self.s.used_by_comm(points)
            if self.rank != self.src:
                for i, c in enumerate(mine):
                    b.send(points[i, :c].contiguous(), self.src)
                self.gathering2[k] = (None, None, None, b.post())
                return
            if self.s.cuda:
                self.s.comm.synchronize()    # the gathered counts have landed (posted a step ago)
            allc = outs[-1]
            every = [int(c) for c in allc.cpu().tolist()]
            pts: List[Optional[torch.Tensor]] = [None] * n
            for r in range(self.world):
                a, e = shard_range(n, r, self.world)
                for i in range(a, e):
                    if r == self.src:
                        pts[i] = points[i - a, :every[i]]
                    else:
                        pts[i] = torch.empty((every[i], 3), dtype=points.dtype, device=points.device)
                        b.recv(pts[i], r)
            self.gathering2[k] = (outs[:-1], pts, allc, b.post())

    def _finish_points(self, k):
        dense, pts, counts, b = self.gathering2.pop(k)
        with self.s.on_comm():
            b.wait()
        if self.rank == self.src:
            self.done.append(tuple(dense) + (pts, counts))

    def _tick(self, k, lefts, rights, feed: bool):
        """One step of the schedule (the same order on every rank, so that transfers pair up)."""
        if k - 4 in self.gathering2:
            self._finish_points(k - 4)
        if k - 3 in self.gathering:
            self._finish_gather(k - 3)
        if k - 2 in self.computed:
            self._post_gather(k - 2)
        if feed:
            self._post_scatter(k, lefts, rights)
        if k - 1 in self.scattered:
            self._compute(k - 1)

    def step(self, lefts: Optional[torch.Tensor], rights: Optional[torch.Tensor]):
        """Feed batch k: finish gather(k - 3), post gather(k - 2), post scatter(k), enqueue compute(k - 1)."""
        self._tick(self.k, lefts, rights, True)
        self.k += 1

    def drain(self):
        """Finish everything in flight; returns (on `src`) the list of per-batch results, else an empty list.  Raises if
        the engine behind `compute` reports an error for the work it was handed without synchronising (compute.check)."""
        for j in range(5):
            self._tick(self.k + j, None, None, False)
        if self.s.cuda:
            self.s.comm.synchronize()
            if self.s.compute is not None:
                self.s.compute.synchronize()
        check = getattr(self.compute, "check", None)
        if check is not None:
            check()
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
    the call returns (disparity, points [n, H*W, 3] float32, counts [n] int64 on the device): the valid points of every
    frame (main.ipynb:726-737: finite X and disparity > 0) packed to the front of its row (the rest of the row is
    unspecified), computed in stream order without a host round trip; a consumer that only wants the cloud gathers
    counts first and then counts[i] points per frame (gather_compacted, IngestPipeline(compact=True)).

    The returned callable has a `check()` attribute (Engine.check of the engines it used): IngestPipeline.drain calls
    it, so a chained sweep that gave up inside stream-ordered work is reported instead of handing wrong maps on."""
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
            # valid points of every frame packed to the front of its row, counts in device memory: three launches and a
            # one-thread kernel per frame on the engine's stream, no host round trip (sgm_compact_points_device_async)
            # (torch.empty: nothing is enqueued on torch's stream that the engine's stream could race with -- round 3 zero-filled
            #  the rows there; every count is written by the engine, rows past the count are unspecified)
            pts = torch.empty((n, H * W, 3), dtype=torch.float32, device=dev)
            counts = torch.empty((n,), dtype=torch.int64, device=dev)
            if stream is not None:
                pts.record_stream(stream)
                counts.record_stream(stream)
            for i in range(n):
                eng.compact_points_device_async(xyz[i].data_ptr(), dispf[i].data_ptr(), None, H * W, pts[i].data_ptr(), None,
                                                counts[i:].data_ptr())
            if synchronize:
                eng.synchronize()
            return disp, pts, counts
        if synchronize:
            eng.synchronize()
        if Qm is None:
            return disp
        return (disp, dispf, xyz) if want_float else (disp, xyz)

    def check():
        """status of the engines this callable has used, without waiting for them (Engine.check)"""
        for e in own.values():
            e.check()

    compute.check = check
    return compute


def gather_compacted(points: torch.Tensor, counts: torch.Tensor, n_frames: int, dst: int = 0, group=None):
    """Gather of compacted point lists (hip_batch_compute(compact=True)): counts first (8 bytes per frame), then
    exactly counts[i] points of every frame -- for a 4K pair with a third of its pixels valid 33 MB instead of the
    99.5 MB of the dense XYZ image (the synthetic bench pairs are 91 % valid: what is saved is the invalid share).  Rank `dst` returns (list of [counts[i], 3] tensors, counts), the others None."""
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
