"""Chained sweeps (SGM_OPT_SCHEDULE = 2, kernels_sweep.h: k_sweep_chain): the bands of a sweep take the
state of the row above them from the band above -- through HBM, behind a progress word -- instead of
from a boundary pre-pass.  Same arithmetic, so every stage tap must equal the oracle's bit for bit
(upstream's single top-down / bottom-up pass per row: /root/reference/main.ipynb:668),
for any band height, any number of workgroups in flight (1 = the bands strictly one after the other),
both modes, full and partial wavefronts, D <= 128 / 256 / 512, and launch after launch on one engine
(ticket and progress words are reset by the engine before every launch)."""
import numpy as np
import pytest

import parity_util as U
from oracle import oracle as O
from stereo_reconstruction_cv_amd import _lib, synth
from stereo_reconstruction_cv_amd.stereo import Engine

pytestmark = pytest.mark.gpu

SHAPES = [
    # H, W, D, bs, mode, rows per band
    (61, 300, 128, 5, 0, 3), (47, 420, 256, 7, 1, 2), (90, 200, 80, 3, 1, 4), (33, 1100, 512, 3, 1, 1),
    (130, 228, 128, 5, 1, 9), (58, 500, 192, 5, 0, 11), (75, 400, 256, 7, 1, 5), (29, 640, 160, 5, 0, 7),
    (40, 700, 384, 3, 1, 3), (70, 300, 64, 5, 0, 0), (55, 260, 48, 3, 1, 0),
]


@pytest.mark.parametrize("wgs", [0, 1, 2, 5, 64])
@pytest.mark.parametrize("H,W,D,bs,mode,rows", SHAPES)
def test_chained_sweeps_bit_exact(H, W, D, bs, mode, rows, wgs):
    l, r, _ = synth.make_pair(H, W, D, 4000 + H + D)
    p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
    rep, t, h = U.compare_stages(l, r, p, schedule=2, sweep_rows=rows, chain_wgs=wgs)
    assert t["headroom_ok"]
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, f"wgs={wgs}: " + "\n".join(bad)


def test_back_to_back_frames_on_one_engine():
    """ticket / progress words are per launch: a second and third frame (different images, then a
    different shape) on the same engine must not see anything of the first."""
    p = U.params(256, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, 4)
    for (H, W, seed) in ((50, 420, 1), (50, 420, 2), (37, 500, 3), (50, 420, 1)):
        l, r, _ = synth.make_pair(H, W, 256, seed)
        want = O.sgbm_compute(l, r, **p)
        got = eng.compute_host(l, r)
        assert np.array_equal(got, want), (H, W, seed, int((got != want).sum()))


def test_small_disparity_ranges_keep_their_own_schedule():
    """D <= 32 runs the lane-grouped kernels whatever the schedule option says (D = 48 .. 64 is chained in throughput
    mode: SHAPES below holds D = 64 and 48 frames with the engine's own band height)."""
    l, r, _ = synth.make_pair(60, 300, 32, 9)
    p = U.params(32, 5, 0, 1, speckleWindowSize=30, speckleRange=2)
    rep, t, h = U.compare_stages(l, r, p, schedule=2)
    assert not [k for k, n in rep.items() if n]


@pytest.mark.parametrize("H,W,D,bs,mode,rows,N", [(47, 420, 256, 7, 1, 2, 3), (61, 300, 128, 5, 0, 3, 5), (40, 520, 256, 5, 0, 4, 2),
                                                   (33, 300, 96, 3, 1, 1, 18), (52, 700, 512, 3, 1, 5, 4)])
def test_batch_entry_shares_one_chained_launch(H, W, D, bs, mode, rows, N):
    """sgm_pipeline_batch_device: N resident pairs, one chained sweep launch per pass for a group of up to 16 pairs
    (18 pairs = two groups), every pair's cost stage / epilogue on an engine of its own.  Each map, float map and XYZ image
    must equal the oracle's for its own pair -- frames of a group share nothing but the ticket counter."""
    import torch
    p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    pairs = [synth.make_pair(H, W, D, 700 + i)[:2] for i in range(N)]
    dev = torch.device("cuda", 0)
    dl = [torch.from_numpy(a).to(dev) for a, _ in pairs]
    dr = [torch.from_numpy(b).to(dev) for _, b in pairs]
    dd = [torch.empty((H, W), dtype=torch.int16, device=dev) for _ in range(N)]
    df = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(N)]
    dx = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(N)]
    torch.cuda.synchronize()
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, rows)
    ptr = lambda ts: [t.data_ptr() for t in ts]
    for rep in range(2):      # twice: the second call reuses the group's engines and control words
        for t in dd:
            t.fill_(-7)
        torch.cuda.synchronize()
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, Q, ptr(dd), ptr(df), ptr(dx))
        eng.synchronize()
        for i, (a, b) in enumerate(pairs):
            want = O.sgbm_compute(a, b, **p)
            got = dd[i].cpu().numpy()
            assert np.array_equal(got, want), (rep, i, int((got != want).sum()))
            wf = O.disp_to_float(want)
            assert np.array_equal(df[i].cpu().numpy().view(np.uint32), wf.view(np.uint32)), (rep, i)
            ref = O.reproject(wf, Q)
            x = dx[i].cpu().numpy()
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(x), fin) and np.array_equal(x[fin], ref[fin]), (rep, i)


def test_sharded_compute_returns_compacted_point_lists():
    """dist.hip_batch_compute(compact=True): what a rank hands to dist.gather_compacted instead of the dense XYZ image --
    the valid points of every frame (main.ipynb:726-737: finite X and disparity > 0) packed to the front of its row, in
    row-major order, with their count; equal to numpy boolean indexing on the oracle's XYZ."""
    import torch
    from stereo_reconstruction_cv_amd import dist as D_
    H, W, D, N = 60, 420, 128, 3
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    pairs = [synth.make_pair(H, W, D, 900 + i)[:2] for i in range(N)]
    L = torch.from_numpy(np.stack([a for a, _ in pairs])).cuda()
    R = torch.from_numpy(np.stack([b for _, b in pairs])).cuda()
    for schedule in (1, 2):
        disp, pts, counts = D_.hip_batch_compute(p, Q, schedule=schedule, compact=True)(L, R)
        assert pts.shape == (N, H * W, 3) and counts.shape == (N,)
        for i, (a, b) in enumerate(pairs):
            want = O.sgbm_compute(a, b, **p)
            assert np.array_equal(disp[i].cpu().numpy(), want)
            f = O.disp_to_float(want)
            xyz = O.reproject(f, Q)
            mask = O.valid_mask(xyz, f)
            n = int(counts[i])
            assert n == int(mask.sum()) and n > 0
            assert np.array_equal(pts[i, :n].cpu().numpy(), xyz[mask])


def test_ingest_pipeline_orders_the_engine_stream_with_events():
    """dist.IngestPipeline on the GPU with one process: batches are handed to an engine that works on a torch stream of
    its own and returns without synchronising (hip_batch_compute(stream=..., synchronize=False)); the pipeline orders
    transfers and kernels with events only and drain() returns every batch's results, in order."""
    import torch
    from stereo_reconstruction_cv_amd import dist as D_
    H, W, D, N = 50, 420, 128, 2
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    dev = torch.device("cuda", 0)
    cs = torch.cuda.Stream(dev)
    compute = D_.hip_batch_compute(p, Q, schedule=2, stream=cs, synchronize=False)
    batches, wants = [], []
    for b in range(4):
        pairs = [synth.make_pair(H, W, D, 1300 + 10 * b + i)[:2] for i in range(N)]
        batches.append((torch.from_numpy(np.stack([a for a, _ in pairs])), torch.from_numpy(np.stack([c for _, c in pairs]))))
        wants.append([O.sgbm_compute(a, c, **p) for a, c in pairs])
    pipe = D_.IngestPipeline(compute, src=0, device=dev, compute_stream=cs)
    for l, r in batches:
        pipe.step(l, r)
    res = pipe.drain()
    assert len(res) == 4
    for b, (disp, xyz) in enumerate(res):
        for i in range(N):
            assert np.array_equal(disp[i].cpu().numpy(), wants[b][i]), (b, i)
            ref = O.reproject(O.disp_to_float(wants[b][i]), Q)
            fin = np.isfinite(ref)
            assert np.array_equal(xyz[i].cpu().numpy()[fin], ref[fin])


def test_chained_batch_beside_other_work_on_the_gpu():
    """The hand-off between bands (write-through record + progress word, polled by the band below) under UNEVEN load:
    while a chained batch of six medium frames runs on one engine, a second engine keeps the GPU busy with
    latency-mode frames on a stream of its own (pre-pass, sweeps, cost kernels competing for CUs, L2 and HBM), twice
    over.  Workgroups of the chained launch then start late, out of order and on whatever CUs are free -- the ticket
    order makes that harmless -- and every map of both engines must still equal the oracle's."""
    import torch
    H, W, D = 260, 1500, 128
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    pairs = [synth.make_pair(H, W, D, 2100 + i)[:2] for i in range(3)]
    wants = [O.sgbm_compute(a, b, **p) for a, b in pairs]
    dev = torch.device("cuda", 0)
    n = 6
    dl = [torch.from_numpy(pairs[i % 3][0]).to(dev) for i in range(n)]
    dr = [torch.from_numpy(pairs[i % 3][1]).to(dev) for i in range(n)]
    dd = [torch.empty((H, W), dtype=torch.int16, device=dev) for _ in range(n)]
    other = [torch.empty((H, W), dtype=torch.int16, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    a = Engine(p)
    a.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    b = Engine(p)                      # latency mode, its own stream
    ptr = lambda ts: [t.data_ptr() for t in ts]
    for rep in range(2):
        for t in dd + other:
            t.fill_(-9)
        torch.cuda.synchronize()
        for k in range(4):
            b.compute_device(dl[k % 3].data_ptr(), dr[k % 3].data_ptr(), H, W, W, other[k].data_ptr())
        a.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
        for k in range(4, 8):
            b.compute_device(dl[k % 3].data_ptr(), dr[k % 3].data_ptr(), H, W, W, other[k].data_ptr())
        a.synchronize()
        b.synchronize()
        for i in range(n):
            got = dd[i].cpu().numpy()
            assert np.array_equal(got, wants[i % 3]), (rep, i, int((got != wants[i % 3]).sum()))
        for k in range(8):
            assert np.array_equal(other[k].cpu().numpy(), wants[k % 3]), (rep, k)
