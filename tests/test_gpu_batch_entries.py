"""The batch entries of the C ABI beyond the happy path (needs an MI355X): groups smaller than the batch (group size from
SGM_OPT_GROUP_MAX -- what running out of device memory does to a large batch), error returns that leave nothing in
flight and the engine usable, the headroom record over every pair of a call, sgm_check / sgm_trim, and the host entry
in throughput mode with two groups in flight (uploads / downloads beside the kernels).  Every map is compared with the
oracle's for its own pair (upstream's compute per pair: /root/reference/main.ipynb:668, 780-797)."""
import ctypes as C

import numpy as np
import pytest

import parity_util as U
from oracle import oracle as O
from stereo_reconstruction_cv_amd import _lib, synth
from stereo_reconstruction_cv_amd import stereo as cv
from stereo_reconstruction_cv_amd.stereo import Engine

pytestmark = pytest.mark.gpu


def _resident(pairs, H, W, with_q):
    import torch
    dev = torch.device("cuda", 0)
    n = len(pairs)
    dl = [torch.from_numpy(a).to(dev) for a, _ in pairs]
    dr = [torch.from_numpy(b).to(dev) for _, b in pairs]
    dd = [torch.full((H, W), -7, dtype=torch.int16, device=dev) for _ in range(n)]
    df = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(n)] if with_q else None
    dx = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(n)] if with_q else None
    torch.cuda.synchronize()
    return dl, dr, dd, df, dx


ptr = lambda ts: [t.data_ptr() for t in ts] if ts is not None else None


@pytest.mark.parametrize("H,W,D,bs,mode,rows,N,gmax", [(40, 300, 128, 5, 1, 3, 8, 3), (33, 420, 256, 7, 0, 2, 7, 2),
                                                        (36, 300, 96, 3, 1, 4, 5, 4), (30, 260, 64, 5, 0, 0, 9, 4)])
def test_batches_larger_than_a_group(H, W, D, bs, mode, rows, N, gmax):
    """N pairs with at most gmax per chained launch: the batch is cut into groups of equal size (8 with 3 -> 3 + 3 + 2,
    7 with 2 -> 2 + 2 + 2 + 1: the last pair alone takes the single-pair entry), the engines of the first group are
    reused by the next, and every pair still gets its own result; the headroom record covers all of them."""
    p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    pairs = [synth.make_pair(H, W, D, 5100 + i)[:2] for i in range(N)]
    oracle = [O.sgbm_compute(a, b, taps=True, **p) for a, b in pairs]
    dl, dr, dd, df, dx = _resident(pairs, H, W, True)
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, rows)
    eng.set_option(_lib.SGM_OPT_GROUP_MAX, gmax)
    for rep in range(2):
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, Q, ptr(dd), ptr(df), ptr(dx))
        eng.synchronize()
        for i in range(N):
            want = oracle[i][0]
            got = dd[i].cpu().numpy()
            assert np.array_equal(got, want), (rep, i, int((got != want).sum()))
            ref = O.reproject(O.disp_to_float(want), Q)
            fin = np.isfinite(ref)
            x = dx[i].cpu().numpy()
            assert np.array_equal(np.isfinite(x), fin) and np.array_equal(x[fin], ref[fin]), (rep, i)
        hr = eng.headroom()
        assert hr == dict(ok=all(t["headroom_ok"] for _, t in oracle), max_cost_plus_p2=max(t["max_cost_plus_p2"] for _, t in oracle),
                          max_delta=max(t["max_delta"] for _, t in oracle)), hr


def test_headroom_of_a_batch_sees_an_overflowing_pair_at_any_index():
    """One pair of a batch leaves the int16 regime (constant 0 against constant 255 at blockSize = 11 with P2 = 24500: C + P2 =
    32 816) among low-contrast pairs that stay inside; it sits at index 2 of 4 -- on an internal engine of the group.  Round 3
    reported pair 0 only."""
    H, W, D = 40, 300, 64
    p = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=100, P2=24500, mode=1, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=10, speckleWindowSize=0, speckleRange=0)
    soft = lambda a: (128 + (a.astype(int) - 128) // 8).astype(np.uint8)
    pairs = [tuple(soft(x) for x in synth.make_pair(H, W, D, 5200 + i)[:2]) for i in range(4)]
    pairs[2] = (np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8))
    oracle = [O.sgbm_compute(a, b, taps=True, **p)[1] for a, b in pairs]
    assert not oracle[2]["headroom_ok"] and all(oracle[i]["headroom_ok"] for i in (0, 1, 3)), [t["max_cost_plus_p2"] for t in oracle]
    dl, dr, dd, _, _ = _resident(pairs, H, W, False)
    for schedule, gmax in ((2, 0), (2, 2), (1, 0)):      # one group; two groups; pair after pair (latency mode)
        eng = Engine(p)
        eng.set_option(_lib.SGM_OPT_SCHEDULE, schedule)
        eng.set_option(_lib.SGM_OPT_GROUP_MAX, gmax)
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
        hr = eng.headroom()
        assert hr["ok"] is False, (schedule, gmax, hr)
        assert hr["max_cost_plus_p2"] == max(t["max_cost_plus_p2"] for t in oracle), (schedule, gmax, hr)
        assert hr["max_delta"] == max(t["max_delta"] for t in oracle), (schedule, gmax, hr)
        # ... and a later call without that pair is inside the regime again
        eng.pipeline_batch_device(ptr(dl[:2]), ptr(dr[:2]), H, W, W, None, ptr(dd[:2]))
        assert eng.headroom()["ok"] is True


def test_error_returns_leave_the_engine_usable():
    """A batch call that fails (a null pointer in the middle of the batch, a row stride smaller than a row, a missing Q)
    returns an error with nothing in flight; the next good call on the same engine is bit-exact."""
    H, W, D, N = 44, 360, 128, 5
    p = U.params(D, 5, 0, 1, speckleWindowSize=30, speckleRange=2)
    pairs = [synth.make_pair(H, W, D, 5300 + i)[:2] for i in range(N)]
    wants = [O.sgbm_compute(a, b, **p) for a, b in pairs]
    dl, dr, dd, df, dx = _resident(pairs, H, W, True)
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, 4)
    eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))      # a good call first: the group exists, streams have history
    bad = ptr(dl)
    bad[3] = 0
    with pytest.raises(cv.error, match="pair 3"):
        eng.pipeline_batch_device(bad, ptr(dr), H, W, W, None, ptr(dd))
    with pytest.raises(cv.error, match="stride"):
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W - 1, None, ptr(dd))
    with pytest.raises(cv.error, match="Q is null"):
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd), ptr(df), ptr(dx))
    with pytest.raises(cv.error):
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, 1, 1, None, ptr(dd))   # W < 2
    eng.check()                                                                  # nothing pending, nothing wrong
    for t in dd:
        t.fill_(-7)
    eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
    eng.synchronize()
    for i in range(N):
        assert np.array_equal(dd[i].cpu().numpy(), wants[i]), i
    # a shape change between two calls of one engine (every internal engine regrows its buffers)
    H2, W2 = 30, 500
    pairs2 = [synth.make_pair(H2, W2, D, 5400 + i)[:2] for i in range(3)]
    dl2, dr2, dd2, _, _ = _resident(pairs2, H2, W2, False)
    eng.pipeline_batch_device(ptr(dl2), ptr(dr2), H2, W2, W2, None, ptr(dd2))
    eng.synchronize()
    for i, (a, b) in enumerate(pairs2):
        assert np.array_equal(dd2[i].cpu().numpy(), O.sgbm_compute(a, b, **p)), i


def test_trim_gives_the_group_memory_back_and_the_engine_still_works():
    import torch
    H, W, D, N = 200, 1200, 128, 6
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    pairs = [synth.make_pair(H, W, D, 5500 + i)[:2] for i in range(2)]
    wants = [O.sgbm_compute(a, b, **p) for a, b in pairs]
    dl, dr, dd, _, _ = _resident([pairs[i % 2] for i in range(N)], H, W, False)
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    free0 = torch.cuda.mem_get_info()[0]
    eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
    eng.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    eng.trim()
    free2 = torch.cuda.mem_get_info()[0]
    vol = 2 * H * (W - D) * D
    assert free0 - free1 > N * 2 * vol * 0.9, (free0, free1)          # six engines' C and S volumes
    assert free2 - free1 > (N - 1) * 2 * vol * 0.9, (free1, free2)    # five of them came back
    for t in dd:
        t.fill_(-7)
    eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
    eng.synchronize()
    for i in range(N):
        assert np.array_equal(dd[i].cpu().numpy(), wants[i % 2]), i


def test_group_shrinks_to_what_device_memory_holds():
    """Device memory is squeezed (torch tensors that are never touched) until only a few pairs of a batch of ten fit beside the
    reserve the engine keeps free (4 GiB / 5 %): the call must not fail -- and must not die of a kernel launch that finds no
    memory -- but cut the batch into smaller groups, every map still its own pair's; with the memory back, the same engine
    takes the batch in one group again."""
    import torch
    from stereo_reconstruction_cv_amd import stereo as cv_
    cv_.clear_engine_cache()
    H, W, D, N = 540, 1920, 128, 10
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    pairs = [synth.make_pair(H, W, D, 5800 + i)[:2] for i in range(2)]
    wants = [O.sgbm_compute(a, b, **p) for a, b in pairs]
    dl, dr, dd, _, _ = _resident([pairs[i % 2] for i in range(N)], H, W, False)
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    torch.cuda.empty_cache()
    free_b, total = torch.cuda.mem_get_info()
    per_pair = 2 * 2 * H * (W - D) * D + 3 * (H // 12 + 1) * (W - D) * D * 2      # C + S + hand-off record
    reserve = max(4 << 30, total // 20)
    keep = reserve + 4 * per_pair                                                    # room for about four pairs beyond the reserve
    hog = torch.empty((max(free_b - keep, 0),), dtype=torch.uint8, device="cuda") if free_b > keep else None
    try:
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
        eng.synchronize()
        for i in range(N):
            assert np.array_equal(dd[i].cpu().numpy(), wants[i % 2]), i
    finally:
        del hog
        torch.cuda.empty_cache()
    for t in dd:
        t.fill_(-7)
    eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
    eng.synchronize()
    for i in range(N):
        assert np.array_equal(dd[i].cpu().numpy(), wants[i % 2]), i


@pytest.mark.parametrize("N,gmax,with_q", [(8, 3, True), (7, 4, False), (5, 0, True), (3, 2, False)])
def test_host_entry_in_throughput_mode_with_groups_in_flight(N, gmax, with_q):
    """sgm_compute_batch with SGM_OPT_SCHEDULE = 2: host arrays in, host arrays out (cv2.imread -> compute,
    /root/reference/main.ipynb:362-363, 668), N no multiple of the group size; uploads of group g + 1 and downloads of
    group g - 1 run on copy streams beside the kernels of group g, each pair's cost stage waits for its own images only."""
    H, W, D = 48, 420, 128
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    pairs = [synth.make_pair(H, W, D, 5600 + i)[:2] for i in range(N)]
    L = np.stack([a for a, _ in pairs])
    R = np.stack([b for _, b in pairs])
    eng = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, 5)
    eng.set_option(_lib.SGM_OPT_GROUP_MAX, gmax)
    res = None
    for rep in range(2):
        if res is not None:      # the second call fills the first call's arrays again (out=)
            for a in (res if with_q else (res,)):
                a[...] = 0
        res = eng.compute_batch_host(L, R, Q if with_q else None, out=res)
        disps, xyz = res if with_q else (res, None)
        for i, (a, b) in enumerate(pairs):
            want = O.sgbm_compute(a, b, **p)
            assert np.array_equal(disps[i], want), (rep, i, int((disps[i] != want).sum()))
            if with_q:
                ref = O.reproject(O.disp_to_float(want), Q)
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(xyz[i]), fin) and np.array_equal(xyz[i][fin], ref[fin]), (rep, i)
    assert eng.headroom()["ok"]


def test_ingest_pipeline_survives_the_allocator_reusing_blocks_between_steps():
    """dist.IngestPipeline on the GPU, one process, the engine on a torch stream of its own: between steps the test drops
    every reference it can and allocates and fills blocks of the very sizes the pipeline uses, on torch's current stream --
    if a result or shard tensor went back to the caching allocator while a transfer or a kernel of another stream was
    still pending on it (round 3: the gather of rank 0's own shard ran on the communication stream without the tensor
    being announced to it), that memory would be overwritten here.  Ordering must hold by events and record_stream."""
    import torch
    from stereo_reconstruction_cv_amd import dist as D_
    H, W, D, N = 60, 420, 128, 3
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    Q = synth.default_Q(W)
    dev = torch.device("cuda", 0)
    cs = torch.cuda.Stream(dev)
    batches, wants = [], []
    for b in range(5):
        pairs = [synth.make_pair(H, W, D, 5700 + 10 * b + i)[:2] for i in range(N)]
        batches.append((torch.from_numpy(np.stack([a for a, _ in pairs])), torch.from_numpy(np.stack([c for _, c in pairs]))))
        wants.append([O.sgbm_compute(a, c, **p) for a, c in pairs])
    for compact in (False, True):
        compute = D_.hip_batch_compute(p, Q, schedule=2, stream=cs, synchronize=False, compact=compact)
        pipe = D_.IngestPipeline(compute, src=0, device=dev, compute_stream=cs, compact=compact)
        for l, r in batches:
            pipe.step(l, r)
            for shape, dt in (((N, H, W), torch.int16), ((N, H, W, 3), torch.float32), ((N, H, W), torch.uint8), ((N, H * W, 3), torch.float32)):
                junk = [torch.empty(shape, dtype=dt, device=dev) for _ in range(3)]
                for j in junk:
                    j.fill_(77)
                del junk
        res = pipe.drain()
        assert len(res) == 5
        for b, out in enumerate(res):
            disp = out[0]
            for i in range(N):
                assert np.array_equal(disp[i].cpu().numpy(), wants[b][i]), (compact, b, i)
                f = O.disp_to_float(wants[b][i])
                ref = O.reproject(f, Q)
                if compact:
                    pts, counts = out[1], out[2]
                    mask = O.valid_mask(ref, f)
                    assert int(counts[i]) == int(mask.sum()) and np.array_equal(pts[i].cpu().numpy(), ref[mask]), (b, i)
                else:
                    fin = np.isfinite(ref)
                    assert np.array_equal(out[1][i].cpu().numpy()[fin], ref[fin]), (b, i)
