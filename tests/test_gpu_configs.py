"""Every BASELINE.json config at FULL size, with the benchmark's own parameters and entry points,
bit-exact against the CPU oracle (needs an MI355X; the oracle side costs about a minute in all).

Parameters follow the notebook call (/root/reference/main.ipynb:655-666): P1 = 8*3*bs^2,
P2 = 32*3*bs^2, disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100,
speckleRange=32; inputs are the bench's synthetic pairs (synth.make_pair(seed=1234 + i),
SURVEY.md 8d).  Entry points are the ones bench.py times: sgm_pipeline_device for the single
frames, sgm_compute_batch and dist.hip_batch_compute for the batch config.

Parity statement: bit-exact vs. a restatement of OpenCV 4.11 MODE_SGBM / MODE_HH inside the int16
no-overflow regime (the oracle is unpinned against cv2); XYZ: identical non-finite
masks and 1e-4 relative on finite values (north_star's tolerance; the kernel is in fact bit-exact).
"""
import functools

import numpy as np
import pytest

from oracle import oracle as O
from stereo_reconstruction_cv_amd import _lib, synth

pytestmark = pytest.mark.gpu

NB = dict(disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32)


def nb_params(D, bs, mode):
    return dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs, mode=mode, **NB)


@functools.lru_cache(maxsize=2)
def pair(H, W, D, seed):
    l, r, _ = synth.make_pair(H, W, D, seed)
    return l, r


def oracle_frame(l, r, p, Q=None):
    want, t = O.sgbm_compute(l, r, taps="light", **p)
    assert t["headroom_ok"], "synthetic input left the int16 no-overflow regime"
    if Q is None:
        return want, None, None
    f = O.disp_to_float(want)
    return want, f, O.reproject(f, Q)


@functools.lru_cache(maxsize=3)
def oracle_4k(D, bs, mode, seed, with_q):
    """oracle results of one 4K bench pair, kept for the tests that run the same pair through several entry points"""
    H, W = 2160, 3840
    l, r, _ = synth.make_pair(H, W, D, seed)
    return oracle_frame(l, r, nb_params(D, bs, mode), synth.default_Q(W) if with_q else None)


def check_xyz(got, ref):
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin), "non-finite masks differ"
    # tolerance from BASELINE.json north_star: 1e-4 relative
    assert np.allclose(got[fin], ref[fin], rtol=1e-4, atol=0.0)
    assert np.array_equal(got[fin], ref[fin])   # and in fact bit-exact


def run_pipeline_device(l, r, p, Q):
    """The call bench.py times: device-resident inputs -> sgm_pipeline_device."""
    import torch
    import stereo_reconstruction_cv_amd as cv
    H, W = l.shape
    dev = torch.device("cuda", 0)
    dl, dr = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    dd = torch.empty((H, W), dtype=torch.int16, device=dev)
    df = torch.empty((H, W), dtype=torch.float32, device=dev) if Q is not None else None
    dx = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if Q is not None else None
    torch.cuda.synchronize(dev)
    eng = cv.Engine(p, device=0)
    eng.set_option(_lib.SGM_OPT_PROFILE, 1)   # as the bench does
    for _ in range(2):   # twice: buffer reuse across frames is part of what the bench runs
        eng.pipeline_device(dl.data_ptr(), dr.data_ptr(), H, W, W, Q, dd.data_ptr(),
                            df.data_ptr() if Q is not None else None, dx.data_ptr() if Q is not None else None)
        eng.synchronize()
    names = [n for n, _, _ in eng.stage_times()]
    out = dd.cpu().numpy(), (df.cpu().numpy() if Q is not None else None), (dx.cpu().numpy() if Q is not None else None)
    del eng
    return out + (names,)


def test_c1_720p_d64_host_entry():
    """configs[0]: single 1280x720 pair, D=64, blockSize=5, through the cv2-shaped host call."""
    import stereo_reconstruction_cv_amd as cv
    H, W, D, bs = 720, 1280, 64, 5
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, 0)
    want, _, _ = oracle_frame(l, r, p)
    got = cv.StereoSGBM_create(**p).compute(l, r)
    assert got.dtype == np.int16 and got.shape == (H, W)
    assert int((got != want).sum()) == 0
    assert (got >= 0).mean() > 0.5
    got2, _, _, _ = run_pipeline_device(l, r, p, None)
    assert int((got2 != want).sum()) == 0


def test_c2_4k_d128_pipeline_device():
    """configs[1]: single 3840x2160 pair, D=128, blockSize=7, 5 paths."""
    H, W, D, bs = 2160, 3840, 128, 7
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, 0)
    want, _, _ = oracle_frame(l, r, p)
    got, _, _, names = run_pipeline_device(l, r, p, None)
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} of {got.size} differ"
    assert "sweep_dn" in names      # the fused-sweep schedule ran (not a fallback)


def test_c3_c5_4k_d256_hh_pipeline_device_with_xyz():
    """configs[2] + configs[4] = bench.py's default workload c3c5: 3840x2160, D=256, MODE_HH (8
    paths) + LR + sub-pixel + median + speckle + reprojectImageTo3D through sgm_pipeline_device."""
    H, W, D, bs = 2160, 3840, 256, 7
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, 1)
    Q = synth.default_Q(W)
    want, wf, wxyz = oracle_4k(D, bs, 1, 1234, True)
    got, gf, gxyz, names = run_pipeline_device(l, r, p, Q)
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} of {got.size} differ"
    assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32))      # incl. the sign of zero
    check_xyz(gxyz, wxyz)
    assert {"sweep_dn", "sweep_up", "wta", "float_xyz"} <= set(names)


def run_batch_device(frames, p, Q, schedule=2, repeat_to=0):
    """sgm_pipeline_batch_device on resident pairs, as `bench.py --batch` calls it (profiling on, two rounds).
    repeat_to = N > len(frames): the batch holds N pairs, pair i a copy of frames[i % len(frames)] in buffers of its own (the
    bench does the same: a handful of different images, every pair computed); returned are the results of the first
    len(frames) pairs and, for the others, whether their three outputs equal their source pair's bit for bit."""
    import torch
    import stereo_reconstruction_cv_amd as cv
    H, W = frames[0][0].shape
    nu = len(frames)
    n = max(repeat_to, nu)
    dev = torch.device("cuda", 0)
    dl = [torch.from_numpy(frames[i % nu][0]).to(dev) for i in range(n)]
    dr = [torch.from_numpy(frames[i % nu][1]).to(dev) for i in range(n)]
    dd = [torch.empty((H, W), dtype=torch.int16, device=dev) for _ in range(n)]
    df = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(n)] if Q is not None else None
    dx = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(n)] if Q is not None else None
    torch.cuda.synchronize(dev)
    eng = cv.Engine(p, device=0)
    eng.set_option(_lib.SGM_OPT_PROFILE, 1)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, schedule)
    ptr = lambda ts: [t.data_ptr() for t in ts] if ts is not None else None
    for _ in range(2):
        for t in dd:
            t.fill_(-7)
        torch.cuda.synchronize(dev)
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, Q, ptr(dd), ptr(df), ptr(dx))
        eng.synchronize()
    names = [nm for nm, _, _ in eng.stage_times()]
    assert eng.headroom()["ok"]          # (the record of a batch call covers every pair of it)
    out = [(dd[i].cpu().numpy(), df[i].cpu().numpy() if Q is not None else None, dx[i].cpu().numpy() if Q is not None else None)
           for i in range(nu)]
    # bitwise comparison of the repeats (int32 views: NaN payloads and the sign of zero count)
    same = [bool(torch.equal(dd[i], dd[i % nu]) and (Q is None or (torch.equal(df[i].view(torch.int32), df[i % nu].view(torch.int32)) and
                                                                  torch.equal(dx[i].view(torch.int32), dx[i % nu].view(torch.int32)))))
            for i in range(nu, n)]
    del eng
    return out, names, same


def bench_workload(name):
    """(H, W, D, blockSize, mode, pairs per step, reproject) of a bench.py workload: the tests below run the bench's own batch"""
    import bench
    H, W, D, bs, mode, ppg, with_xyz, schedule, batch, _ = bench.WORKLOADS[name]
    assert schedule == 2 and batch
    return H, W, D, bs, mode, ppg, with_xyz


def test_c3_c5_4k_d256_hh_throughput_mode_batch():
    """bench.py's DEFAULT workload, as the bench runs it: 3840x2160, D=256, MODE_HH + reprojection, the default's own
    number of pairs per step (17) through ONE call of sgm_pipeline_batch_device with chained sweeps (SGM_OPT_SCHEDULE 2:
    no pre-pass, one sweep launch per pass for the whole group).  Two different images fill the batch alternately: the
    first two pairs' maps, float maps and XYZ images equal the oracle's, every other pair equals its source pair bit for
    bit -- computed in a slot of its own, on an engine of its own."""
    import bench
    H, W, D, bs, mode, ppg, with_xyz = bench_workload(bench.DEFAULT_WORKLOAD)
    assert (H, W, D, bs, mode, with_xyz) == (2160, 3840, 256, 7, 1, True) and ppg >= 12
    Q = synth.default_Q(W)
    seeds = (1234, 1235)
    frames = [synth.make_pair(H, W, D, s)[:2] for s in seeds]
    outs, names, same = run_batch_device(frames, nb_params(D, bs, mode), Q, repeat_to=ppg)
    assert {"chain_dn", "chain_up", "wta", "float_xyz"} <= set(names) and "prepass_dn" not in names
    for i, s in enumerate(seeds):
        want, wf, wxyz = oracle_4k(D, bs, 1, s, True)
        got, gf, gxyz = outs[i]
        assert int((got != want).sum()) == 0, f"pair {i}: {int((got != want).sum())} of {got.size} differ"
        assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32))
        check_xyz(gxyz, wxyz)
    assert len(same) == ppg - 2 and all(same), [i + 2 for i, ok in enumerate(same) if not ok]


@pytest.mark.parametrize("workload", ["c4t", "c1t"])
def test_bench_batches_of_smaller_frames_at_their_own_size(workload):
    """The other throughput-mode workloads of bench.py with their own pair counts: c4t = 32 pairs 1920x1080 D=128 (half of
    BASELINE configs[3] on one GPU; the fifth path beside the chained sweep into a volume of its own), c1t = 64 pairs
    1280x720 D=64 (k_sweep_chain<1, partial>: half the lanes idle).  Two different images alternate."""
    H, W, D, bs, mode, ppg, with_xyz = bench_workload(workload)
    p = nb_params(D, bs, mode)
    frames = [synth.make_pair(H, W, D, s)[:2] for s in (1234, 1235)]
    outs, names, same = run_batch_device(frames, p, None, repeat_to=ppg)
    assert "chain_dn" in names
    for i, (a, b) in enumerate(frames):
        want, _, _ = oracle_frame(a, b, p)
        assert int((outs[i][0] != want).sum()) == 0, f"pair {i}: {int((outs[i][0] != want).sum())} differ"
    assert len(same) == ppg - 2 and all(same), [i + 2 for i, ok in enumerate(same) if not ok]


@pytest.mark.parametrize("D,mode", [(256, 0), (128, 0)])
def test_4k_5path_chained_single_pair_and_batch(D, mode):
    """MODE_SGBM with chained sweeps at full size: D = 256 (fifth path fused with the winner-take-all) through
    sgm_pipeline_device, D = 128 (fifth path beside the sweep into a volume of its own) as a group of two."""
    import torch
    import stereo_reconstruction_cv_amd as cv
    H, W, bs = 2160, 3840, 7
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, mode)
    want, _, _ = oracle_frame(l, r, p)
    if D == 256:
        dev = torch.device("cuda", 0)
        dl, dr = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
        dd = torch.empty((H, W), dtype=torch.int16, device=dev)
        eng = cv.Engine(p, device=0)
        eng.set_option(_lib.SGM_OPT_PROFILE, 1)
        eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
        for _ in range(2):
            eng.pipeline_device(dl.data_ptr(), dr.data_ptr(), H, W, W, None, dd.data_ptr(), None, None)
            eng.synchronize()
        assert "chain_dn" in [n for n, _, _ in eng.stage_times()]
        got = dd.cpu().numpy()
        assert int((got != want).sum()) == 0
    else:
        outs, names, _ = run_batch_device([(l, r), (l, r)], p, None)
        assert "chain_dn" in names and "path_W" in names
        for got, _, _ in outs:
            assert int((got != want).sum()) == 0


def test_c5_4k_d256_5path_pipeline_device_with_xyz():
    """configs[4] as the notebook would run it (default mode = 5 paths) + XYZ."""
    H, W, D, bs = 2160, 3840, 256, 7
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, 0)
    Q = synth.default_Q(W)
    want, wf, wxyz = oracle_frame(l, r, p, Q)
    got, gf, gxyz, _ = run_pipeline_device(l, r, p, Q)
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} of {got.size} differ"
    assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32))
    check_xyz(gxyz, wxyz)


def test_c4_batch_1080p_d128_both_batch_entries():
    """configs[3], one GPU's share reduced to 5 frames (not a multiple of the three engines the batch
    path keeps in flight: they end differently): sgm_compute_batch (host pointers, page-locked staging, XYZ) and dist.hip_batch_compute
    (device tensors, XYZ) -- the entry the sharded multi-GPU path calls on every rank."""
    import torch
    import stereo_reconstruction_cv_amd as cv
    from stereo_reconstruction_cv_amd import dist as D_
    H, W, D, bs, N = 1080, 1920, 128, 7, 5
    p = nb_params(D, bs, 0)
    Q = synth.default_Q(W)
    frames = [synth.make_pair(H, W, D, 1234 + i)[:2] for i in range(N)]
    L = np.stack([a for a, _ in frames])
    R = np.stack([b for _, b in frames])
    wants = [oracle_frame(L[i], R[i], p, Q) for i in range(N)]
    eng = cv.get_engine(p)
    disps, xyz = eng.compute_batch_host(L, R, Q)
    for i in range(N):
        assert int((disps[i] != wants[i][0]).sum()) == 0, i
        check_xyz(xyz[i], wants[i][2])
    compute = D_.hip_batch_compute(p, Q, want_float=True)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    d2, f2, x2 = compute(dl, dr)
    assert d2.is_cuda and d2.dtype == torch.int16 and x2.shape == (N, H, W, 3)
    d2, f2, x2 = d2.cpu().numpy(), f2.cpu().numpy(), x2.cpu().numpy()
    for i in range(N):
        assert int((d2[i] != wants[i][0]).sum()) == 0, i
        assert np.array_equal(f2[i].view(np.uint32), wants[i][1].view(np.uint32))
        check_xyz(x2[i], wants[i][2])
    d3 = D_.hip_batch_compute(p)(dl[:2], dr[:2]).cpu().numpy()     # without Q: disparities only
    assert np.array_equal(d3, d2[:2])
    # throughput mode (chained sweeps): the host entry stages groups of pairs for sgm_pipeline_batch_device
    e2 = cv.Engine(p, device=0)
    e2.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    disps, xyz = e2.compute_batch_host(L, R, Q)
    for i in range(N):
        assert int((disps[i] != wants[i][0]).sum()) == 0, i
        check_xyz(xyz[i], wants[i][2])


@pytest.mark.parametrize("D,mode", [(64, 1), (32, 0)])
def test_4k_small_d_schedule(D, mode):
    """D <= 64 takes its own schedule (per-role grouped pre-pass with band height 1, element-wise
    vertical kernel, in-row kernels): at 4K its boundary-state buffer is 3 volumes (3.1 GB at D = 64,
    32-bit offsets beyond 2^31) -- full size, both modes."""
    H, W = 2160, 3840
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, 7, mode)
    want, _, _ = oracle_frame(l, r, p)
    got, _, _, names = run_pipeline_device(l, r, p, None)
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} of {got.size} differ"


def test_4k_d512_hh_volumes_beyond_4_gib():
    """The largest supported numDisparities at 4K, MODE_HH: W1 = 3328 columns, C and S of 7.4 GB each
    (3.7 G elements: past 2^31 elements and past the 4 GiB a buffer descriptor spans).  This frame takes
    the routes small frames only reach through debug switches -- the int16 cost pipeline (the byte
    volume would pass 2 GiB), the pre-pass as three launches of the single-direction kernel with one
    descriptor per row, NP = 4 sweeps -- with real 64-bit offsets."""
    H, W, D, bs = 2160, 3840, 512, 7
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, bs, 1)
    want, _, _ = oracle_frame(l, r, p)
    got, _, _, names = run_pipeline_device(l, r, p, None)
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} of {got.size} differ"
    assert {"cost_hsum", "cost_vsum", "sweep_dn", "sweep_up"} <= set(names)


def test_notebook_setting_4k_d16_bs11():
    """The notebook as run (main.ipynb:781: ndisp=16, mindis=0, blockSize=11) on a 4K synthetic
    pair through the notebook-shaped functions.  bs=11 / P2=11616 is outside the worst-case
    headroom bound; the oracle confirms this input stays inside the regime."""
    import stereo_reconstruction_cv_amd as cv
    H, W, D = 2160, 3840, 16
    l, r = pair(H, W, D, 1234)
    p = nb_params(D, 11, 0)
    Q = synth.default_Q(W)
    want, wf, wxyz = oracle_frame(l, r, p, Q)
    disp, pts, mask = cv.run_disparity(l, r, Q, 16, 0)
    assert np.array_equal(disp.view(np.uint32), wf.view(np.uint32))
    check_xyz(pts, wxyz)
    assert np.array_equal(mask, O.valid_mask(wxyz, wf))
