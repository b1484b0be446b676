#!/usr/bin/env python3
"""Headline benchmark: dense stereo disparity throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--ingest resident|rank0]

A "step" = one pass of the hot path (SGBM compute -> float scaling -> reprojectImageTo3D) over
one batch of synthetic rectified pairs that is already resident in HBM.  One process per GPU:
started plainly with --gpus N > 1 this script launches its own N ranks (a child
`python -m torch.distributed.run`, decided before torch or HIP are touched) and relays rank 0's
JSON line; started under torch.distributed.run (RANK/WORLD_SIZE in the environment) it is a rank.
Frames are sharded by rank with no data-path collective (weak scaling); the timed region is
bracketed by barrier + synchronize on both sides and the maximum over ranks is reported.  Rank 0
prints ONE JSON line.

--ingest rank0: rank 0 owns every frame of the step; each step scatters the u8 pairs to their
ranks, computes, and gathers the int16 disparities (and XYZ when the workload reprojects) back
over torch.distributed ("nccl" = RCCL over xGMI), all inside the timed region
(stereo_reconstruction_cv_amd/dist.py).  Default "resident": every rank generates its own
frames before the timed region; after the timed region an N > 1 run still pushes one batch
through the scatter/compute/gather path and reports whether it reproduced the resident results
("rccl_ingest_check"), so the collective path is exercised on hardware by the default command.

Default workload "c3c5": 3840x2160, D=256, blockSize=7, MODE_HH (8 paths) + LR check +
sub-pixel + median + speckle + reprojection to XYZ, i.e. the union of BASELINE.json configs[2]
and configs[4]; the other configs are selectable with --workload and are parity-test cases
(tests/test_gpu_configs.py runs every one of them at full size against the oracle).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

NB = dict(disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32)

WORKLOADS = {
    # name: (H, W, D, blockSize, mode, pairs per GPU per step, reproject, description)
    "c1": (720, 1280, 64, 5, 0, 1, False, "1280x720 D=64 bs=5 5-path (BASELINE configs[0])"),
    "c2": (2160, 3840, 128, 7, 0, 1, False, "3840x2160 D=128 bs=7 5-path (configs[1])"),
    "c3": (2160, 3840, 256, 7, 1, 1, False, "3840x2160 D=256 bs=7 8-path (configs[2])"),
    "c4": (1080, 1920, 128, 7, 0, 8, False, "8x 1920x1080 D=128 bs=7 5-path per GPU (configs[3]: 64 frames over 8 GPUs)"),
    "c5": (2160, 3840, 256, 7, 0, 1, True, "3840x2160 D=256 bs=7 5-path + reproject (configs[4])"),
    "c3c5": (2160, 3840, 256, 7, 1, 1, True,
             "3840x2160 D=256 bs=7 MODE_HH 8-path + LR + subpixel + median + speckle + reprojectImageTo3D"),
    "c3c5x2": (2160, 3840, 256, 7, 1, 2, True,
               "two concurrent 3840x2160 D=256 bs=7 MODE_HH pairs per step on two HIP streams (+ reproject)"),
    "c3c5x3": (2160, 3840, 256, 7, 1, 3, True, "three concurrent 3840x2160 D=256 bs=7 MODE_HH pairs per step on three HIP streams (+ reproject)"),
    "c3c5x4": (2160, 3840, 256, 7, 1, 4, True, "four concurrent 3840x2160 D=256 bs=7 MODE_HH pairs per step on four HIP streams (+ reproject)"),
    "c3c5x6": (2160, 3840, 256, 7, 1, 6, True, "six concurrent 3840x2160 D=256 bs=7 MODE_HH pairs per step on six HIP streams (+ reproject)"),
    "c3c5x8": (2160, 3840, 256, 7, 1, 8, True, "eight 3840x2160 D=256 bs=7 MODE_HH pairs per step (+ reproject)"),
    "c3c5x12": (2160, 3840, 256, 7, 1, 12, True, "twelve 3840x2160 D=256 bs=7 MODE_HH pairs per step (+ reproject)"),
    "nb": (2160, 3840, 16, 11, 0, 1, True, "3840x2160 D=16 bs=11 5-path + reproject (the notebook as run)"),
    "tiny": (96, 480, 64, 7, 1, 2, True, "96x480 D=64 MODE_HH x2 (launcher rehearsal only, not a BASELINE config)"),
}

# stage (HIP-event bracket inside the engine) -> kernel that runs in it, for the roofline record
STAGE_KERNEL = {
    "chain_dn": "k_sweep_chain<NP,*,SWEEP_FIRST>", "chain_up": "k_sweep_chain<NP,*,SWEEP_ACCUM>",
    "sweep_dn": "k_sweep<NP,*,SWEEP_FIRST>", "sweep_up": "k_sweep<NP,*,SWEEP_ACCUM>", "sweep_up_wta": "k_sweep<NP,*,SWEEP_LAST>",
    "prepass_dn": "k_prepass3<NP,*>", "prepass_up": "k_prepass3<NP,*>", "path_W_wta": "k_rows_g<64,NP,*,PATH_LAST>",
    "path_W": "k_rows_g<GW,NP,*,PATH_ACCUM>", "wta": "k_wta_t", "cost_pix": "k_pix<NP>", "cost_box": "k_box_u8<R,NP>",
    "cost_hsum": "k_hsum<NP,RS>", "cost_vsum": "k_vsum_ring<SH2,NW>", "features": "k_features", "select_lr": "k_select",
    "median3": "k_median3", "speckle": "k_ccl_*", "to_float": "k_disp_to_float", "reproject": "k_reproject",
    "post": "k_post_*", "float_xyz": "k_float_xyz",
}


def sgbm_params(D, bs, mode):
    return dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs, mode=mode, **NB)


def source_stamp() -> str:
    """sha256 over the kernel sources: profiles/pmc_traffic.json is only trusted when it was
    measured on exactly these kernels (tools/pmc_to_json.py writes the same stamp)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "stereo_reconstruction_cv_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def min_traffic_model(stage: str, V: int, R: int, HW: int, mode: int) -> int | None:
    """Bytes a launch of this design has to move at least (used for roofline.achieved when no PMC
    record of the current kernels is committed).  V = 2*H*W1*D; R = rows per sweep band."""
    bnd = 3 * V // max(R, 1)
    return {
        "sweep_dn": 2 * V + bnd, "sweep_up": 3 * V + bnd, "sweep_up_wta": 2 * V + bnd,
        "prepass_dn": 3 * V + bnd, "prepass_up": 3 * V + bnd,
        "path_W_wta": 2 * V, "path_W": 3 * V, "wta": V + 8 * HW,
        "cost_pix": V // 2 + 14 * HW, "cost_box": V // 2 + V, "cost_hsum": V + 14 * HW, "cost_vsum": 2 * V,
    }.get(stage)


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int, argv: list[str]) -> int:
    """Parent of an N > 1 run started without a launcher: spawn the ranks as a CHILD process tree
    (never exec: nothing in this process has touched the GPU, and nothing will), relay rank 0's
    JSON line, return the children's exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for out in p.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line:
        print(line)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks produced no JSON line\n")
    return rc


class MockEngine:
    """BENCH_MOCK=1 only (tests/test_bench_launcher.py): stands in for the HIP engine so that the
    launcher, rendezvous, sharding, timing and JSON plumbing can run on a box without a GPU.  The
    JSON line says so in `data`; no number from it means anything."""

    def __init__(self, p, device=0):
        self.p = p

    def set_option(self, *_):
        pass

    def geometry(self, W):
        return self.p["numDisparities"], W - self.p["numDisparities"]

    def algorithmic_bytes(self, H, W, xyz=False):
        return 1

    def pipeline_device(self, *a):
        time.sleep(0.002)

    def stage_times(self):
        return [("sweep_dn", 1.0, 1), ("cost_pix", 0.5, 1), ("_wall", 1.5, 0)]

    def synchronize(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3c5", choices=sorted(WORKLOADS))
    ap.add_argument("--ingest", default="resident", choices=("resident", "rank0"),
                    help="rank0: frames start on rank 0; scatter / gather over torch.distributed inside the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline runs (0 = auto)")
    ap.add_argument("--verify", action="store_true", help="also run the oracle on the full frame and compare")
    ap.add_argument("--stages", action="store_true", help="print the per-stage HIP-event table to stderr")
    ap.add_argument("--concurrent", type=int, default=0,
                    help="engines (HIP streams) working on different pairs at the same time; 0 = workload default")
    ap.add_argument("--schedule", type=int, default=1, help="0: one kernel per path direction, 1: fused sweeps behind a pre-pass, 2: chained sweeps (no pre-pass)")
    ap.add_argument("--chain-wgs", type=int, default=0, help="schedule 2: workgroups per frame of a sweep launch (0 = automatic)")
    ap.add_argument("--batch", action="store_true",
                    help="hand all pairs of a step to sgm_pipeline_batch_device of ONE engine (schedule 2: one chained sweep launch per pass for all of them)")
    ap.add_argument("--debug", type=int, default=0, help="SGM_OPT_DEBUG bit mask (A/B measurements; include/sgm_hip.h)")
    ap.add_argument("--prepass-rows", type=int, default=0, help="rows per chunk of the boundary pre-pass (0 = automatic)")
    ap.add_argument("--sweep-rows", type=int, default=0, help="rows per band of the fused sweeps (0 = automatic)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become the parent of N ranks (before torch / HIP are imported)
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    from stereo_reconstruction_cv_amd import dist as sharding
    from stereo_reconstruction_cv_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    mock = os.environ.get("BENCH_MOCK") == "1"
    # BENCH_REHEARSE=1: functional rehearsal of the N > 1 code path on a box with fewer GPUs than
    # ranks -- every rank uses cuda:(local_rank % device_count) and the rendezvous runs over gloo
    # (RCCL refuses two ranks on one GPU).  Its numbers mean nothing; the driver never sets it.
    rehearse = os.environ.get("BENCH_REHEARSE") == "1" or mock
    if mock:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
        if rehearse:
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    comm_dev = torch.device("cpu") if rehearse else dev   # where tensors sit while they travel

    if mock:
        Engine = MockEngine
        OPT_PROFILE = OPT_SCHEDULE = 0
    else:
        import stereo_reconstruction_cv_amd as cv
        from stereo_reconstruction_cv_amd import _lib
        Engine = cv.Engine
        OPT_PROFILE, OPT_SCHEDULE = _lib.SGM_OPT_PROFILE, _lib.SGM_OPT_SCHEDULE

    def sync():
        if not mock:
            torch.cuda.synchronize(dev)

    alloc = torch.zeros if mock else torch.empty   # (the mock engine writes nothing)

    H, W, D, bs, mode, ppg, with_xyz, desc = WORKLOADS[args.workload]
    p = sgbm_params(D, bs, mode)
    Q = synth.default_Q(W)

    # ---- inputs resident in HBM before the timed region ----
    # resident: every rank makes its own ppg pairs; rank0 ingest: rank 0 makes all world * ppg
    if args.ingest == "rank0":
        seeds = [1234 + i for i in range(ppg * world)] if rank == 0 else []
    else:
        seeds = [1234 + rank * ppg + i for i in range(ppg)]
    pairs = [synth.make_pair(H, W, D, seed=s)[:2] for s in seeds]
    d_left = [torch.from_numpy(a).to(dev) for a, _ in pairs]
    d_right = [torch.from_numpy(b).to(dev) for _, b in pairs]
    d_disp = [alloc((H, W), dtype=torch.int16, device=dev) for _ in range(ppg)]
    d_dispf = [alloc((H, W), dtype=torch.float32, device=dev) for _ in range(ppg)] if with_xyz else None
    d_xyz = [alloc((H, W, 3), dtype=torch.float32, device=dev) for _ in range(ppg)] if with_xyz else None
    sync()

    # `nconc` engines = HIP streams; pair i of a step runs on engine i % nconc, so independent
    # pairs overlap on the GPU (each engine owns its own device buffers)
    nconc = 1 if args.batch else max(1, min(args.concurrent or ppg, ppg))
    engines = []
    for _ in range(nconc):
        e = Engine(p, device=local_rank)
        e.set_option(OPT_PROFILE, 1)
        e.set_option(OPT_SCHEDULE, args.schedule)
        if not mock and args.debug:
            e.set_option(_lib.SGM_OPT_DEBUG, args.debug)
        if not mock and args.prepass_rows:
            e.set_option(_lib.SGM_OPT_PREPASS_ROWS, args.prepass_rows)
        if not mock and args.sweep_rows:
            e.set_option(_lib.SGM_OPT_SWEEP_ROWS, args.sweep_rows)
        if not mock and args.chain_wgs:
            e.set_option(_lib.SGM_OPT_CHAIN_WGS, args.chain_wgs)
        engines.append(e)
    eng = engines[0]

    def run_local(lefts, rights, disps, dispfs, xyzs):
        """enqueue every pair of this rank on its engine, then collect the HIP-event stage times"""
        acc = []
        n = len(lefts)
        if args.batch:
            ptrs = lambda ts: [t.data_ptr() for t in ts] if ts is not None else None
            eng.pipeline_batch_device(ptrs(lefts), ptrs(rights), H, W, W, Q if with_xyz else None, ptrs(disps),
                                      ptrs(dispfs) if with_xyz else None, ptrs(xyzs) if with_xyz else None)
            st = eng.stage_times()       # (synchronises; the first pair's stages + the joint sweep launches)
            return [st] * n
        for i0 in range(0, n, nconc):
            group = range(i0, min(i0 + nconc, n))
            for i in group:  # enqueue on every stream first ...
                engines[i % nconc].pipeline_device(
                    lefts[i].data_ptr(), rights[i].data_ptr(), H, W, W, Q if with_xyz else None, disps[i].data_ptr(),
                    dispfs[i].data_ptr() if with_xyz else None, xyzs[i].data_ptr() if with_xyz else None)
            for i in group:  # ... then collect the stage times (synchronises that stream)
                acc.append(engines[i % nconc].stage_times())
        return acc

    stage_log = []

    def ingest_compute(l, r):
        """compute() of dist.run_sharded: this rank's shard [n, H, W] -> (disp[, xyz])"""
        l, r = l.to(dev), r.to(dev)
        n = l.shape[0]
        disp = alloc((n, H, W), dtype=torch.int16, device=dev)
        dispf = alloc((n, H, W), dtype=torch.float32, device=dev) if with_xyz else None
        xyz = alloc((n, H, W, 3), dtype=torch.float32, device=dev) if with_xyz else None
        sync()
        stage_log.extend(run_local([l[i] for i in range(n)], [r[i] for i in range(n)], [disp[i] for i in range(n)],
                                   [dispf[i] for i in range(n)] if with_xyz else None,
                                   [xyz[i] for i in range(n)] if with_xyz else None))
        if with_xyz:
            return disp.to(comm_dev), xyz.to(comm_dev)
        return disp.to(comm_dev)

    def step_ingest(lefts0, rights0):
        """one timed step of --ingest rank0: scatter u8 frames, compute, gather disparity (+ XYZ)"""
        return sharding.run_sharded(ingest_compute, lefts0, rights0, src=0, device=comm_dev)

    all_left = all_right = None
    if args.ingest == "rank0" and rank == 0:
        all_left = torch.stack(d_left).to(comm_dev)
        all_right = torch.stack(d_right).to(comm_dev)

    def step():
        if args.ingest == "rank0":
            step_ingest(all_left, all_right)
            return []
        return run_local(d_left, d_right, d_disp, d_dispf, d_xyz)

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    stage_log.clear()
    barrier()
    t0 = time.perf_counter()
    stage_acc = []
    for _ in range(args.steps):
        stage_acc.extend(step())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.ingest == "rank0":
        stage_acc = list(stage_log)

    # ---- the collective path on hardware, outside the timed region (default N > 1 runs) ----
    ingest_check = None
    if world > 1 and args.ingest == "resident":
        t1 = time.perf_counter()
        ls = torch.stack(d_left).to(comm_dev)
        rs = torch.stack(d_right).to(comm_dev)
        gl = sharding.gather_results(ls, ppg * world, dst=0)       # frames to rank 0 ...
        gr = sharding.gather_results(rs, ppg * world, dst=0)
        res = step_ingest(gl, gr)                                   # ... and through scatter / compute / gather
        mine = torch.stack(d_disp).to(comm_dev)
        ref = sharding.gather_results(mine, ppg * world, dst=0)
        if rank == 0:
            got = res[0] if isinstance(res, tuple) else res
            ingest_check = {"ok": bool(torch.equal(got, ref)), "frames": ppg * world, "xyz": bool(with_xyz),
                            "backend": "gloo" if rehearse else "nccl", "s": None}
        barrier()
        if rank == 0:
            ingest_check["s"] = time.perf_counter() - t1

    # ---- per-stage HIP-event durations (this rank) ----
    stage_acc = [[t for t in st if t[0] != "_wall"] for st in stage_acc]
    names = [n for n, _, _ in stage_acc[0]]
    ms = np.array([[m for _, m, _ in st] for st in stage_acc])  # [frames][stages]
    mean_ms = ms.mean(axis=0)
    if args.stages and rank == 0:
        for n, m in zip(names, mean_ms):
            print(f"  {n:<14s} {m:9.4f} ms", file=sys.stderr)
        print(f"  {'sum':<14s} {mean_ms.sum():9.4f} ms   wall/frame {dt / args.steps / ppg * 1e3:9.4f} ms", file=sys.stderr)

    # ---- roofline of the dominant kernel ----
    # Stages are grouped by the kernel that runs in them (both pre-pass stages are k_prepass3; the two
    # sweeps are different template instantiations = different kernels, as in rocprofv3's kernel
    # stats).  Dominant = the kernel with the largest total HIP-event time per frame, every stage a
    # candidate.  `achieved` = bytes one launch moves through HBM / its average launch duration
    # (HIP events: total stage time / launches): the PMC record of exactly these kernels when one is
    # committed (profiles/pmc_traffic.json, stamped with the source hash), else the minimal-traffic
    # model of the design -- a fraction of peak by construction.  The SURVEY 8(d)
    # algorithmic model (unfused: 3 V per path scan) is kept as separate fields.
    _, W1 = eng.geometry(W)
    V = 2 * H * max(W1, 0) * D
    R = min(11, max(4, -(-H // (200 if mode else 240))))   # rows per sweep band (sweep_rows_for in sgm_engine.hip)
    launches = np.array([[n for _, _, n in st] for st in stage_acc]).mean(axis=0)
    by_kernel = {}
    for n, m, nl in zip(names, mean_ms, launches):
        k = STAGE_KERNEL.get(n, n)
        rec = by_kernel.setdefault(k, {"ms": 0.0, "launches": 0.0, "stages": []})
        rec["ms"] += float(m)
        rec["launches"] += float(max(nl, 1))
        rec["stages"].append(n)
    kdom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom_stages = by_kernel[kdom]["stages"]
    k_ms = by_kernel[kdom]["ms"] / by_kernel[kdom]["launches"]     # average launch duration
    alg_model = {"sweep_dn": 12 * V, "sweep_up": 12 * V, "sweep_up_wta": 12 * V, "path_W_wta": 3 * V, "path_W": 3 * V,
                 "cost_pix": V // 2, "cost_box": V // 2, "cost_hsum": V // 2, "cost_vsum": V // 2}
    alg_launch = sum(alg_model.get(n, 0) for n in dom_stages) / by_kernel[kdom]["launches"]
    traffic, basis, pmc_stages = None, "model", {}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("workload") == args.workload and rec.get("source_stamp") == source_stamp():
                pmc_stages = rec.get("stages", {})
                if all(n in pmc_stages for n in dom_stages):
                    traffic = sum(pmc_stages[n]["traffic_bytes_per_launch"] for n in dom_stages) / by_kernel[kdom]["launches"]
        except (OSError, ValueError, KeyError):
            traffic = None
    moved = traffic
    if moved is None:
        per_stage = [min_traffic_model(n, V, R, H * W, mode) for n in dom_stages]
        moved = (sum(per_stage) / by_kernel[kdom]["launches"]) if all(x is not None for x in per_stage) else None
    else:
        basis = "pmc"
    achieved = (moved / (k_ms * 1e-3) / 1e9) if moved else None
    # the same kernel where it has the GPU to itself (the downward pre-pass; the upward one shares it with the sweep)
    alone = None
    if "prepass_dn" in dom_stages and moved:
        i = names.index("prepass_dn")
        a_ms = float(mean_ms[i]) / max(1, int(launches[i]))
        alone = {"stage": "prepass_dn", "avg_launch_ms": a_ms, "achieved": moved / (a_ms * 1e-3) / 1e9,
                 "frac": moved / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    whole_traffic = sum(s.get("traffic_bytes_per_launch", 0) for s in pmc_stages.values()) or None

    frames = args.steps * ppg * world
    mdisp = frames * H * W * D / dt / 1e6
    alg_bytes = eng.algorithmic_bytes(H, W, with_xyz)
    step_s = dt / args.steps

    out = {
        "metric": "Mdisparities/s",
        "value": mdisp,
        "unit": "Mdisparities/s (H*W*D per second, whole job)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int16",
        "data": "synthetic" + (" (BENCH_MOCK: no engine, launcher rehearsal, numbers meaningless)" if mock else
                               " (BENCH_REHEARSE: ranks share GPUs, numbers meaningless)" if rehearse else ""),
        "config": {"workload": f"{args.workload}: {desc}", "height": H, "width": W, "numDisparities": D,
                   "blockSize": bs, "mode": "MODE_HH" if mode else "MODE_SGBM", "pairs_per_gpu_per_step": ppg,
                   "global_pairs_per_step": ppg * world, "streams_per_gpu": nconc, "batch_entry": bool(args.batch),
                   "schedule": ("one kernel per direction", "fused sweeps", "chained sweeps")[args.schedule],
                   **({"debug_mask": args.debug} if args.debug else {}),
                   "world_size": dist.get_world_size() if world > 1 else 1,
                   "frames_per_rank": [b - a for a, b in (sharding.shard_range(ppg * world, r, world) for r in range(world))],
                   "ingest": args.ingest,
                   "parallelism": (f"frame-sharded x{world}, rank 0 scatters frames / gathers results over "
                                   f"{'gloo' if rehearse else 'RCCL'} inside the step" if args.ingest == "rank0"
                                   else f"frame-sharded x{world}, no data-path collective")},
        "pairs_per_s": frames / dt,
        "algorithmic_GBps_whole_step": alg_bytes * ppg * world / step_s / 1e9,
        "frac_whole_step": alg_bytes * ppg * world / step_s / 1e9 / HBM_PEAK_GBS / world,
        "traffic_GBps_whole_step": (whole_traffic * ppg / step_s / 1e9) if whole_traffic else None,
        "traffic_frac_whole_step": (whole_traffic * ppg / step_s / 1e9 / HBM_PEAK_GBS) if whole_traffic else None,
        "roofline": {"bound": "hbm", "kernel": kdom, "stages": dom_stages,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "traffic": traffic, "basis": basis, "bytes_per_launch": moved, "avg_launch_ms": k_ms,
                     "launches_per_step": by_kernel[kdom]["launches"] * ppg, "kernel_ms_per_frame": by_kernel[kdom]["ms"],
                     "algorithmic_bytes_per_launch": alg_launch,
                     "algorithmic_GBps": alg_launch / (k_ms * 1e-3) / 1e9,
                     "alone": alone,
                     "note": "dominant = kernel with the largest total HIP-event time per frame (all stages); achieved = "
                             "bytes one launch moves (PMC 2*FETCH_SIZE+WRITE_SIZE of these kernels when basis=pmc, else "
                             "the design's minimal-traffic model) / average launch duration; the upward pre-pass shares "
                             "the GPU with the downward sweep (auxiliary stream), which lengthens both; algorithmic_* = "
                             "SURVEY 8(d) unfused model (a fused sweep = 4 path scans = 12 V; the pre-pass has no "
                             "counterpart = 0)"},
        "stage_ms": {n: float(m) for n, m in zip(names, mean_ms)},
    }
    if ingest_check is not None:
        out["rccl_ingest_check"] = ingest_check

    # ---- CPU baseline: the oracle (a scalar port, one thread per frame like upstream), bounded
    # sample, rank 0 at N = 1; plus the frame-parallel all-cores rate (one frame per core) ----
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not mock:
        from concurrent.futures import ThreadPoolExecutor

        from oracle import oracle as O  # test infrastructure; timed here only as the CPU baseline
        model, nproc, usable = cpu_info()
        rows = args.cpu_rows or min(H, max(64, int(1080 * (3840 * 256) / (W * D))))
        rows = min(rows, H)
        l, r = pairs[0][0][:rows], pairs[0][1][:rows]
        O.lib()
        t0 = time.perf_counter()
        want = O.sgbm_compute(l, r, **p)
        if with_xyz:
            O.reproject(O.disp_to_float(want), Q)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": rows * W * D / cdt / 1e6, "unit": "Mdisparities/s", "cores": 1, "kind": "port",
            "cpu_model": model, "nproc": nproc, "usable_cores": usable,
            "sample": f"rows 0..{rows - 1} of the same {W}x{H} pair (full width, D={D}, same parameters), "
                      f"{cdt:.1f} s on one host core; OpenCV-4.11-equivalent C restatement, not cv2 "
                      f"(upstream MODE_SGBM / MODE_HH are single-threaded)",
        }
        out["speedup_vs_cpu_baseline"] = mdisp / out["cpu_baseline"]["value"]
        # frame-parallel: `usable` threads, each one frame-sample of its own (ctypes releases the GIL)
        nthreads = max(1, min(usable, 64))
        prow = max(32, min(rows, int(rows * 1.0 / max(cdt, 1e-3))))   # about 1 s per thread alone (several when all contend for memory)
        pl, pr = pairs[0][0][:prow], pairs[0][1][:prow]

        def one(_):
            d = O.sgbm_compute(pl, pr, **p)
            if with_xyz:
                O.reproject(O.disp_to_float(d), Q)
            return 0

        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(one, range(nthreads)))
        pdt = time.perf_counter() - t0
        out["cpu_baseline"]["all_cores"] = {
            "value": nthreads * prow * W * D / pdt / 1e6, "unit": "Mdisparities/s", "cores": nthreads,
            "sample": f"{nthreads} threads, each rows 0..{prow - 1} of the pair (one frame per core, the batch "
                      f"configs' CPU counterpart), {pdt:.1f} s wall",
        }
        out["speedup_vs_cpu_all_cores"] = mdisp / out["cpu_baseline"]["all_cores"]["value"]
    if args.verify and world == 1 and rank == 0 and not mock and args.ingest == "resident":
        from oracle import oracle as O  # the checker (test infrastructure)
        t0 = time.perf_counter()
        full = O.sgbm_compute(pairs[0][0], pairs[0][1], taps=False, **p)
        got = d_disp[0].cpu().numpy()
        nbad = int((got != full).sum())
        out["verify"] = {"full_frame_mismatches": nbad, "pixels": int(got.size),
                         "valid_fraction": float((got >= 0).mean()), "oracle_s": time.perf_counter() - t0}
        if with_xyz:
            ref = O.reproject(O.disp_to_float(full), Q)
            x = d_xyz[0].cpu().numpy()
            fin = np.isfinite(ref)
            out["verify"]["xyz_finite_mask_equal"] = bool(np.array_equal(np.isfinite(x), fin))
            out["verify"]["xyz_max_rel_err"] = float(np.max(np.abs(x[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30))) if fin.any() else 0.0

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
