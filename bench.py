#!/usr/bin/env python3
"""Headline benchmark: Mvoxel.views/s of the space carve, 512^3 x 72 synthetic views per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts its N ranks itself (this script
again, with the environment ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would give them) as
fresh children BEFORE it makes any GPU call, relays rank 0's JSON line and exits with their code.  Under
torch.distributed.run (the driver's N > 1 command) it is a rank as before.

A "step" is one pass of the hot path over one batch of synthetic input: clear the grid, take
the 72 uint8 masks that are ALREADY RESIDENT IN HBM, pack them to bit tiles and carve the
whole slab (the ``Backprojection.process_fileset`` work after ingest).  The result stays in HBM.

Workload (config.workload): BASELINE cfg 3, 512^3 voxels x 72 views, scene S1 "plant"
(SURVEY.md 8d).  N GPUs: weak scaling -- a near-cubic grid of about N x 512^3 voxels (N=8 is
BASELINE cfg 4, 1024^3) whose x-planes are dealt round-robin over the ranks, so every rank
carves ~512^3 voxels holding the same share of the object; the carve itself needs no collective
(voxels are independent).  For N > 1 ``value`` is SURVEY 8d's ``t_device + collective``: K steps of
carve + pack to 2 bits per label + RCCL all-gather, every GPU ending every step with the whole grid (packed: the
form ``proc3d.vol2pcd`` reads), the collective of step k running beside the carve of step k + 1 and the last one
waited for inside the timed region.  Beside it: ``value_carve_only`` (no collective), ``assembly`` (the same
steps one after the other, with the unpack kernel, at 1 bit, as int8), the time of ``gather_to_host`` (the
reference's ``get_values``, cl.py:229-232), and ``strong``: BASELINE's own metric -- ONE 512^3 x 72 grid split
over the N ranks -- with and without the assembly.

Beside the headline, in the same line (N = 1):
  stream   one launch per view as the reference does (cl.py:223-226), the formulation SURVEY 8d's
           algorithmic-bytes figure (~4 B per voxel.view) is defined on;
  scenes   the fused carve on S2 "solid", S3 "noise" and "dense" (a 20 % solid object, ~30 %
           foreground: no all-empty / all-white shortcut applies to most of it);
  average  the `average` kernel (backprojection.c:36-55) on uint8 binary, uint8 grey and float32
           masks, against the ceiling its own instruction mix sets at the measured per-instruction costs;
  e2e      the reference's real interface (cl.py:190-232): 72 uint8 masks in HOST memory through
           ``process_view`` to int32 labels in HOST memory (PCIe both ways; never ``value``);
  cpu_baseline  the oracle on this box's host cores, on a bounded sample of the same workload (S1, and
           S2 "solid" beside it), and the NumPy float32 per-view restatement at cfg 1 (128^3 x 12).
The ``roofline`` object describes the ``value`` path with ITS OWN algorithmic bytes;
``equiv_streaming_note`` restates its speed in units of the streaming roofline (a speed ratio,
never an HBM-utilisation claim -- SURVEY 8d honesty guard).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# The `average` kernel's ceiling (round 5): it is bound by what it ISSUES -- waves wait for the
# vector ALU, not for memory (SQ_WAIT_INST_ANY 46 % of the wave-cycles, vector ALU active 85 % of the SIMD time:
# profiles/r05_avg_*_counters.json) -- and a vector instruction does not cost "2 cycles": tools/probes/valu_probe.hip
# measures, per wavefront instruction and SIMD with 8 wavefronts resident, 2.5-2.7 cycles for v_mul_f32 / v_add_f32 /
# and / or / shifts right / moves, 4.1-4.7 for v_fma_f32, comparisons, conversions and every three-operand form, 8.3
# for v_rcp_f32 (profiles/r05_valu_probe.txt, cycles at the 2.4 GHz the probe assumes).  The ceiling prices the
# kernel's own instruction mix (the SQ_INSTS_VALU_* counters of a run on random grey masks, every (brick, view) pair
# projected) at those costs; instructions no class counter names are priced as the cheapest class, so the ceiling is
# a LOWER bound of the time the mix needs and `frac` = ceiling / measured a lower bound of the issue utilisation.
# Round 6: what a vector instruction costs, from tools/probes/valu_probe2.hip -- s_memtime around the
# loop AND the wall clock, no assumed frequency (profiles/r06_valu_probe2.txt).  In shader cycles a v_fma_f32 issues every
# 1.7-2.0 cycles per SIMD with four or more waves (the guide's SIMD-32 rate, MI355X_MICROARCH.md "Per-instruction cycle
# constants") -- but with every SIMD of the chip issuing vector instructions the clock is 1.3-1.55 GHz, not 2.4, so in
# TIME a wave-instruction costs what round 5's table said in "cycles at 2.4 GHz".  The costs below are nanoseconds per
# wavefront instruction and SIMD at 8 waves per SIMD (wall time / instructions: nothing assumed).
VALU_COST_NS = {"SQ_INSTS_VALU_ADD_F32": 1.098, "SQ_INSTS_VALU_MUL_F32": 1.134, "SQ_INSTS_VALU_FMA_F32": 1.317,
                "SQ_INSTS_VALU_TRANS_F32": 3.687, "SQ_INSTS_VALU_CVT": 1.956, "SQ_INSTS_VALU_INT32": 1.32,
                "SQ_INSTS_VALU_INT64": 1.83, "other": 1.06}
# ... and what the SAME classes cost when they are one stream -- a projection's 32 instructions for four voxels side by
# side, as the kernels issue them: 1.591 ns per instruction, 25 % above the sum of its classes (the classes are not
# independent: clock and issue ports are shared)
VALU_MIX_STREAM_NS = 1.591
VALU_PROBE = "profiles/r06_valu_probe2.txt"
AVG_COUNTERS = {"u8": "profiles/r05_avg_u8_grey_counters.json", "f32": "profiles/r05_avg_f32_grey_counters.json"}
AVG_COUNTERS_NVV = 512 ** 3 * 72  # the voxel.views of the run the counters were taken on
SIMDS = 1024


def valu_ceiling(form, nvv):
    """(ceiling_ms, lane_ops_per_voxel_view, ns_per_wave_voxel_view, mix, mix_stream_ms) of the averaging kernel for `nvv`
    voxel.views, from the committed counters of its `form` ("u8" / "f32") priced at VALU_COST_NS (a lower bound of the time
    the mix needs: every class at the rate it reaches alone) and at VALU_MIX_STREAM_NS (the rate a stream of that mix
    reaches); None without the counters."""
    try:
        d = json.load(open(os.path.join(ROOT, AVG_COUNTERS[form])))
        k = [v for n, v in d["kernels"].items() if n.startswith("average_brick_kernel")][0]
    except Exception:
        return None
    total = float(k["SQ_INSTS_VALU"])
    named = {c: k.get(c, 0.0) for c in VALU_COST_NS if c != "other"}
    mix = dict(named, other=max(0.0, total - sum(named.values())))
    ns = sum(mix[c] * VALU_COST_NS[c] for c in mix)  # per dispatch of AVG_COUNTERS_NVV voxel.views, summed over SIMDs
    scale = nvv / AVG_COUNTERS_NVV
    return (ns * scale / SIMDS * 1e-6, total * 64.0 / AVG_COUNTERS_NVV, ns / (AVG_COUNTERS_NVV / 64.0),
            {c: v * 64.0 / AVG_COUNTERS_NVV for c, v in mix.items()}, total * VALU_MIX_STREAM_NS * scale / SIMDS * 1e-6)


# Weak scaling: N GPUs carve a near-cubic grid of ~N x 512^3 voxels (N = 8: 1024^3, BASELINE cfg 4),
# x-planes dealt round-robin over the ranks.  Shapes for n = 512: nx divisible by N, ny by 16 and
# nz by 64 (whole bricks of the dense stage, include/spacecarve.h SC_OPT_BRICK; not required):
GRIDS_512 = {1: (512, 512, 512), 2: (640, 640, 640), 4: (808, 800, 832), 8: (1024, 1024, 1024)}


COLLECTIVE = False
COMM_STUCK = False  # a thread of this process never came back from sc_comm_create (the ranks fell back together)
BARRIER = None  # the library communicator's barrier once it exists (sc_comm_barrier), else torch.distributed's


def _barrier(dist):
    """The barrier that brackets every timed region.  With the library's own RCCL communicator it is a 16-byte all-gather
    on that communicator, waited for on the host (~20 us); torch.distributed's gloo barrier -- the control plane then --
    costs 0.3-0.5 ms over loopback TCP, a tenth of twenty 0.17 ms steps."""
    if BARRIER is not None:
        BARRIER()
    else:
        dist.barrier()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # a fused step is ~0.25 ms: 200 steps = 50 ms
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=512, help="per-GPU grid edge (voxels)")
    ap.add_argument("--views", type=int, default=72)
    ap.add_argument("--scene", default="plant", choices=["plant", "solid", "noise", "dense"])
    ap.add_argument("--path", default="fused", choices=["fused", "stream"],
                    help="schedule reported as `value` (the other is reported beside it)")
    ap.add_argument("--gather", default="none", choices=["none", "allgather", "allgather8", "allreduce"],
                    help="also time this other assembly once (N > 1 always times the int8 all-gather)")
    ap.add_argument("--extra-steps", type=int, default=10, help="steps per extra scene / averaging form (0 = skip)")
    ap.add_argument("--assembly-steps", type=int, default=20, help="N > 1: steps of carve + all-gather")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--e2e-reps", type=int, default=3, help="host masks -> host labels repetitions (0 = skip)")
    ap.add_argument("--strong-steps", type=int, default=20, help="N > 1: steps of the 512^3 grid split over the ranks")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--traffic-passes", default="auto", choices=["auto", "on", "off"],
                    help="N = 1, fused path: measure roofline.traffic in this invocation -- two short child runs of this "
                         "script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes), started "
                         "before this process touches the GPU; auto = when rocprofv3 is on PATH and this process is "
                         "not itself being profiled; off / failure: the committed file is cited instead")
    ap.add_argument("--cold-reps", type=int, default=2, help="fresh engines timed for cold_first_batch (0 = skip)")
    ap.add_argument("--cold-process", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: a fresh child process (started before this one touches the GPU) times import -> "
                         "sc_create -> ONE 72-view batch -> synchronize, once: cold_process_first_batch_ms")
    ap.add_argument("--cold-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cold-files-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--parity-check", default="on", choices=["on", "off"],
                    help="outside every timed region: fused digest == per-view digest, a 20 000-voxel closed-form "
                         "sample against the oracle's projection, the histogram; a mismatch exits non-zero")
    ap.add_argument("--scene-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "sc_bench_scene"),
                    help="prefix of the .npy cache of the synthetic masks (the child passes reuse the parent's)")
    ap.add_argument("--skip-other-path", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="engine option KEY=VALUE (sc_set_option)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "torch-nccl", "gloo"],
                    help="nccl (default): the collectives of the data path are enqueued by the library's own RCCL communicator "
                         "(sc_comm_*), torch.distributed (gloo) only carries the barriers and the MAX over ranks; torch-nccl: "
                         "torch's NCCL process group for everything (rounds 2-5); gloo + --share-device rehearses the N>1 "
                         "path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use device 0 (rehearsal)")
    ap.add_argument("--pipelined-steps", type=int, default=40,
                    help="N = 1: steps of the two-engines-in-turn throughput leg (0 = skip)")
    ap.add_argument("--twin-engine", type=int, default=1,
                    help="N > 1 (and --rccl-rehearsal): 1 = two engines take the steps of the assembled headline in turn, 0 = one engine")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="seconds sc_comm_create (ncclCommInitRank) may take on a rank before ALL ranks fall back to the "
                         "staged gloo collectives")
    ap.add_argument("--rccl-rehearsal", action="store_true",
                    help="one rank, but with an RCCL process group of one: barriers, the timing all-reduce and the "
                         "assembly collectives run as they do at N > 1 (reported under 'assembly')")
    return ap.parse_args()


def global_shape(n, gpus):
    if n == 512 and gpus in GRIDS_512:
        return list(GRIDS_512[gpus])
    # general case: a cube of about gpus * n^3 voxels, nx a multiple of gpus, ny of 16, nz of 64
    edge = (gpus * n ** 3) ** (1.0 / 3.0)
    nx = max(gpus, int(round(edge / gpus)) * gpus)
    ny = max(16, int(round(edge / 16)) * 16)
    nz = max(64, int(round(edge / 64)) * 64)
    return [nx, ny, nz]


def run_steps(engine, nat, K, R, t, masks_dev, V, H, W, steps, vpl):
    engine.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
    for _ in range(steps):
        engine.clear()
        engine.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
        engine.flush()


def timed(engine, nat, torch, dist, args_tuple, steps, vpl, world, time_kernels=2):
    """Barrier + synchronize on both sides, MAX over ranks; returns (seconds, kernel stats).
    time_kernels=2: HIP events around the carve kernel only (the roofline's kernel); 1: around
    every kernel (each pair costs stream time, so the breakdown is taken in a separate pass)."""
    world = 2 if COLLECTIVE else world  # barriers and the MAX over ranks also in the one-rank RCCL rehearsal
    span = time_kernels == "span"  # ONE event pair around all the steps: nothing between the batches
    engine.set_option(nat.SC_OPT_TIME_KERNELS, 0 if span else time_kernels)
    engine.reset_kernel_stats()
    # the events of the timed launches exist before the clock starts (creating one costs the host tens of
    # microseconds; the warm-up cannot be relied on to have left enough of them)
    engine.set_option(nat.SC_OPT_RESERVE_EVENTS,
                      8 if span else min(65536, (2 if time_kernels == 2 else 16) * steps * (1 if vpl == 0 else 80) + 64))
    if world > 1:
        _barrier(dist)
    engine.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if span:
        engine.span_begin()
    run_steps(engine, nat, *args_tuple, steps, vpl)
    span_ms = engine.span_end() if span else 0.0  # waits for the second event
    engine.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        _barrier(dist)
    dt = time.perf_counter() - t0
    if world > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = {}
    for name, kid in (("carve", nat.SC_KERNEL_CARVE), ("list", nat.SC_KERNEL_LIST),
                      ("pack", nat.SC_KERNEL_PACK), ("fill", nat.SC_KERNEL_FILL),
                      ("flags", nat.SC_KERNEL_FLAGS), ("step", nat.SC_KERNEL_STEP)):
        n, ms = engine.kernel_stats(kid)
        stats[name] = {"launches": n, "total_ms": ms, "avg_ms": (ms / n if n else 0.0)}
    if span:
        stats["step"] = {"launches": steps, "total_ms": span_ms, "avg_ms": span_ms / steps,
                         "timing": "one HIP event pair on the engine's stream around all the steps"}
    engine.set_option(nat.SC_OPT_TIME_KERNELS, 0)
    engine.reset_kernel_stats()
    return dt, stats


def cpu_baseline(shape, origin, vs, views, budget_s):
    """The oracle (a port of the reference kernel, oracle/spacecarve_oracle.c) on this box's
    host cores, on a bounded sample: a central block of X-planes of the same grid, all views.
    Threads: the box's CPU share for one GPU (16), or fewer if fewer are available."""
    from oracle import oracle_c
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    nx, ny, nz = shape
    plane = ny * nz
    V = len(views)
    masks32 = [np.ascontiguousarray(m, dtype=np.int32) for _, _, _, m in views]  # cl.py:215, untimed

    def run(planes):
        nonlocal cores
        i0 = (nx - planes) // 2
        vol = oracle_c.OracleVolume(shape, origin, vs, "carving", 0)
        t0 = time.perf_counter()
        for (K, R, t, _), m32 in zip(views, masks32):
            vol.process_view(K, R, t, m32, nthreads=cores, begin=i0 * plane, end=(i0 + planes) * plane)
        return time.perf_counter() - t0

    planes = max(1, min(nx, 8))
    t = run(planes)
    while planes < nx and t < budget_s / 3.0:
        planes = min(nx, max(planes * 2, int(planes * (budget_s * 0.7) / max(t, 1e-3))))
        t = run(planes)
    reps = 1
    if t < budget_s / 3.0:  # the whole grid is quicker than the budget: repeat it
        reps = int(min(50, max(2, budget_s * 0.7 / max(t, 1e-3))))
        t = sum(run(planes) for _ in range(reps)) / reps
    vv = planes * plane * V
    # one thread, on a thinner block (same code path, nthreads=1)
    cores_all = cores
    cores = 1
    p1 = max(1, min(nx, planes // 16))
    t1 = run(p1)
    cores = cores_all
    extra = {}
    if budget_s >= 4:
        # S2 "solid" (all-foreground masks: nothing is ever carved, every voxel does every view), a quarter of the budget
        from plant3dvision_amd import scenes as _sc
        _, _, _, sviews = _sc.make_scene(tuple(shape), V, "solid")
        full32 = np.ascontiguousarray(sviews[0][3], dtype=np.int32)
        p2 = max(1, min(nx, int(planes * 0.25 * budget_s / max(t, 1e-3) / 4)))  # S2 never stops early: ~4x the work per voxel
        i0 = (nx - p2) // 2
        vol = oracle_c.OracleVolume(shape, origin, vs, "carving", 0)
        t0 = time.perf_counter()
        for (K, R, tq, _) in sviews:
            vol.process_view(K, R, tq, full32, nthreads=cores, begin=i0 * plane, end=(i0 + p2) * plane)
        t2 = time.perf_counter() - t0
        extra["s2"] = {"value": p2 * plane * V / t2 / 1e6, "unit": "Mvoxel*views/s", "cores": cores,
                       "sample": f"S2 'solid': {p2} central X-planes x {V} views in {t2:.2f} s"}
        # the "NumPy fallback" BASELINE cfg 1 alludes to (the reference has none): float32, one vectorised pass
        # per view over the whole 128^3 grid, 12 views, single process (oracle/oracle_np.py)
        from oracle import oracle_np
        sh1, o1, vs1, v1 = _sc.make_scene(128, 12, "plant")
        t0 = time.perf_counter()
        lab = oracle_np.carve(sh1, o1, vs1, v1)
        t3 = time.perf_counter() - t0
        extra["numpy_cfg1"] = {"value": int(np.prod(sh1)) * len(v1) / t3 / 1e6, "unit": "Mvoxel*views/s", "cores": 1,
                               "seconds": t3, "sample": "BASELINE cfg 1: 128^3 x 12 views, scene S1, oracle/oracle_np.py "
                               "(NumPy float32, one ufunc per operation)", "carved": int((lab == -1).sum())}
    return {**extra, "value": vv / t / 1e6, "unit": "Mvoxel*views/s", "cores": cores, "kind": "port",
            "sample": f"{planes} central X-planes of the {nx}x{ny}x{nz} grid x {V} views "
                      f"({vv / 1e6:.0f} Mvoxel*views in {t:.2f} s, mean of {reps} run(s)), "
                      f"oracle/spacecarve_oracle.c, int32 masks as cl.py:215, {cores} threads of "
                      f"{avail} visible",
            "value_1thread": p1 * plane * V / t1 / 1e6,
            "sample_1thread": f"{p1} central X-planes x {V} views in {t1:.2f} s, 1 thread"}


def cached_scene(a, scenes, shape):
    """The synthetic scene of the run; its masks (the costly part: 72 splatted silhouettes) are kept as one .npy under
    --scene-cache so that the traffic child passes do not rebuild them."""
    key = "%s_%dx%dx%d_%d_%s.npy" % (a.scene_cache, shape[0], shape[1], shape[2], a.views, a.scene)
    if os.path.exists(key):
        try:
            stack = np.load(key)
            gshape, origin, vs, poses, _, _ = scenes.scene_poses(tuple(shape), a.views, a.scene)
            if stack.shape == (a.views, scenes.HEIGHT, scenes.WIDTH) and stack.dtype == np.uint8:
                return gshape, origin, vs, [(K, R, t, stack[q]) for q, (K, R, t) in enumerate(poses)]
        except Exception:
            pass
    gshape, origin, vs, views = scenes.make_scene(tuple(shape), a.views, a.scene)
    try:
        tmp = key + ".%d.tmp.npy" % os.getpid()
        np.save(tmp, np.stack([m for _, _, _, m in views]))
        os.replace(tmp, key)
    except Exception:
        pass
    return gshape, origin, vs, views


def traffic_passes(a):
    """roofline.traffic measured by THIS invocation: two short child runs of this script under rocprofv3, one per PMC
    counter (separate passes, kernel trace only beside them -- /opt/skills/guides/MI355X_MICROARCH.md, HBM), started
    before this process initialises the GPU.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE counts the
    wide (16 B per lane) streaming reads of the pack kernel at half their bytes on gfx950 (x2 there; the riders'
    reads inside carve_brick_kernel stay raw: the sum understates a batch by <= 50 MB).  Returns a dict or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = {"per_kernel": {}, "source": "this run: child passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` of "
                                       "bench.py --steps 3 --warmup 1 (same box, same process tree).  KNOWN HOLE: FETCH_SIZE "
                                       "counts 16-byte-per-lane streaming reads at half their bytes on gfx950; the pack "
                                       "kernel's are doubled here, the riders' (the same reads, inside carve_brick_kernel "
                                       "beside other loads) are not -- of 112 MB of mask bytes about 41 MB go uncounted, so "
                                       "the true figure is ~1.06 x this one"}
    work = tempfile.mkdtemp(prefix="sc_traffic_", dir=os.environ.get("TMPDIR", "/tmp"))
    sums = {}
    batches = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1",
                   "--cpu-seconds", "0", "--extra-steps", "0", "--e2e-reps", "0", "--skip-other-path", "--cold-reps", "0",
                   "--traffic-passes", "off", "--parity-check", "off", "--cold-process", "off", "--n", str(a.n), "--views", str(a.views), "--scene", a.scene,
                   "--scene-cache", a.scene_cache]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            r = subprocess.run(cmd, cwd=env["TMPDIR"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name", counter) != counter:
                        continue
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                    per.setdefault(name, []).append(float(row["Counter_Value"]))
            nb = len(per.get("carve_list_kernel<true, 2>", []) or per.get("carve_special_kernel", []))
            if nb == 0:
                return None
            batches[counter] = nb
            for name, vals in per.items():
                if name.startswith("fill_kernel") or name.startswith("__amd") or "selftest" in name:
                    continue
                corr = 2.0 if (counter == "FETCH_SIZE" and name.startswith("pack")) else 1.0
                b = sum(vals) * 1024.0 * corr / nb
                sums[counter] = sums.get(counter, 0.0) + b
                out["per_kernel"].setdefault(name, {})[counter + "_bytes_per_batch"] = b
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if "FETCH_SIZE" not in sums or "WRITE_SIZE" not in sums:
        return None
    out["hbm_bytes_per_launch"] = sums["FETCH_SIZE"] + sums["WRITE_SIZE"]
    out["read_bytes"] = sums["FETCH_SIZE"]
    out["write_bytes"] = sums["WRITE_SIZE"]
    out["batches_per_pass"] = batches
    return out


def cold_first_batch(a, nat, shape, origin, vs, call, device, reps):
    """What a Voxels run pays (tasks/cl.py:162-165 of the reference: ONE batch on a fresh engine): sc_create -> the
    72 resident masks enqueued -> flush -> synchronize, the engine's one-off allocations (label volume, survivor
    lists, control block, packed-mask arena) and every kernel of the batch included; host clock.  The same engine's
    second batch beside it (clear + batch + synchronize, host clock): what is left when nothing is allocated."""
    K, R, t, masks_dev, V, H, W = call
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng = nat.Engine(list(shape), origin, vs, nat.SC_MODE_CARVE, device=device)
        t1 = time.perf_counter()
        eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
        eng.flush()
        t2 = time.perf_counter()
        eng.synchronize()
        t3 = time.perf_counter()
        eng.clear()
        eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
        eng.flush()
        eng.synchronize()
        t4 = time.perf_counter()
        # device time of a first batch by itself: a third batch with an event pair around every kernel
        eng.set_option(nat.SC_OPT_TIME_KERNELS, 1)
        eng.clear()
        eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
        eng.flush()
        eng.synchronize()
        ksum = 0.0
        for kid in (nat.SC_KERNEL_PACK, nat.SC_KERNEL_FLAGS, nat.SC_KERNEL_CARVE, nat.SC_KERNEL_LIST):
            _, ms = eng.kernel_stats(kid)
            ksum += ms
        eng.close()
        runs.append({"total_ms": (t3 - t0) * 1e3, "create_ms": (t1 - t0) * 1e3, "enqueue_ms": (t2 - t1) * 1e3,
                     "wait_ms": (t3 - t2) * 1e3, "second_batch_ms": (t4 - t3) * 1e3, "kernels_ms": ksum})
    first = runs[0]
    return {"cold_first_batch_ms": first["total_ms"], "breakdown": first, "all_total_ms": [r["total_ms"] for r in runs],
            "note": "fresh sc_create -> 72 resident masks enqueued -> flush -> synchronize, host clock: the FIRST such engine "
                    "of %d (round 5: no best-of -- the later ones reuse what the first one's hipFree left in the runtime's "
                    "pools and read 3 x lower; the first engine of a PROCESS is `cold_process`); create_ms = the 512 MiB "
                    "label volume and a stream; enqueue_ms = the engine's one-off allocations (survivor lists, control "
                    "block, packed-mask arena) interleaved with its launches; wait_ms = what was left of the device work "
                    "when the host was through; second_batch_ms = clear + the same batch + synchronize on that engine; "
                    "kernels_ms = HIP events around every kernel of a batch (the bulk decision is taken on the device "
                    "inside each batch: a first batch runs the kernels of every later one)" % reps}


def host_timed(engine, torch, fn, steps, warmup=1, runs=3):
    """Host clock around `steps` calls of fn() with a synchronize on both sides (ms per step), the best of
    `runs` such runs: these are the extra legs of the line (other scenes, averaging), a few milliseconds of
    device time each, and one stall of the box (seen once: 40 ms while the previous engine's gigabyte went back
    to the driver) would otherwise be the figure.  `value` itself is timed once over exactly --steps steps."""
    for _ in range(warmup):
        fn()
    best = float("inf")
    for _ in range(max(1, runs)):
        engine.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        engine.synchronize()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


def extra_scenes(a, nat, torch, eng, shape, masks_dev, steps):
    """The fused carve on the other scenes of SURVEY 8d (+ "dense"), same grid, same engine."""
    from plant3dvision_amd import scenes
    out = {}
    n_local = eng.num_voxels()
    # (3 x --extra-steps per run since round 5: a run carries ~90 us of launch and wake-up latency on the host's
    # clock whatever its length -- 4 % of a 0.2 ms batch over 10 steps, what tools/bench_scenes.py's 40-step runs
    # read lower than this line did)
    steps = 3 * steps
    for kind in ("dense", "solid", "noise"):
        _, _, _, views = scenes.make_scene(tuple(shape), a.views, kind)
        V = len(views)
        H, W = views[0][3].shape
        eng.dev_upload(masks_dev, np.ascontiguousarray(np.stack([m for _, _, _, m in views])))
        K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
        fg = float(np.mean([(m != 0).mean() for _, _, _, m in views]))
        eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)

        def step():
            eng.clear()
            eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
            eng.flush()

        ms = host_timed(eng, torch, step, steps, warmup=2)
        bytes_step = 4.0 * n_local + float(V) * W * H
        out[kind] = {"ms_per_step": ms, "value": n_local * V / ms / 1e3, "unit": "Mvoxel*views/s", "steps": steps,
                     "timing": "best of 3 runs of that many steps",
                     "mask_foreground": fg,
                     "roofline_frac_hbm": bytes_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "fused_counts": scene_counts(eng)}
    # the reference's own configuration (configs/test_geom_pipe_real.toml:27-36 -> 301 x 301 x 561 voxels,
    # the 60 views of tests/testdata/real_plant): an unaligned grid, a third of it seen by no view
    shape_l, origin_l, vs_l, views_l = scenes.literal_real_plant_scene(60, "plant")
    lit = nat.Engine(shape_l, origin_l, vs_l, nat.SC_MODE_CARVE, device=eng.device)
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views_l]))
    ptr = lit.dev_alloc(stack.nbytes)
    lit.dev_upload(ptr, stack)
    K = np.stack([v[0] for v in views_l]); R = np.stack([v[1] for v in views_l]); t = np.stack([v[2] for v in views_l])
    Vl, Hl, Wl = stack.shape

    def step_l():
        lit.clear()
        lit.process_views_device(K, R, t, ptr, Vl, Hl, Wl, nat.SC_MASK_U8)
        lit.flush()

    ms = host_timed(lit, torch, step_l, steps, warmup=2)
    nl = lit.num_voxels()
    out["literal_301x301x561_60"] = {
        "ms_per_step": ms, "value": nl * Vl / ms / 1e3, "unit": "Mvoxel*views/s", "steps": steps,
        "timing": "best of 3 runs of that many steps",
        "roofline_frac_hbm": (4.0 * nl + float(Vl) * Wl * Hl) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "fused_counts": scene_counts(lit),
        "note": "the reference's literal test configuration; 32 % of the grid is seen by no view"}
    lit.dev_free(ptr)
    lit.close()
    return out


def scene_counts(eng):
    """What the last fused launch left at each stage, incl. the bulk units asked about as a whole and the
    work items their undecided views became."""
    c = eng.fused_counts_ex()
    return c


def average_forms(a, nat, torch, shape, origin, vs, views, device, steps):
    """The `average` kernel (backprojection.c:36-55) on the three mask forms the host hands over
    (cl.py:205-215): uint8 binary and uint8 grey (bytes + 256-entry table, SC_MASK_U8_LUT) and
    float32 (the binary masks converted, and random grey values).  Bound: VALU issue -- every in-image voxel.view costs the projection (~50 lane-ops,
    SURVEY 8d) whatever the mask holds; HBM traffic is 4 B per voxel once."""
    from plant3dvision_amd.cl import averaging_table, img_as_float32
    eng = nat.Engine(shape, origin, vs, nat.SC_MODE_AVERAGE, device=device)
    eng.set_lut(averaging_table(False))
    V = len(views)
    H, W = views[0][3].shape
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    binary = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    grey = np.random.default_rng(4321).integers(0, 256, binary.shape, dtype=np.uint8)
    buf = eng.dev_alloc(binary.size * 4)
    n = eng.num_voxels()
    out = {}
    for name, data, code in (("u8_binary", binary, nat.SC_MASK_U8_LUT), ("u8_grey", grey, nat.SC_MASK_U8_LUT),
                             ("f32", img_as_float32(binary), nat.SC_MASK_F32),
                             ("f32_grey", img_as_float32(grey), nat.SC_MASK_F32)):
        eng.dev_upload(buf, data)

        def step():
            eng.clear()
            eng.process_views_device(K, R, t, buf, V, H, W, code)
            eng.flush()

        ms = host_timed(eng, torch, step, steps, warmup=1)
        ent = {"ms_per_step": ms, "value": n * V / ms / 1e3, "unit": "Mvoxel*views/s", "steps": steps,
               "timing": "best of 3 runs of that many steps"}
        ceil = valu_ceiling("u8" if name.startswith("u8") else "f32", float(n) * V)
        if ceil is not None:
            ceiling_ms, lane_ops, ns_vv, mix, stream_ms = ceil
            roof = {"bound": "valu-issue", "ceiling_ms": ceiling_ms, "frac": ceiling_ms / ms,
                    "mix_stream_ms": stream_ms, "frac_of_mix_stream": stream_ms / ms,
                    "lane_ops_per_voxel_view": lane_ops, "ns_per_wavefront_voxel_view": ns_vv,
                    "mix_lane_ops_per_voxel_view": mix, "cost_ns": VALU_COST_NS, "mix_stream_ns_per_instruction": VALU_MIX_STREAM_NS,
                    "model": "the kernel's vector instructions (SQ_INSTS_VALU_* of %s: random grey masks, every (brick, view) "
                             "pair projected) over %d SIMDs.  ceiling_ms: each class priced at the nanoseconds per wavefront "
                             "instruction and SIMD it costs as a stream of its own at 8 waves per SIMD (%s: wall time, no "
                             "assumed clock; instructions no class counter names at the cheapest class) -- a lower bound of "
                             "the time the mix needs.  mix_stream_ms: the whole count at the rate a stream of a projection's "
                             "own mix reaches there (1.591 ns per instruction, 25 %% above the sum of its classes: the "
                             "classes share the clock -- 1.3-1.55 GHz with every SIMD issuing, not 2.4 -- and the issue ports)"
                             % (AVG_COUNTERS["u8" if name.startswith("u8") else "f32"], SIMDS, VALU_PROBE)}
            if name in ("u8_binary", "f32"):
                # flat footprints (all 0 / all 255 under a whole brick) add table[0] / table[255] without
                # projecting: the model above counts work the kernel did not do
                roof["equivalent_frac"] = roof.pop("frac")
                roof["frac"] = None
                roof.pop("frac_of_mix_stream", None)
                roof["note"] = ("brick form skips the projection of (brick, view) pairs with a flat footprint: "
                                "an equivalent rate, not issue utilisation")
            ent["roofline"] = roof
        out[name] = ent
    eng.dev_free(buf)
    eng.close()
    return out


def _maxed(torch, dist, dt):
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def assembled_steps(nat, torch, dist, sb, eng, call, steps, warmup, bits=2, overlap=True, form="dense", twin=None):
    """`steps` steps of carve + assembly, barrier + synchronize on both sides, MAX over ranks (seconds).  A step:
    clear, the batch of resident masks, pack the labels, all-gather (RCCL over xGMI) -- every rank ends every step
    holding the whole grid in its packed form (what proc3d.vol2pcd reads).
    form "sparse" (round 6, the headline at N > 1): one code per 16 x 64-voxel brick + the 2-bit labels of the mixed
    bricks only, packed from the batch's verdict bytes and live list (sc_values_sparse), the collective enqueued by
    the library itself (sc_all_gather_sparse over the communicator of sc_comm_create; gloo rehearsals: torch); the
    headers of step k -- did every rank's mixed bricks fit the capacity all ranks sent with? -- are read while step
    k + 1 runs, and a step that did not fit fails the leg (the capacity is settled before the timed steps).
    form "dense": `bits` per label for every voxel (rounds 3-5).
    overlap: the collective of step k runs beside the carve of step k + 1 (receive buffers alternate); the last
    collective is waited for inside the timed region.
    twin (form "sparse"): a second engine on the same planes (ShardedBackprojection.twin) -- the two take the steps in
    turn, so that the PACK of step k (on its engine's stream) is beside the carve of step k + 1 as well: the double
    buffering of a pipeline of scans (every step is still clear + all the views + pack + collective, all waited for
    inside the timed region)."""
    dev = torch.device("cuda", eng.device)
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    if form == "sparse":
        sbs = (sb,) if twin is None else (sb, twin)

        def batch(q):
            q.engine.clear()
            q.engine.process_views_device(*call, nat.SC_MASK_U8)

        for q in sbs:
            q.engine.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
            batch(q)
            q.all_gather(compress="sparse", unpack=False)  # settles the capacity (every rank alike), synchronously
        prev = [None]

        def step(i):
            q = sbs[i % len(sbs)]
            batch(q)
            g = q.all_gather(compress="sparse", unpack=False, overlap=overlap, check=False)
            if prev[0] is not None:
                prev[0].verify()  # the previous step's headers, while this one runs
            prev[0] = g
            return g

        def finish():
            if prev[0] is not None:
                prev[0].verify()
                prev[0] = None
            for q in sbs:
                q.synchronize()
            if sb.comm is not None:
                sb.comm.synchronize()
    else:
        comp = "2bit" if bits == 2 else "1bit"
        recv = [torch.empty(sb.packed_rank_bytes(bits) * sb.world_size, dtype=torch.uint8, device=dev) for _ in range(2)]

        def step(i):
            eng.clear()
            eng.process_views_device(*call, nat.SC_MASK_U8)
            return sb.all_gather(compress=comp, recv=recv[i & 1], unpack=False, overlap=overlap)

        def finish():
            sb.synchronize()          # the engine's stream (behind the last collective)

    for i in range(max(1, warmup)):
        step(i)
    finish()
    torch.cuda.synchronize()
    _barrier(dist)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    finish()
    torch.cuda.synchronize()  # the collectives' stream (torch path)
    _barrier(dist)
    dt = _maxed(torch, dist, time.perf_counter() - t0)
    return dt


def _gather_to_host_legs(res, sb, dist, maxed):
    """gather_to_host timed once per form (the engine's state is whatever the last leg left)."""
    dest = np.zeros(sb.shape, dtype=np.int32) if dist.get_rank() == 0 else None  # (its pages touched: see below)
    sb.gather_to_host(dst=0, out=dest)  # once untimed: the engine's packed buffer, RCCL's first gather
    _barrier(dist)
    t0 = time.perf_counter()
    hostvol = sb.gather_to_host(dst=0, out=dest)
    _barrier(dist)
    res["gather_to_host_ms"] = maxed(time.perf_counter() - t0) * 1e3
    _barrier(dist)
    t0 = time.perf_counter()
    fresh = sb.gather_to_host(dst=0)
    _barrier(dist)
    res["gather_to_host_fresh_array_ms"] = maxed(time.perf_counter() - t0) * 1e3
    sb.gather_to_host(dst=0, compress="sparse", out=dest)
    _barrier(dist)
    t0 = time.perf_counter()
    hostvol = sb.gather_to_host(dst=0, compress="sparse", out=dest)
    _barrier(dist)
    res["gather_to_host_sparse_ms"] = maxed(time.perf_counter() - t0) * 1e3
    res["gather_to_host_note"] = ("labels at 2 bits each to rank 0's GPU over the collective, one PCIe copy of 1/16 of the "
                                  "grid's bytes, widened and interleaved into the int32 grid by the host pool "
                                  "(sc_widen_labels2_ranks); into an array whose pages have been touched -- a fresh "
                                  "np.empty pays first-touch page faults for 4 bytes per voxel (gather_to_host_fresh_array_ms)")
    del hostvol, fresh, dest



def assembly(a, nat, torch, dist, sb, eng, call, steps, n_total, V, host=True):
    """N > 1, beside `value`: the same carve + 2-bit all-gather steps one after the other (no overlap), with the
    kernel that unpacks the grid into global order on every GPU, the occupancy alone (1 bit per label), the int8
    form of round 2; barrier + synchronize on both sides, MAX over ranks; then gather_to_host once."""
    world = sb.world_size
    dev = torch.device("cuda", eng.device)
    n_grid = int(np.prod(sb.shape))

    def maxed(dt):
        return _maxed(torch, dist, dt)

    def run(step, nsteps):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        _barrier(dist)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            full = step()
        sb.synchronize()
        torch.cuda.synchronize()
        _barrier(dist)
        dt = maxed(time.perf_counter() - t0)
        del full
        return dt

    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    res = {"steps": steps}
    dss = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, overlap=False, form="sparse")
    res["sparse_grid_serial"] = {"ms_per_step": dss / steps * 1e3, "steps": steps, "value": n_total * V * steps / dss / 1e6,
                                 "bytes_sent_per_rank": int(sb.sparse_rank_bytes()), "capacity_bricks": int(sb._sparse_cap),
                                 "note": "as `value` (the brick-sparse form), the collective on the engine's own stream: "
                                         "the next carve waits for it"}
    # the same steps + the kernel that writes the int8 grid in global order from the sparse buffers on every GPU
    sgrid = [None]

    def step_su():
        eng.clear()
        eng.process_views_device(*call, nat.SC_MASK_U8)
        sgrid[0] = sb.all_gather(compress="sparse", unpack=False, check=False)
        return sgrid[0].unpack(widen=False, out=out8s[0])

    out8s = [None]
    first = sb.all_gather(compress="sparse", unpack=False).unpack(widen=False)
    out8s[0] = first
    dsu = run(step_su, steps)
    res["sparse_grid_unpacked_int8"] = {"ms_per_step": dsu / steps * 1e3, "steps": steps, "value": n_total * V * steps / dsu / 1e6,
                                        "note": "+ sc_unpack_sparse: the int8 grid in global order on every GPU (1 byte per voxel written)"}
    del first
    out8s[0] = None
    if sb.comm is not None:
        # the dense forms of rounds 3-5 through the same communicator (sc_all_gather_packed), for comparison
        from plant3dvision_amd.sharded import DevMem

        def dense_library(bits, overlap):
            stride = sb.packed_rank_bytes(bits)
            recv = [DevMem(eng, stride * world) for _ in range(2)]

            def step(i):
                eng.clear()
                eng.process_views_device(*call, nat.SC_MASK_U8)
                eng.all_gather_packed(sb.comm, bits, recv[i & 1].ptr, stride, overlap=overlap)

            def fin():
                eng.synchronize()
                sb.comm.synchronize()

            for i in range(2):
                step(i)
            fin()
            _barrier(dist)
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            fin()
            _barrier(dist)
            dt = maxed(time.perf_counter() - t0)
            for r in recv:
                r.free()
            return {"ms_per_step": dt / steps * 1e3, "steps": steps, "value": n_total * V * steps / dt / 1e6,
                    "bytes_sent_per_rank": int(stride)}

        res["packed_grid_serial"] = dict(dense_library(2, False), note="the DENSE 2-bit form of rounds 3-5 (every voxel packed), "
                                         "sc_all_gather_packed on the engine's stream")
        res["packed_grid_overlapped"] = dict(dense_library(2, True), note="the dense 2-bit form, the collective beside the next "
                                             "carve: `value` of round 5")
        res["occupancy_1bit"] = dict(dense_library(1, True), note="only the occupancy `label == 1` travels, 1 bit per voxel")
        if host:
            _gather_to_host_legs(res, sb, dist, maxed)
        return res
    dts = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, bits=2, overlap=False)
    res["packed_grid_serial"] = {"ms_per_step": dts / steps * 1e3, "steps": steps, "value": n_total * V * steps / dts / 1e6,
                                 "note": "the DENSE 2-bit form of rounds 3-5: carve, pack every voxel to 2 bits, all-gather, one "
                                         "after the other"}
    dtd = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, bits=2, overlap=True)
    res["packed_grid_overlapped"] = {"ms_per_step": dtd / steps * 1e3, "steps": steps, "value": n_total * V * steps / dtd / 1e6,
                                     "bytes_sent_per_rank": int(sb.packed_rank_bytes(2)),
                                     "note": "the dense 2-bit form with the collective beside the next carve: `value` of round 5"}
    dt1 = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, bits=1, overlap=True)
    res["occupancy_1bit"] = {"ms_per_step": dt1 / steps * 1e3, "steps": steps, "value": n_total * V * steps / dt1 / 1e6,
                             "bytes_received_per_rank": int(sb.packed_rank_bytes(1) * world),
                             "note": "as `value`, but only the occupancy `label == 1` travels (1 bit per label: what "
                                     "proc3d.vol2pcd binarises to, proc3d.py:515)"}
    recv2 = torch.empty(sb.packed_rank_bytes(2) * world, dtype=torch.uint8, device=dev)
    out8 = torch.empty(n_grid, dtype=torch.int8, device=dev)

    def step2():
        eng.clear()
        eng.process_views_device(*call, nat.SC_MASK_U8)
        return sb.all_gather(compress="2bit", widen=False, recv=recv2, out=out8)

    dt = run(step2, steps)
    res.update({"kind": "all-gather of the labels at 2 bits each + one kernel that unpacks them into global order on every "
                        "GPU (int8 on the device; vol2pcd takes 1-byte volumes)",
                "ms_per_step": dt / steps * 1e3,
                "value_with_unpacked_assembly": n_total * V * steps / dt / 1e6,
                "bytes_received_per_rank": int(recv2.numel())})
    del recv2
    # the int8 wire form (round 2), a few steps, for comparison
    pad = sb._planes_max() * sb.shape[1] * sb.shape[2]
    recv = torch.empty(pad * world, dtype=torch.int8, device=dev)
    out = torch.empty(pad * world, dtype=torch.int8, device=dev)

    def step8():
        eng.clear()
        eng.process_views_device(*call, nat.SC_MASK_U8)
        return sb.all_gather(compress=True, widen=False, recv=recv, out=out)

    n8 = max(2, steps // 4)
    dt8 = run(step8, n8)
    res["int8_wire"] = {"ms_per_step": dt8 / n8 * 1e3, "steps": n8, "bytes_received_per_rank": int(recv.numel())}
    del recv, out, out8
    if host:
        _gather_to_host_legs(res, sb, dist, maxed)
    return res


def strong_scaling(a, nat, torch, dist, scenes, ShardedBackprojection, rank, world, local_rank, steps, comm_library=False, all_ok=None):
    """BASELINE.json's metric literally: ONE 512^3 x 72 grid (cfg 3) split over the N ranks (x-planes dealt
    round-robin), carve only and carve + assembly; barrier + synchronize on both sides, MAX over ranks."""
    gshape, origin, vs, views = scenes.make_scene((a.n,) * 3, a.views, a.scene)
    V = len(views)
    H, W = views[0][3].shape
    sb = ShardedBackprojection(gshape, origin, vs, rank=rank, world_size=world, device=local_rank)
    sb.force_collective = bool(a.rccl_rehearsal)
    eng = sb.engine
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    masks_dev = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(masks_dev, stack)
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
    call = (K, R, t, masks_dev, V, H, W)
    n_total = int(np.prod(gshape))
    run_steps(eng, nat, *call, 3, 0)
    eng.synchronize()
    dt, _ = timed(eng, nat, torch, dist, call, steps, 0, world, time_kernels="span")
    if comm_library:
        try:
            sb.init_comm()
            made = True
        except Exception:  # noqa: BLE001
            made = False
        if not all_ok(made):  # every rank skips the leg together
            eng.dev_free(masks_dev)
            sb.close()
            return {"error": "a rank could not create a second communicator: the strong leg was skipped on every rank"}
    twin = sb.twin() if (a.twin_engine and sb.comm is not None) else None  # two engines in turn, as the headline at N > 1
    dta = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, overlap=True, form="sparse", twin=twin)
    dts = assembled_steps(nat, torch, dist, sb, eng, call, steps, 2, overlap=False, form="sparse")
    if twin is not None:
        twin.close()
    out = {"workload": f"ONE {a.n}^3 x {V} grid split over {world} rank(s), x-planes cyclic ({len(sb.planes)} planes per rank)",
           "value": n_total * V * steps / dta / 1e6, "ms_per_step": dta / steps * 1e3, "steps": steps,
           "value_is": "carve + sparse pack + all-gather per step, the collective beside the next step's carve"
                       + (", two engines taking the steps in turn" if a.twin_engine and sb.comm is not None else "") + " (as the headline at N > 1)",
           "value_carve_only": n_total * V * steps / dt / 1e6, "ms_per_step_carve_only": dt / steps * 1e3,
           "value_serial_assembly": n_total * V * steps / dts / 1e6, "ms_per_step_serial_assembly": dts / steps * 1e3,
           "unit": "Mvoxel*views/s", "scaling": "strong"}
    eng.dev_free(masks_dev)
    sb.close()
    return out


def self_launch(a):
    """`python3 bench.py --gpus N` from a plain shell (no WORLD_SIZE): start the N ranks as fresh child processes of
    this script -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would set -- BEFORE this process makes any GPU
    call (it never makes one), relay rank 0's JSON line on stdout and return the first non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # rank 0's stdout carries the line; the other ranks print nothing there (their stdout joins stderr)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # Rank 0's stdout is drained on a thread while ALL children are polled: the first rank that exits non-zero ends
    # the job for the others at once (what torchrun does) instead of leaving them in a rendezvous or a collective
    # until a watchdog fires; they are this process's own children, ended by their own pids
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0):
                    failed = True
        if failed:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            deadline = time.time() + 10
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    line = None
    for ln in out0.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    if bad:
        own = [c for c in bad if c > 0]  # (negative: a rank this function ended after another one failed)
        return own[0] if own else 1
    return 0 if line is not None else 1


def cold_child(a):
    """The body of `bench.py --cold-child`: what a Voxels run costs the FIRST engine of a process
    (tasks/cl.py:162-165 of the reference) -- import the binding, sc_create, one batch of 72 masks already in HBM,
    synchronize; every part on the host clock, once.  Prints one JSON object."""
    t = [time.perf_counter()]
    from plant3dvision_amd import _native as nat
    from plant3dvision_amd import scenes
    nat.backend()
    t.append(time.perf_counter())  # binding + libspacecarve.so loaded
    shape = global_shape(a.n, 1)
    gshape, origin, vs, views = cached_scene(a, scenes, shape)
    V = len(views)
    H, W = views[0][3].shape
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); tt = np.stack([v[2] for v in views])
    t.append(time.perf_counter())  # scene from the cache
    # the process's first HIP calls: runtime and device initialisation, the context (what ANY use of the GPU pays once)
    import ctypes
    hip = nat.hip_runtime()
    hip.hipInit(0)
    hip.hipSetDevice(0)
    hip.hipFree(ctypes.c_void_p(0))
    hip.hipDeviceSynchronize()
    t.append(time.perf_counter())
    eng = nat.Engine(list(gshape), origin, vs, nat.SC_MODE_CARVE, device=0)
    t.append(time.perf_counter())
    masks_dev = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(masks_dev, stack)
    t.append(time.perf_counter())  # ingest stand-in: 112 MB of masks to HBM (not part of the batch)
    eng.process_views_device(K, R, tt, masks_dev, V, H, W, nat.SC_MASK_U8)
    eng.flush()
    t.append(time.perf_counter())
    eng.synchronize()
    t.append(time.perf_counter())
    eng.clear()
    eng.process_views_device(K, R, tt, masks_dev, V, H, W, nat.SC_MASK_U8)
    eng.flush()
    eng.synchronize()
    t.append(time.perf_counter())
    hist = None
    eng.dev_free(masks_dev)
    eng.close()
    ms = [(b - a_) * 1e3 for a_, b in zip(t[:-1], t[1:])]
    out = {"import_ms": ms[0], "scene_ms": ms[1], "hip_runtime_init_ms": ms[2], "create_ms": ms[3], "mask_upload_ms": ms[4],
           "enqueue_ms": ms[5], "wait_ms": ms[6], "second_batch_ms": ms[7], "first_batch_ms": ms[3] + ms[5] + ms[6]}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def write_gray_png(path, img):
    """8-bit greyscale, non-interlaced PNG with the standard library (what `io.write_image(f, im, 'png')` makes of a
    uint8 mask of tasks/proc2d.py, as far as a reader is concerned)."""
    import struct
    import zlib
    H, W = img.shape
    raw = np.empty((H, W + 1), dtype=np.uint8)
    raw[:, 0] = 0  # filter type 0 on every row
    raw[:, 1:] = img

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw.tobytes(), 1)) + chunk(b"IEND", b""))


def cold_files_child(a):
    """The body of `bench.py --cold-files-child DIR`: a fresh process from its first line to the int32 volume in host
    memory -- import of the drop-in module, the 72 PNG files of DIR (poses in DIR/poses.npz), Backprojection(...),
    process_fileset (cl.py:234-305 of the reference): what a cold `Voxels.run` (tasks/cl.py:100, 162-165) costs."""
    t0 = time.perf_counter()
    from plant3dvision_amd.cl import Backprojection
    from plant3dvision_amd import scenes
    t1 = time.perf_counter()
    meta = np.load(os.path.join(a.cold_files_child, "poses.npz"))
    shape, origin, vs = [int(x) for x in meta["shape"]], [float(x) for x in meta["origin"]], float(meta["vs"])

    class PngFile:
        def __init__(self, fid, path, md):
            self.id, self.path, self._md = fid, path, md

        def get_metadata(self, key=None, default=None):
            return self._md if key is None else self._md.get(key, default)

        def read_raw(self):  # plantdb's File.read_raw(): the bytes io.read_image decodes
            with open(self.path, "rb") as f:
                return f.read()

    files = [PngFile("%05d_mask" % q, os.path.join(a.cold_files_child, "%05d_mask.png" % q),
                     {"colmap_camera": scenes.camera_dict(meta["K"][q], meta["R"][q], meta["t"][q])})
             for q in range(len(meta["K"]))]
    t2 = time.perf_counter()
    bp = Backprojection(shape, origin, vs)
    t3 = time.perf_counter()
    vol = bp.process_fileset(files, "colmap_camera")
    t4 = time.perf_counter()
    hist = [int((vol == -1).sum()), int((vol == 0).sum()), int((vol == 1).sum())]
    setup_ms, waited_ms = bp._engine.setup_times()
    # the same files again through the warm engine: reads + decode + carve + read-back without any set-up
    bp.clear()
    t5 = time.perf_counter()
    bp.process_fileset(files, "colmap_camera")
    t6 = time.perf_counter()
    bp.close()
    out = {"import_ms": (t1 - t0) * 1e3, "file_list_ms": (t2 - t1) * 1e3, "constructor_ms": (t3 - t2) * 1e3,
           "process_fileset_ms": (t4 - t3) * 1e3, "files_to_volume_ms": (t4 - t0) * 1e3,
           "warm_files_to_volume_ms": (t6 - t5) * 1e3, "labels_histogram": hist,
           "device_setup_ms": setup_ms, "waited_for_setup_ms": waited_ms,
           "process_fileset_minus_wait_ms": (t4 - t3) * 1e3 - waited_ms}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def cold_files(a, views, shape, origin, vs):
    """`files_to_volume_ms` of a fresh process, with the engine's device half deferred (the default) and, for comparison,
    all inside the constructor (SC_ASYNC_CREATE=0); the parent writes the PNG files first and never touches the GPU."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="sc_cold_files_")
    try:
        for q, (K, R, t, m) in enumerate(views):
            write_gray_png(os.path.join(tmp, "%05d_mask.png" % q), m)
        np.savez(os.path.join(tmp, "poses.npz"), K=np.stack([v[0] for v in views]), R=np.stack([v[1] for v in views]),
                 t=np.stack([v[2] for v in views]), shape=np.array(shape), origin=np.array(origin, dtype=np.float64), vs=float(vs))
        res = {}
        for name, flag in (("deferred", "1"), ("all_in_constructor", "0")):
            env = dict(os.environ, SC_ASYNC_CREATE=flag)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-files-child", tmp], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            if r.returncode != 0:
                res[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
                continue
            for ln in r.stdout.decode(errors="replace").splitlines():
                if ln.startswith("{"):
                    res[name] = json.loads(ln)
        return res
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cold_process(a):
    """Runs `bench.py --cold-child` as a fresh process (this one has not touched the GPU yet) with the library's
    allocation trace on; returns its figures + the allocations of a millisecond or more."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cold-child", "--n", str(a.n), "--views", str(a.views),
           "--scene", a.scene, "--scene-cache", a.scene_cache]
    env = dict(os.environ, SC_TRACE_ALLOC="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}
    if r.returncode != 0:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
    out = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{"):
            out = json.loads(ln)
    if out is None:
        return {"error": "no figures from the child"}
    allocs = []
    for ln in r.stderr.decode(errors="replace").splitlines():
        if ln.startswith("sc_alloc "):
            _, what, nbytes, ms_ = ln.split()[:4]
            allocs.append({"what": what, "bytes": int(nbytes), "ms": float(ms_)})
    out["allocations_ms_total"] = sum(x["ms"] for x in allocs)
    out["allocations_over_1ms"] = [x for x in allocs if x["ms"] >= 1.0]
    out["allocations"] = len(allocs)
    out["note"] = ("a fresh process, started before the bench process touched the GPU, once (no best-of): import_ms = "
                   "binding + libspacecarve.so; hip_runtime_init_ms = hipInit + the device's context (any use of the GPU pays "
                   "it once per process); create_ms = sc_create (the code object's load, the label volume, a stream); mask_upload_ms = 112 MB of "
                   "masks to HBM (the ingest stand-in, not the batch); enqueue_ms = the batch's launches with the engine's "
                   "one-off allocations between them and each kernel's first-launch set-up in the runtime; wait_ms = what was left of the device work; first_batch_ms = create + "
                   "enqueue + wait; allocations* = the library's own hipMalloc / hipHostMalloc calls (SC_TRACE_ALLOC)")
    return out


def parity_check(a, nat, eng, call, shape, origin, vs, views, planes=None, rank=0, world=1):
    """A correctness statement for the line the driver archives ("bit-identity check ... on every
    run"), OUTSIDE every timed region; the oracle is the checker here.
      * the fused batch's labels and the labels of the reference's cadence (one launch per view, file order,
        cl.py:223-226) have the same SHA-256;
      * that SHA-256 equals the ORACLE's, committed in tests/golden/synthetic_digests.json
        (tests/golden/make_golden.py big: oracle/spacecarve_oracle.c over the whole grid, or over a rank's planes);
      * with --cpu-seconds > 0 (or a grid of at most 2^24 voxels) the oracle itself runs here over EVERY voxel this rank owns and the labels are
        compared element by element (`oracle_whole_grid`; 0.3 .. 5 s of host threads by scene);
      * the histogram (-1 / 0 / 1).
    `planes`: the global x indices of this rank's planes (N > 1)."""
    import hashlib
    from oracle import oracle_c
    K, R, t, masks_dev, V, H, W = call
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    eng.clear()
    eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
    fused = eng.get_values()
    dig = hashlib.sha256(fused.tobytes()).hexdigest()
    hist = [int((fused == -1).sum()), int((fused == 0).sum()), int((fused == 1).sum())]
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 1)
    eng.set_option(nat.SC_OPT_VIEW_ORDER, 0)
    eng.clear()
    eng.process_views_device(K, R, t, masks_dev, V, H, W, nat.SC_MASK_U8)
    per_view = eng.get_values()
    dig1 = hashlib.sha256(per_view.tobytes()).hexdigest()
    del per_view
    eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, 0)
    eng.set_option(nat.SC_OPT_VIEW_ORDER, 1)
    # the oracle's digest of this very workload, committed with the tests
    key = f"{a.scene}_{shape[0]}_{V}" if world == 1 and list(shape) == [shape[0]] * 3 else \
        f"{a.scene}_{shape[0]}_{V}_cyclic_rank{rank}of{world}"
    committed = None
    try:
        committed = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_digests.json"))).get(key)
    except Exception:  # noqa: BLE001
        committed = None
    dig_ok = None if committed is None else committed["sha256_int32"] == dig
    # ... and the oracle itself, every voxel of this rank
    whole = None
    oracle_s = None
    if a.cpu_seconds > 0 or fused.size <= 1 << 24:  # (small grids: always, it is milliseconds)
        nthreads = min(32, os.cpu_count() or 8)
        t0 = time.perf_counter()
        if planes is None:
            want = oracle_c.carve(list(shape), origin, vs, views, nthreads=nthreads)
        else:
            pl = range(planes.start, planes.stop, planes.step) if isinstance(planes, range) else None
            if pl is None:
                raise ValueError("planes must be a range")
            want = oracle_c.carve_planes(list(shape), origin, vs, views, pl.start, pl.step, len(pl), nthreads=nthreads)
        oracle_s = time.perf_counter() - t0
        whole = bool(np.array_equal(fused.reshape(want.shape), want))
        del want
    ok_all = dig == dig1 and dig_ok is not False and whole is not False
    return {"ok": bool(ok_all), "fused_sha256": dig, "per_view_sha256": dig1, "fused_equals_per_view": dig == dig1,
            "oracle_digest_key": key if committed is not None else None,
            "fused_equals_committed_oracle_digest": dig_ok,
            "oracle_whole_grid": whole, "oracle_voxels": int(fused.size) if whole is not None else 0,
            "oracle_seconds": oracle_s,
            "labels_histogram": hist,
            "note": "outside the timed regions: SHA-256 of the fused batch's int32 labels == that of one launch per view in "
                    "file order (cl.py:223-226) == the oracle's digest committed in tests/golden/synthetic_digests.json; "
                    "oracle_whole_grid: the C oracle (oracle/spacecarve_oracle.c, backprojection.c:57-84 restated) run here "
                    "over every voxel of this rank and compared element by element; histogram of -1 / 0 / 1"}



def e2e_host(a, shape, origin, vs, views, device, reps):
    """The reference's real interface (cl.py:190-232): V uint8 masks in HOST memory, one ``process_view`` each,
    then ``get_values``: int32 labels in HOST memory -- mask upload and label read-back over PCIe included
    (SURVEY 8d ``t_e2e``).  Through the drop-in class, plant-3d-vision_amd/cl.py."""
    from plant3dvision_amd.cl import Backprojection
    bp = Backprojection(list(shape), origin, vs, device=device)
    V = len(views)

    def run():
        bp.clear()
        for K, R, t, m in views:
            bp.process_view(K, R, t, m)
        return bp.get_values()

    vol = run()  # warm-up: allocations, the pinned ring, touched pages
    hist = [int((vol == -1).sum()), int((vol == 0).sum()), int((vol == 1).sum())]
    ts = []
    for _ in range(reps):
        bp.recycle(vol)
        del vol
        t0 = time.perf_counter()
        vol = run()
        ts.append(time.perf_counter() - t0)
    bp.close()
    n = int(np.prod(shape))
    best = min(ts)
    med = float(np.median(ts))
    return {"ms": med * 1e3, "ms_best": best * 1e3, "ms_all": [x * 1e3 for x in ts], "value": n * V / med / 1e6,
            "unit": "Mvoxel*views/s", "reps": reps, "labels_histogram": hist, "ms_is": "median of ms_all",
            "note": f"{V} uint8 masks in host memory -> Backprojection.process_view x {V} -> get_values(): int32 "
                    f"[{shape[0]}][{shape[1]}][{shape[2]}] in host memory; PCIe both ways (labels cross at 2 bits "
                    f"each, in pieces, and are widened by host threads inside the library as they land); median of {reps}"}


def main():
    a = parse()
    if a.cold_child:
        cold_child(a)
        return
    if a.cold_files_child:
        cold_files_child(a)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))  # the ranks are fresh child processes; this one never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # roofline.traffic measured in this invocation: child passes under rocprofv3, BEFORE this process touches the GPU
    # (a GPU-initialised process must not start other programs on these boxes)
    live_traffic = None
    profiled = any(k in os.environ for k in ("ROCPROFILER_TOOL_LIBRARIES", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
        "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and rank == 0 and a.path == "fused" and not a.rccl_rehearsal and \
            (a.traffic_passes == "on" or (a.traffic_passes == "auto" and not profiled)):
        sys.path.insert(0, ROOT)
        from plant3dvision_amd import scenes as _scenes
        cached_scene(a, _scenes, global_shape(a.n, 1))  # built once, the children load it
        live_traffic = traffic_passes(a)
    # a cold Voxels run measured where it happens: a fresh process, before this one touches the GPU
    cold_proc = None
    if world == 1 and rank == 0 and a.path == "fused" and not a.rccl_rehearsal and not profiled and \
            (a.cold_process == "on" or (a.cold_process == "auto" and a.cold_reps > 0)):
        from plant3dvision_amd import scenes as _scenes
        _g, _o, _v, _views = cached_scene(a, _scenes, global_shape(a.n, 1))
        cold_proc = cold_process(a)
        cf = cold_files(a, _views, _g, _o, _v)
        cold_proc["files_to_volume"] = cf
        if isinstance(cf.get("deferred"), dict) and "files_to_volume_ms" in cf["deferred"]:
            cold_proc["files_to_volume_ms"] = cf["deferred"]["files_to_volume_ms"]
        cold_proc["files_to_volume_note"] = (
            "a fresh process, its first line to the int32 volume in host memory: import of the drop-in module, 72 PNG files, "
            "Backprojection(...), process_fileset (cl.py:234-305).  `deferred` (the default): the engine's device half -- "
            "hip_runtime_init_ms + create_ms above -- runs on a thread of the library's from the constructor on "
            "(sc_create_ex, SC_CREATE_DEFERRED) and the files are read and decoded beside it; `all_in_constructor`: "
            "SC_ASYNC_CREATE=0, the constructor waits for all of it (rounds 1-5); warm_files_to_volume_ms: the same files "
            "again through the engine that is up (reads + decode + carve + read-back); device_setup_ms: what the engine's "
            "device half took (runtime initialisation, the first stream, the state); waited_for_setup_ms: how long "
            "process_fileset -- its files read and decoded by then -- waited for it; process_fileset_minus_wait_ms: what of "
            "process_fileset was NOT hidden behind the set-up (the copy of the bits, the carve, the read-back)")
        del _views
    # ONE JSON line on stdout: RCCL prints a version banner to stdout when its communicator comes up, so
    # everything but that line (libraries included, file descriptor 1) goes to stderr from here on
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    global COLLECTIVE, BARRIER, COMM_STUCK
    collective = COLLECTIVE = world > 1 or a.rccl_rehearsal
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.dist_backend == "torch-nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            # "nccl": the library's own RCCL communicator carries the data (sc_comm_create below); torch.distributed is
            # the control plane only -- barriers, the MAX over ranks, the 128-byte id -- and a gloo group is enough for
            # that.  (Measured: with torch's NCCL group alive in the process the same carve + gather steps take 0.197 ms
            # instead of 0.177 -- its streams and watchdog share the hardware queues with the engine's and the
            # communicator's.)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from plant3dvision_amd import _native as nat
    from plant3dvision_amd import scenes
    from plant3dvision_amd.sharded import ShardedBackprojection

    shape = global_shape(a.n, world)
    if world > 1:
        # 72 splatted silhouettes of a 1024^3 scene take the host a while: rank 0 builds them, the others load its file
        if rank == 0:
            cached_scene(a, scenes, shape)
        _barrier(dist)
    gshape, origin, vs, views = cached_scene(a, scenes, shape)
    V = len(views)
    H, W = views[0][3].shape
    sb = ShardedBackprojection(gshape, origin, vs, rank=rank, world_size=world, device=local_rank)
    sb.force_collective = bool(a.rccl_rehearsal)
    eng = sb.engine
    comm_library = collective and a.dist_backend == "nccl"  # RCCL bound by the library itself (sc_comm_*); gloo: torch
    comm_error = None
    all_ok = None
    if comm_library:
        # The library's communicator has never met eight GPUs: if librccl cannot be opened or ncclCommInitRank fails on
        # ANY rank, EVERY rank falls back together (a MIN over the control plane after each step) to the collectives of
        # torch.distributed's gloo group, staged through the hosts -- slower, still correct, and the line says which
        # transport carried the data (`assembly_transport`).
        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        try:
            opened = nat.Comm.available()  # (librccl opens on every rank; rank 0 makes the id inside init_comm)
        except Exception as ex:  # noqa: BLE001
            opened, comm_error = False, repr(ex)
        if all_ok(opened):
            # ncclCommInitRank blocks inside the library: it runs on a thread of its own so that a bootstrap that never
            # completes (a rank that cannot reach its peers) becomes a fallback of ALL ranks after --comm-timeout seconds,
            # not a job that hangs until the launcher's limit.  A thread stuck in there is left behind (daemon) and the
            # process leaves through os._exit at the end.
            import threading
            box = {}

            def make():
                try:
                    sb.init_comm()
                    box["made"] = True
                except Exception as ex:  # noqa: BLE001
                    box["error"] = repr(ex)
                if box.get("abandoned"):  # came back after the ranks had fallen back: nobody may find it
                    box["late"], sb.comm = sb.comm, None

            th = threading.Thread(target=make, daemon=True, name="sc-comm-init")
            th.start()
            th.join(a.comm_timeout)
            if th.is_alive():
                box["abandoned"] = True
                made, comm_error, COMM_STUCK = False, f"sc_comm_create did not return within {a.comm_timeout:.0f} s", True
            else:
                made, comm_error = bool(box.get("made")), box.get("error")
            if all_ok(made):
                BARRIER = sb.comm.barrier
            else:
                if not all_ok(not COMM_STUCK):
                    COMM_STUCK = True  # some rank is still inside: a destroy on the others would wait for it
                if sb.comm is not None and not COMM_STUCK:
                    sb.comm.close()
                sb.comm = None
                comm_library = False
                comm_error = comm_error or "another rank could not create the communicator"
        else:
            comm_library = False
            comm_error = comm_error or "another rank could not open librccl"
    n_local = eng.num_voxels()
    n_total = int(np.prod(gshape))
    stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
    masks_dev = eng.dev_alloc(stack.nbytes)
    eng.dev_upload(masks_dev, stack)
    K = np.stack([v[0] for v in views])
    R = np.stack([v[1] for v in views])
    t = np.stack([v[2] for v in views])
    call = (K, R, t, masks_dev, V, H, W)

    for kv in a.opt:
        k, val = kv.split("=")
        eng.set_option(int(k), int(val))
    vpl = {"fused": 0, "stream": 1}
    other = "stream" if a.path == "fused" else "fused"
    if a.path == "stream":
        eng.set_option(nat.SC_OPT_VIEW_BRICK, 0)  # the streaming kernel, not the brick form of a one-view launch
    # The other schedules run FIRST: a rocprofv3 trace of a long run (profiles/r04_clock_ramp.txt) shows every
    # kernel of the batch ~6 % slower during the first ~15 ms of device activity after an idle spell (scene
    # building on the host) than in the sustained state; the headline's W warm-up + K timed steps follow these
    # legs' device work, so `value` is the sustained rate whatever K is.  (--skip-other-path: no such lead-in.)
    t_legs0 = time.perf_counter()
    res_other = per_view = None
    eng.set_option(nat.SC_OPT_VIEW_BRICK, 0)  # "stream" is the streaming kernel: every view reads the whole state
    if not a.skip_other_path:
        osteps = max(4, a.steps // 4) if other == "stream" else a.steps
        run_steps(eng, nat, *call, 1, vpl[other])
        eng.synchronize()
        dto, statso = timed(eng, nat, torch, dist, call, osteps, vpl[other], world)
        res_other = (dto, statso, osteps)
        # the same cadence (one launch per view) in its brick form: dead bricks are skipped
        eng.set_option(nat.SC_OPT_VIEW_BRICK, 1)
        psteps = max(2, a.steps // 4)
        run_steps(eng, nat, *call, 1, 1)
        eng.synchronize()
        dtp, _ = timed(eng, nat, torch, dist, call, psteps, 1, world)
        _, pvk = timed(eng, nat, torch, dist, call, 2, 1, world, time_kernels=1)  # events around every kernel
        per_view = {"value": n_total * V * psteps / dtp / 1e6, "unit": "Mvoxel*views/s", "steps": psteps,
                    "ms_per_step": dtp / psteps * 1e3, "launches_per_step": 2 * V,
                    "device_ms_per_step": sum(pvk[k]["total_ms"] for k in ("carve", "flags", "pack", "fill")) / 2,
                    "note": "one launch (+ its verdict kernel) per view, the reference's cadence cl.py:223-226, brick "
                            "verdicts and dead-brick skipping (SC_OPT_VIEW_BRICK 1, the default)"}
    eng.set_option(nat.SC_OPT_VIEW_BRICK, 0 if a.path == "stream" else 1)
    t_device_before = time.perf_counter() - t_legs0

    # warmup (untimed), in the timing mode of the timed steps: the first batches with event pairs create
    # their HIP events (tens of microseconds each), which is warm-up work, not a step's
    eng.set_option(nat.SC_OPT_TIME_KERNELS, 2)
    # set-up, not a step: the engine allocates its working buffers (survivor lists, brick tables, packed-mask
    # arena: ~0.5 GB of hipMalloc) the first time a batch goes through; like the upload of the masks above this
    # happens once per engine, whatever --warmup says
    run_steps(eng, nat, *call, 1, vpl[a.path])
    eng.synchronize()
    run_steps(eng, nat, *call, a.warmup, vpl[a.path])
    eng.synchronize()
    dt, stats = timed(eng, nat, torch, dist, call, a.steps, vpl[a.path], world,
                      time_kernels="span" if a.path == "fused" else 2)
    # per-kernel breakdown: a separate short pass with events around every kernel
    bsteps = max(2, min(a.steps, 5))
    dtb, breakdown = timed(eng, nat, torch, dist, call, bsteps, vpl[a.path], world, time_kernels=1)
    breakdown["ms_per_step_with_all_events"] = dtb / bsteps * 1e3
    if a.path == "fused":
        live, s0, s1n, ovf = eng.fused_counts()
        breakdown["fused_counts"] = {"live_bricks": live, "alive_after_dense_stage": s0,
                                     "alive_after_first_list_stage": s1n, "list_overflow": ovf}
    # Throughput with two scans in flight (N = 1): a second engine of the same grid, the two taking the steps in turn on
    # their own streams and label volumes -- a queue of scans, as a service would run them.  Reported BESIDE `value`
    # (which stays one engine, one batch at a time: the quantity every round has measured).
    pipelined = None
    if world == 1 and not a.rccl_rehearsal and a.path == "fused" and a.pipelined_steps > 0:
        try:  # (N = 1, no collective anywhere near: a failure here is reported, not fatal to the line)
            import hashlib
            eng2 = nat.Engine(gshape, origin, vs, nat.SC_MODE_CARVE, device=local_rank)
            pair = (eng, eng2)
            for q in pair:
                run_steps(q, nat, *call, 2, 0)
                q.synchronize()
            psteps = int(a.pipelined_steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(psteps):
                q = pair[i & 1]
                q.clear()
                q.process_views_device(*call, nat.SC_MASK_U8)
                q.flush()
            for q in pair:
                q.synchronize()
            dtq = time.perf_counter() - t0
            same = hashlib.sha256(eng2.get_values().tobytes()).hexdigest() == hashlib.sha256(eng.get_values().tobytes()).hexdigest()
            eng2.close()
            pipelined = {"engines": 2, "steps": psteps, "ms_per_step": dtq / psteps * 1e3, "value": n_total * V * psteps / dtq / 1e6,
                         "unit": "Mvoxel*views/s", "frac_of_hbm_roofline": (4.0 * n_local + float(V) * W * H) / (dtq / psteps) / 8e12,
                         "labels_equal": bool(same),
                         "note": "the same steps (clear + 72 resident masks + fused carve) dealt in turn to TWO engines of the same grid "
                                 "(two label volumes, two streams), host clock around all of them, everything waited for inside: the "
                                 "kernels of one batch run in the other's kernel boundaries and tails.  Throughput of a queue of "
                                 "scans; a single scan's latency is `ms_per_step`.  `value` and `roofline` above are one engine"}
        except Exception as ex:  # noqa: BLE001
            pipelined = {"error": repr(ex)}
    # N > 1: `value` = carve + assembly (t_device + collective), W warm-up and exactly K timed steps
    # These legs are sequences of collectives: an exception on ONE rank must not be swallowed there (its peers would
    # sit in a collective it never joins, and its next leg would pair with their pending one out of phase -- ADVICE
    # r05).  A rank that fails exits non-zero and the launcher (self_launch / torchrun) ends the job.
    asm = None
    dt_asm = None
    twin = None
    twin_check = None
    dt_asm_one = None
    if collective and a.path == "fused":
        # two engines take the steps in turn (ShardedBackprojection.twin: the double buffering of a pipeline of scans):
        # the pack of step k, on its engine's stream, is beside the carve of step k + 1 like the collective
        # (with the library's communicator only: staged through the hosts -- the gloo fallback -- a step is bound by the
        # host and a second engine gains nothing, measured 1.22 against 1.17 ms)
        if a.twin_engine and sb.comm is not None:
            twin = sb.twin()
        dt_asm = assembled_steps(nat, torch, dist, sb, eng, call, a.steps, a.warmup, overlap=True, form="sparse", twin=twin)
        if twin is not None:
            # (outside the timed regions) the twin's labels are the first engine's, which parity_check compares with the oracle
            import hashlib
            twin_check = hashlib.sha256(twin.get_local().tobytes()).hexdigest() == hashlib.sha256(sb.get_local().tobytes()).hexdigest()
            dt_asm_one = assembled_steps(nat, torch, dist, sb, eng, call, a.steps, a.warmup, overlap=True, form="sparse")
            # (the twin is closed at the very end: behind an engine's destruction -- its hipFree / hipHostFree calls -- the
            # small device-to-host copy of the headers on the ENGINE's stream, the serial legs below, took 0.12 ms per step
            # where it takes none before: tools/dbg/sparse_steps.py VARIANTS=ACT, 0.183 -> 0.299 ms)
    if collective and a.assembly_steps > 0:
        asm = assembly(a, nat, torch, dist, sb, eng, call, a.assembly_steps, n_total, V)
        if world == 1:
            asm["rehearsal"] = "process group of one rank on one GPU: the collectives move nothing over xGMI"
    strong = None
    if collective and a.strong_steps > 0 and a.path == "fused":
        strong = strong_scaling(a, nat, torch, dist, scenes, ShardedBackprojection, rank, world, local_rank, a.strong_steps,
                                comm_library=comm_library, all_ok=all_ok if comm_library else None)
    parity = None
    if a.parity_check == "on" and a.path == "fused" and rank == 0:
        parity = parity_check(a, nat, eng, call, gshape, origin, vs, views, planes=None if world == 1 else sb.planes,
                              rank=rank, world=world)
        if twin_check is not None:  # the second engine of the assembled headline carved the same labels
            parity["twin_engine_labels_equal"] = bool(twin_check)
            parity["ok"] = bool(parity["ok"] and twin_check)
    e2e = None
    if world == 1 and not a.rccl_rehearsal and a.e2e_reps > 0 and a.path == "fused":
        e2e = e2e_host(a, gshape, origin, vs, views, local_rank, a.e2e_reps)
    extras = avg = None
    if world == 1 and a.extra_steps > 0 and a.path == "fused":
        extras = extra_scenes(a, nat, torch, eng, gshape, masks_dev, a.extra_steps)
        eng.dev_upload(masks_dev, stack)
        avg = average_forms(a, nat, torch, gshape, origin, vs, views, local_rank, max(2, a.extra_steps // 2))

    cold = None
    if world == 1 and not a.rccl_rehearsal and a.cold_reps > 0 and a.path == "fused":
        cold = cold_first_batch(a, nat, gshape, origin, vs, call, local_rank, a.cold_reps)

    # another assembly on request, once
    gather = None
    if a.gather != "none":
        torch.cuda.synchronize()
        if world > 1:
            _barrier(dist)
        t0 = time.perf_counter()
        if a.gather == "allgather":
            full = sb.all_gather()
        elif a.gather == "allgather8":
            full = sb.all_gather(compress=True)
        else:
            full = sb.all_reduce()
        torch.cuda.synchronize()
        if world > 1:
            _barrier(dist)
        gather = {"kind": a.gather, "ms": (time.perf_counter() - t0) * 1e3,
                  "bytes_out_per_rank": int(full.numel() * full.element_size())}
        del full

    # --- roofline bookkeeping (per launch of the carve kernel, per rank) -------------------
    b_alg_per_vv = (4.0 * n_local * V + 4.0 * n_local + V * W * H) / (n_local * V)
    mask_bits_bytes = ((W + 31) // 32) * ((H + 31) // 32) * 128

    def roof(path, st, traffic, traffic_src=None):
        if path == "fused":
            # One "launch" of the fused schedule is the whole batch: pack16 -> brick_flags ->
            # carve_brick -> brick_confirm -> carve_special -> carve_list<false> -> carve_list<true>.  The timed steps are
            # bracketed by ONE HIP event pair on the engine's stream (sc_span_begin / sc_span_end); a
            # batch's duration is that span over the number of batches in it (they run back to back;
            # an event pair per batch would put a barrier packet between them).  No single kernel of
            # it owns the label write any more (the -1 fill of empty bricks rides with the final list
            # stage), so the roofline is stated for the sequence.
            avg_ms = st["step"]["avg_ms"]
            launches = st["step"]["launches"]
            bytes_launch = 4.0 * n_local + float(V) * W * H
            model = ("fused batch: 4 B/voxel label write (never read) + V*W*H uint8 mask bytes read once; "
                     "the 1-bit tiles and survivor lists are implementation traffic, not counted")
            units = n_local * V
            kernel = "fused batch (pack_band_kernel + brick_flags_kernel + carve_brick_kernel + brick_confirm_kernel + carve_special_kernel + carve_list_kernel x2)"
        else:
            avg_ms = st["carve"]["avg_ms"]
            launches = st["carve"]["launches"]
            bytes_launch = b_alg_per_vv * n_local  # SURVEY 8d per-unit figure x voxel.views/launch
            model = "per-view launch: SURVEY 8d B_alg/(N*V) = %.3f B per voxel.view x N voxel.views" % b_alg_per_vv
            units = n_local
            kernel = "carve_kernel_1<false>"
        ach = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
             "traffic_source": (traffic_src or (os.path.relpath(a.traffic_json, ROOT) + " (builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                "passes, tools/profile_gpu.sh; not measured in this run)")) if traffic is not None else None,
             "kernel": kernel,
             "avg_launch_ms": avg_ms, "launches": launches,
             "algorithmic_bytes_per_launch": bytes_launch, "bytes_model": model,
             "voxel_views_per_launch": units}
        if path == "fused":
            eq = (b_alg_per_vv * n_local * V) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else 0.0
            r["equiv_streaming_note"] = ("the same batch done one view per launch would move %.0fx the bytes: "
                                         "%.1fx the per-view streaming roofline (a speed ratio, not HBM "
                                         "utilisation)" % (b_alg_per_vv * n_local * V / bytes_launch, eq))
        return r

    traffic = {}
    if os.path.exists(a.traffic_json):
        try:
            traffic = json.load(open(a.traffic_json))
        except Exception:
            traffic = {}

    def traffic_for(path):
        # PMC traffic was collected for ONE per-rank shape (the N = 1 workload): any other slab gets null
        if world != 1 or list(sb.slab_shape) != [a.n, a.n, a.n]:
            return None
        key = f"{path}_{a.scene}_{a.n}_{a.views}"
        ent = traffic.get(key)
        return ent.get("hbm_bytes_per_launch") if isinstance(ent, dict) else None

    if rank == 0:
        value_carve = n_total * V * a.steps / dt / 1e6
        dt_value = dt_asm if dt_asm is not None else dt
        value = n_total * V * a.steps / dt_value / 1e6
        out = {
            "metric": "Mvoxels*views/sec space-carve, 512^3 x 72 views per MI355X",
            "value": value, "unit": "Mvoxel*views/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt_value / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE cfg 3: {a.n}^3 voxels x {V} views per GPU, scene S1 "
                                   f"'{a.scene}' (SURVEY 8d), masks {W}x{H} uint8 resident in HBM",
                       "global_grid": gshape, "slab_per_gpu": list(sb.slab_shape),
                       "parallelism": (f"x-planes cyclic over {world} rank(s); `value` = carve + sparse pack + RCCL all-gather of "
                                       f"the labels in the brick-sparse form per step (a code per 16 x 64-voxel brick + the "
                                       f"2-bit labels of the mixed bricks: every GPU ends every step with the whole grid), "
                                       f"the collective enqueued by the library (sc_all_gather_sparse) beside the next "
                                       f"step's carve; `value_carve_only` has no collective; `strong` splits ONE {a.n}^3 "
                                       f"grid over the ranks") if dt_asm is not None
                                      else "one GPU, the whole grid: no collective",
                       "path": a.path, "views_per_launch": V if a.path == "fused" else 1,
                       "arithmetic": "float32 projection (no contraction, correctly rounded divide) into int32 labels"},
            "roofline": (roof(a.path, stats, live_traffic["hbm_bytes_per_launch"], live_traffic["source"])
                         if live_traffic is not None and world == 1 and list(sb.slab_shape) == [a.n, a.n, a.n]
                         else roof(a.path, stats, traffic_for(a.path))),
            "kernels": {k: stats[k] for k in ("carve", "step") if stats[k]["launches"]},
            "kernels_breakdown_pass": breakdown,
        }
        if live_traffic is not None:
            out["roofline"]["traffic_read_bytes"] = live_traffic["read_bytes"]
            out["roofline"]["traffic_write_bytes"] = live_traffic["write_bytes"]
            out["roofline"]["traffic_per_kernel"] = live_traffic["per_kernel"]
        out["lead_in"] = {"seconds_of_other_legs_before_warmup": t_device_before,
                          "note": "the stream + per_view legs run before the headline's warm-up (the W warm-up and K timed "
                                  "steps are unchanged); 0 with --skip-other-path.  Measured and NOT done: the "
                                  "averaging forms and the other scenes in front as well -- behind 150 ms of "
                                  "VALU-heavy averaging the same 20 steps read 0.1755-0.1840 ms (DESIGN_APPENDIX 12)"}
        if cold is not None:
            out["cold_first_batch_ms"] = cold["cold_first_batch_ms"]
            out["cold_first_batch"] = cold
        if res_other is not None:
            dto, statso, osteps = res_other
            out[other] = {"value": n_total * V * osteps / dto / 1e6, "unit": "Mvoxel*views/s",
                          "steps": osteps, "ms_per_step": dto / osteps * 1e3,
                          "roofline": roof(other, statso, traffic_for(other)), "kernels": statso}
        if per_view is not None:
            out["per_view"] = per_view
        if pipelined is not None:
            out["pipelined"] = pipelined
        if dt_asm is not None:
            out["value_carve_only"] = value_carve
            out["ms_per_step_carve_only"] = dt / a.steps * 1e3
            out["value_with_assembly"] = value  # (the name of rounds 2-4; `value` IS the with-assembly rate now)
            out["value_is"] = ("carve + assembly: K steps of clear + 72 resident masks + carve + brick-sparse pack (codes + mixed "
                               "bricks, from the batch's verdict bytes and live list) + all-gather into alternating receive "
                               "buffers, the collective of step k beside the carve of step k + 1, step k's headers checked "
                               "while step k + 1 runs, everything waited for inside the timed region; the roofline object "
                               "describes the carve-only span"
                               + ("; TWO engines on the rank's planes take the steps in turn (double buffering: the pack of "
                                  "step k is beside the carve of step k + 1 too) -- `ms_per_step_one_engine` is the same loop "
                                  "on one engine" if dt_asm_one is not None else ""))
            if dt_asm_one is not None:
                out["ms_per_step_one_engine"] = dt_asm_one / a.steps * 1e3
                out["value_one_engine"] = n_total * V * a.steps / dt_asm_one / 1e6
                out["twin_engine_labels_equal"] = bool(twin_check)
            out["assembly_transport"] = ("library RCCL (sc_comm_create / sc_all_gather_sparse: no torch in the data path)"
                                         if sb.comm is not None else "torch.distributed (%s%s)" % (
                                             "gloo, staged through the hosts" if a.dist_backend != "torch-nccl" else "nccl",
                                             "; the library's communicator failed: " + comm_error if comm_error else ""))
            out["assembly_bytes_sent_per_rank"] = int(sb.sparse_rank_bytes())
        if asm is not None:
            out["assembly"] = asm
        if parity is not None:
            out["parity_check"] = parity
        if cold_proc is not None:
            out["cold_process_first_batch_ms"] = cold_proc.get("first_batch_ms")
            out["cold_process"] = cold_proc
        if strong is not None:
            out["strong"] = strong
        if e2e is not None:
            out["e2e"] = e2e
        if extras is not None:
            out["scenes"] = extras
        if avg is not None:
            out["average"] = avg
        if gather is not None:
            out["gather"] = gather
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(gshape, origin, vs, views, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    eng.dev_free(masks_dev)
    BARRIER = None
    if twin is not None:
        twin.close()
    sb.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: PARITY CHECK FAILED: %s\n" % json.dumps(parity))
        sys.stderr.flush()
        if COMM_STUCK:
            os._exit(3)
        sys.exit(3)
    if COMM_STUCK:  # the stuck thread holds library locks an ordinary interpreter shutdown would wait for
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
