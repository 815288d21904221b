"""bench.py's host-side pieces that need no GPU: the averaging ceiling from the committed counters and probe prices, the
self-launch of N ranks from a plain command (here: that a failure of the ranks reaches the caller as a non-zero exit
code and no line), and the refusal of a packed grid on an engine without a device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_averaging_ceiling_is_built_from_the_committed_counters():
    import bench
    nvv = 512 ** 3 * 72
    for form in ("u8", "f32"):
        ceiling_ms, lane_ops, ns_vv, mix, stream_ms = bench.valu_ceiling(form, float(nvv))
        d = json.load(open(os.path.join(ROOT, bench.AVG_COUNTERS[form])))
        k = [v for n, v in d["kernels"].items() if n.startswith("average_brick_kernel")][0]
        assert abs(lane_ops - k["SQ_INSTS_VALU"] * 64 / nvv) < 1e-9 and 30 < lane_ops < 40
        assert abs(sum(mix.values()) - lane_ops) < 1e-6 and mix["other"] >= 0
        # a lower bound of the time the mix needs: cheaper than the measured launch, dearer than the cheapest class flat;
        # the rate a stream of the projection's own mix reaches lies between the two
        assert ceiling_ms < stream_ms < k["mean_us"] / 1e3
        assert ceiling_ms > lane_ops / 64 * bench.VALU_COST_NS["other"] * nvv / bench.SIMDS * 1e-6
        half = bench.valu_ceiling(form, nvv / 2.0)[0]
        assert abs(half * 2 - ceiling_ms) < 1e-9
    # the prices are the probe's: every class the model names stands in the committed table
    table = open(os.path.join(ROOT, bench.VALU_PROBE)).read()
    for name in ("v_add_f32", "v_mul_f32", "v_fma_f32", "v_rcp_f32", "v_cvt_i32_f32", "v_mad_u32_u24", "four voxels"):
        assert name in table


@pytest.mark.timeout(300)
def test_a_failure_of_the_self_launched_ranks_reaches_the_caller():
    """`python3 bench.py --gpus 2` from a plain shell starts its ranks as children; here they cannot get a device (the
    CPU suite runs without one, or with HIP_VISIBLE_DEVICES emptied), so the parent must exit non-zero and print no line."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--n", "64", "--views", "8", "--dist-backend", "gloo", "--share-device"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""


def test_a_packed_grid_needs_the_hip_engine():
    """`all_gather(unpack=False)` hands out a PackedGrid, which only HIP kernels read; on an engine without
    a device (the CPU rehearsal's stand-in) that is a clear error, not a torch failure three calls later."""
    from plant3dvision_amd.sharded import ShardedBackprojection
    from tests.helpers import OracleEngine, scene
    shape, origin, vs, views = scene((6, 16, 64), 3, "plant")
    sb = ShardedBackprojection(shape, origin, vs, rank=0, world_size=1, engine_factory=OracleEngine)
    for K, R, t, m in views:
        sb.process_view(K, R, t, m)
    for comp in ("2bit", "1bit"):
        with pytest.raises(ValueError, match="HIP engine"):
            sb.all_gather(compress=comp, unpack=False)
    with pytest.raises(ValueError, match="overlap"):
        sb.all_gather(compress="2bit", overlap=True)
    sb.close()
