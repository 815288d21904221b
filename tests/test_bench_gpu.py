"""bench.py as the driver runs it: one JSON line from a plain command, at N = 1 and -- started by bench.py itself as
fresh child processes, the parent never touching the GPU -- at N = 2 (two ranks sharing the one GPU of the box, gloo as
the transport: the code path of an 8-GPU RCCL run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--n", "128", "--views", "12", "--extra-steps", "0", "--cpu-seconds", "0", "--e2e-reps", "0",
         "--cold-reps", "0", "--skip-other-path", "--traffic-passes", "off"]


def _run(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)  # a plain shell
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_value_includes_the_assembly(gpu_device):
    out = _run(["--gpus", "2", "--share-device", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                "--assembly-steps", "2", "--strong-steps", "2"] + SMALL)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    # `value` is the with-assembly rate (t_device + collective); carve only stands beside it
    assert out["value"] == out["value_with_assembly"]
    assert out["value_carve_only"] >= out["value"] > 0
    nvox = 1
    for s in out["config"]["global_grid"]:
        nvox *= s
    assert abs(out["value"] - nvox * 12 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * out["value"]
    asm = out["assembly"]
    assert "error" not in asm and asm["packed_grid_serial"]["value"] > 0 and asm["occupancy_1bit"]["value"] > 0
    assert asm["value_with_unpacked_assembly"] > 0 and asm["gather_to_host_ms"] > 0
    assert "error" not in out["strong"] and out["strong"]["value_carve_only"] >= out["strong"]["value"] > 0
    pc = out["parity_check"]  # rank 0's planes of the grid, every voxel against the oracle
    assert pc["ok"] is True and pc["oracle_whole_grid"] is True and pc["oracle_voxels"] == nvox // 2
    assert out["roofline"]["bound"] == "hbm" and out["cpu_baseline"] is None


@pytest.mark.gpu
def test_bench_line_at_one_gpu_carries_parity_and_cold_process(gpu_device):
    out = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--cold-process", "on"] + SMALL)
    assert out["n_gpus"] == 1 and "value_carve_only" not in out
    pc = out["parity_check"]
    assert pc["ok"] is True and pc["fused_equals_per_view"] and pc["oracle_whole_grid"] is True
    assert pc["oracle_digest_key"] == "plant_128_12" and pc["fused_equals_committed_oracle_digest"] is True
    assert sum(pc["labels_histogram"]) == 128 ** 3
    pl = out["pipelined"]  # two scans in flight: reported beside `value`, never as it
    assert pl["engines"] == 2 and pl["labels_equal"] is True and pl["value"] > 0
    cp = out["cold_process"]
    assert "error" not in cp, cp
    assert out["cold_process_first_batch_ms"] == cp["first_batch_ms"] > 0
    assert abs(cp["first_batch_ms"] - (cp["create_ms"] + cp["enqueue_ms"] + cp["wait_ms"])) < 1e-6
    assert cp["allocations"] >= 3  # the label volume, the survivor lists, the control block at least


@pytest.mark.gpu
def test_bench_rehearsal_carries_the_assembly_over_the_library_communicator(gpu_device):
    """`bench.py --rccl-rehearsal`: the N > 1 code path with an RCCL group of ONE on the one-GPU box -- the library's own
    communicator (sc_comm_create), the brick-sparse form, the collective beside the next carve, the headers checked one
    step late; gloo is the control plane only."""
    out = _run(["--rccl-rehearsal", "--steps", "4", "--warmup", "2", "--assembly-steps", "3", "--strong-steps", "2"] + SMALL)
    assert out["n_gpus"] == 1 and "library RCCL" in out["assembly_transport"]
    assert out["value"] == out["value_with_assembly"] and out["value_carve_only"] >= out["value"] > 0
    assert 0 < out["assembly_bytes_sent_per_rank"] < 128 ** 3 // 4  # less than the dense 2-bit form of the grid
    asm = out["assembly"]
    for leg in ("sparse_grid_serial", "sparse_grid_unpacked_int8", "packed_grid_serial", "packed_grid_overlapped", "occupancy_1bit"):
        assert asm[leg]["value"] > 0, leg
    assert asm["gather_to_host_sparse_ms"] > 0 and "error" not in out["strong"]
    pc = out["parity_check"]
    assert pc["ok"] is True and pc["oracle_whole_grid"] is True and pc["fused_equals_committed_oracle_digest"] is True
    # the assembled headline deals its steps to two engines in turn; the same loop on one engine stands beside it
    assert out["twin_engine_labels_equal"] is True and pc["twin_engine_labels_equal"] is True
    assert out["ms_per_step_one_engine"] > 0 and out["value_one_engine"] > 0 and "TWO engines" in out["value_is"]


@pytest.mark.gpu
def test_bench_falls_back_together_when_the_communicator_never_comes_up(gpu_device):
    """`--comm-timeout 0`: sc_comm_create (on its own thread) is not back in time, so the ranks agree on the staged
    gloo collectives, the line says so, the labels are the oracle's and the process ends with code 0 although a thread
    of it may still be inside the library."""
    out = _run(["--rccl-rehearsal", "--comm-timeout", "0", "--steps", "3", "--warmup", "1", "--assembly-steps", "2",
                "--strong-steps", "2"] + SMALL)
    assert "gloo" in out["assembly_transport"] and "did not return within 0 s" in out["assembly_transport"]
    assert out["value"] == out["value_with_assembly"] and out["value_carve_only"] >= out["value"] > 0
    assert out["parity_check"]["ok"] is True
