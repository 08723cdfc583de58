"""bench.py's N > 1 code path on the 1-GPU box: `JPK_BENCH_ONE_GPU=1` puts every rank on cuda:0 and carries the gather over gloo,
so that launch_children, the per-rank timing all-reduce, the (count, sizes) all_gather + exact-size sends of
jampack_amd/shard.py and the block-order reassembly all run before the driver's first real `--gpus 8`.  The children are fresh
processes spawned by a parent that never touches the GPU (bench.py itself, started here with subprocess).  Not a scaling number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra_args, port):
    env = dict(os.environ)
    env.update({"JPK_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--contexts", "2",
           "--master-port", str(port)] + extra_args
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected (rank 0 only): {lines}"
    return json.loads(lines[0])


def test_default_workload_two_ranks_weak_scaling_line():
    # every rank compresses a batch of its own (three 16 MiB-class blocks), rank 0 gathers both batches every step
    line = _run(["--block-mib", "16", "--limit-bytes", str(40 << 20)], 29731)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["gather_ok"] is True
    assert line["collective"]["ranks"] == 2 and line["collective"]["backend"] == "gloo" and line["collective"]["gathers_timed"] == 2
    assert line["value"] > 0 and line["unit"] == "MB/s"
    assert len(line["config"]["block_bytes"]) == 3
    assert "gloo gather" in line["config"]["parallelism"] and "RCCL" not in line["config"]["parallelism"]


def test_stream_workload_two_ranks_strong_scaling_line():
    # BASELINE config 4's shape, truncated: one stream -> 3 blocks, block b on rank b mod 2, payload reassembled in block order
    line = _run(["--workload", "enwik9", "--limit-bytes", str(150 << 20)], 29741)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["gather_ok"] is True
    assert line["config"]["block_bytes"] == [64 << 20, 64 << 20, 22 << 20]
    assert line["gathered_payload_bytes"] > 0
    assert "TRUNCATED" in line["config"]["workload"] and "gloo gather" in line["config"]["workload"]


def test_more_ranks_than_blocks():
    # a rank that owns no block still takes part in every gather (jampack.cpp:209-213: the last read returns 0 bytes)
    line = _run(["--workload", "enwik9", "--limit-bytes", str(20 << 20)], 29751)
    assert line["n_gpus"] == 2 and line["gather_ok"] is True and line["config"]["block_bytes"] == [20 << 20]
