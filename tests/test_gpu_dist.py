"""RCCL on the hardware the driver tests on: bench.py as a fresh child process with WORLD_SIZE=1 and EVH_BENCH_FORCE_DIST=1, so that
process-group initialisation over RCCL (backend "nccl"), the all_gather_into_tensor of the H records on the side stream and the
event ordering around it (bench.py: step() of the pair configs, gather_static_rows of the two-phase stream) run on every round's
box, not only when an 8-GPU node is at hand.  The world-2 forms stay in tests/test_sharding_gloo.py (gloo, CPU) and in the RCCL
world-2 test of tests/test_gpu_parity.py (skipped on one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               EVH_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-pairs", "0",
                        "--skip-no-temporal", "--gen-procs", "1"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_pair_batch_with_rccl_gather_world_1():
    d = run_bench(["--pairs", "16", "--unique", "4"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["pairs_ok_fraction"] == 1.0
    assert d["roofline"]["kernel"] in d["roofline"]["stage_ms"]


def test_two_phase_stream_with_rccl_gather_world_1():
    d = run_bench(["--config", "4", "--pairs", "6", "--width", "640", "--height", "360"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["pairs_ok_fraction"] > 0.9
