"""The N > 1 path on the real backend (SURVEY 8(e) "tooling caveat"): ``torch.distributed.run`` with one process per
visible GPU (1 on the lease box, 8 when the driver has a node) as a fresh CHILD process -- never an exec of the test
process -- running tests/_dist_nccl_worker.py, which asserts gathered == unsharded bitwise on backend "nccl" (RCCL)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_predict_sharded_on_rccl_equals_unsharded():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    # One rank per visible GPU -- on an 8-GPU node world = 8, so the first world-8 RCCL all_gather is this test and not the benchmark.
    # The lease boxes allow at most 6 processes on a card at once: with fewer than 8 devices the world is capped at 6 (they have one
    # GPU, so world = 1 there).  UFM_TEST_NCCL_RANKS overrides either way.
    ndev = torch.cuda.device_count()
    n = int(os.environ.get("UFM_TEST_NCCL_RANKS", "0")) or (ndev if ndev >= 8 else min(ndev, 6))
    assert 1 <= n <= ndev, f"UFM_TEST_NCCL_RANKS={n} but {ndev} device(s) are visible"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_nccl_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"DIST-NCCL-OK world={n}" in r.stdout


def test_three_ranks_sharing_the_one_gpu_over_gloo_equal_unsharded():
    """More than one rank of the N > 1 path with REAL device compute on a one-GPU box: three processes share cuda:0 (RCCL refuses
    two ranks on one device, so the gather of the device buffers goes through gloo).  Every rank computes its shard with the HIP
    kernels, the packed device results are gathered, and each rank checks gathered == its own unsharded run bit for bit -- the
    synchronous form, the asynchronous ring (flag rows through the side stream + pinned host buffer) and the fewer-pairs-than-ranks
    case.  A functional rehearsal, not a measurement."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", UFM_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_nccl_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST-GLOO-OK world=3" in r.stdout
