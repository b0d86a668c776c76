"""The one collective of the path through the C ABI: bsk_comm_* / bsk_gather_counts (RCCL all_gather of u64 counters).

The GPU box has one GPU, so the communicator has world size 1 -- what this proves is that librccl loads from inside
libbiosketch.so, that both ways of forming a communicator work (unique id + init_rank: one process per GPU; init_all: one
thread driving several contexts) and that the gather moves the counters through the device.  The N > 1 control flow around
it is covered on CPUs by tests/test_shard_gloo.py."""
import ctypes as C

import numpy as np
import pytest

from bio_amd import _lib as L
from bio_amd import sketches as S

pytestmark = pytest.mark.gpu


def test_gather_counts_rank_world1():
    eng = S.Engine(0)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init_rank(uid, 0, 1)
    mine = [123456789012345, 15_000_000_000, 2_211_224_063, 88, 0]
    assert eng.gather_counts(mine) == [mine]
    assert eng.gather_counts([2 ** 64 - 1]) == [[2 ** 64 - 1]]  # u64 all the way
    with pytest.raises(S.DeviceError):  # a context joins one communicator
        eng.comm_init_rank(uid, 0, 1)
    # the engine still sketches after the collective (same stream)
    b = eng.synth(L.ALPHA_DNA, 1000, 150, 7)
    assert eng.run(b, eng.params(L.MINIMIZER, 21, w=11)).info()["n_tuples"] > 20000


def test_gather_counts_all_world1_and_errors():
    lib = L.load()
    eng = S.Engine(0)
    ctxs = (C.c_void_p * 1)(eng.ctx)
    mine = np.array([5, 6, 7], np.uint64)
    out = np.zeros(3, np.uint64)
    # no communicator yet: a loud argument error, not a hang
    assert lib.bsk_gather_counts_all(ctxs, 1, mine.ctypes.data, 3, out.ctypes.data) == L.ERR_ARG
    assert lib.bsk_gather_counts(eng.ctx, mine.ctypes.data, 3, out.ctypes.data) == L.ERR_ARG
    assert lib.bsk_comm_init_all(ctxs, 1) == L.OK
    assert lib.bsk_gather_counts_all(ctxs, 1, mine.ctypes.data, 3, out.ctypes.data) == L.OK
    assert out.tolist() == [5, 6, 7]
    assert lib.bsk_gather_counts(eng.ctx, mine.ctypes.data, 17, out.ctypes.data) == L.ERR_ARG  # more than BSK_MAX_COUNTERS
    lib.bsk_comm_destroy(eng.ctx)
    assert lib.bsk_gather_counts(eng.ctx, mine.ctypes.data, 3, out.ctypes.data) == L.ERR_ARG


def test_bench_multi_rank_path_with_torch_rccl_alive():
    """bench.py's N > 1 branch -- bsk_comm_unique_id -> bsk_comm_init_rank -> bsk_gather_counts INSIDE a process that has torch's own
    RCCL process group alive -- run at world size 1 under torchrun (BSK_BENCH_FORCE_COMM=1): the first 8-GPU scaling run must not be
    the first time this code executes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSK_BENCH_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "1e6", "--no-cpu-baseline", "--no-end-to-end"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert "counters gathered by bsk_gather_counts (RCCL)" in out["config"]["parallelism"], out["config"]["parallelism"]
    assert out["n_gpus"] == 1 and out["value"] > 0
