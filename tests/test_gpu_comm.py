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


def _device_count():
    n = C.c_int()
    assert L.load().bsk_device_count(C.byref(n)) == L.OK
    return n.value


def test_all_devices_shard_one_batch_and_gather_counts():
    """n = every visible GPU (1 on the test box, 8 on a node): n contexts, n threads each sketching its shard_range of ONE batch, then the
    counters of all ranks through bsk_comm_init_all / bsk_gather_counts_all.  The order-independent digest makes the whole-job check exact:
    the ranks' checksums and tuple counts add up to those of the whole batch on one context."""
    import threading
    from bio_amd.shard import shard_range
    n = _device_count()
    assert n >= 1
    lib = L.load()
    rng = np.random.default_rng(11)
    nreads = 60_000
    lens = rng.integers(100, 151, nreads).astype(np.uint64)
    offs = np.zeros(nreads + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    engines = [S.Engine(d) for d in range(n)]
    p = engines[0].params(L.MINIMIZER, 21, w=11)
    whole = engines[0].run(engines[0].batch_from_arrays(data, offs), p).digest()
    mine = np.zeros((n, 3), np.uint64)
    errs = []

    def work(rank):
        try:
            lo, hi = shard_range(nreads, rank, n)
            sub = offs[lo:hi + 1] - offs[lo]
            b = engines[rank].batch_from_arrays(data[int(offs[lo]):int(offs[hi])], sub)
            d = engines[rank].run(b, p).digest()
            mine[rank] = [d["checksum"], d["n_tuples"], hi - lo]
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs, errs
    ctxs = (C.c_void_p * n)(*[e.ctx for e in engines])
    assert lib.bsk_comm_init_all(ctxs, n) == L.OK
    allc = np.zeros((n, n, 3), np.uint64)  # [receiving rank][rank][counter]
    assert lib.bsk_gather_counts_all(ctxs, n, mine.ctypes.data, 3, allc.ctypes.data) == L.OK
    for r in range(n):
        assert np.array_equal(allc[r], mine), r
    assert int(mine[:, 0].sum(dtype=np.uint64)) == whole["checksum"]
    assert int(mine[:, 1].sum()) == whole["n_tuples"] and int(mine[:, 2].sum()) == nreads
    for e in engines:
        lib.bsk_comm_destroy(e.ctx)


def test_pipeline_over_all_devices_equals_one_device():
    """bsk_pipeline_memory_multi / _fastx_multi: one producer side, chunks taken by the workers of every device.  n = every visible GPU, and
    -- so that the dealing between devices runs on a one-GPU box too -- the same GPU named twice."""
    import tempfile
    n = _device_count()
    rng = np.random.default_rng(12)
    nreads = 50_000
    offs = np.arange(nreads + 1, dtype=np.uint64) * 150
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, nreads * 150, dtype=np.uint8)]
    eng = S.Engine(0)
    p = eng.params(L.MINIMIZER, 21, w=11)
    one = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=4096)
    for devices in (list(range(n)), [0, 0], list(range(n)) * 2):
        st = S.Engine.pipeline_memory_multi(devices, data, offs, p, n_streams=2, chunk_records=4096)
        assert (st["records"], st["tuples"], st["checksum"]) == (one["records"], one["tuples"], one["checksum"]), devices
        assert st["n_streams"] == 2 * len(devices) and st["chunks"] == one["chunks"]
    with tempfile.TemporaryDirectory() as d:
        path = d + "/r.fq"
        with open(path, "wb") as f:
            for i in range(5000):
                sq = data[i * 150:(i + 1) * 150].tobytes()
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, sq, b"I" * 150))
        a = S.Engine.pipeline_fastx(path, p, n_streams=2, chunk_records=512)
        b = S.Engine.pipeline_fastx_multi([0, 0], path, p, n_streams=1, chunk_records=512)
        assert (a["records"], a["tuples"], a["checksum"]) == (b["records"], b["tuples"], b["checksum"]) and a["records"] == 5000
    lib = L.load()
    st = L.PipelineStats()
    assert lib.bsk_pipeline_memory_multi(None, 1, data.ctypes.data, offs.ctypes.data, nreads, L.ALPHA_DNA, C.byref(p), 1, 4096, 1, 1, C.byref(st)) == L.ERR_ARG


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


def test_bench_self_launched_form():
    """The self-launched form of bench.py (`python bench.py --gpus N` with no launcher environment re-executes itself through
    torch.distributed.run) on the one GPU of this box: BSK_BENCH_SELF_LAUNCH=1 takes that route at N = 1, BSK_BENCH_FORCE_COMM=1 the
    multi-rank code behind it.  (N = 2 over gloo, no GPU: tests/test_shard_gloo.py.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BSK_BENCH_SELF_LAUNCH="1", BSK_BENCH_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "1e6", "--no-cpu-baseline", "--no-end-to-end", "--no-power-probe"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "bsk_gather_counts (RCCL)" in out["config"]["parallelism"]
