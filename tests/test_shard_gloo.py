"""CPU, 2 processes over gloo: the N>1 plumbing of bench.py (shard by record, gather counters, max/sum)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_tile_exactly():
    from bio_amd.shard import shard_range
    for n in (0, 1, 7, 64, 1_000_003):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_over_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch.distributed as dist
        from bio_amd.shard import shard_range, gather_counters, whole_job
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        lo, hi = shard_range(1001, rank, world)
        # pretend: each rank hashed its records of 150 bases in (rank+1) seconds and found 22 tuples per record
        rows = gather_counters([float(rank + 1), (hi - lo) * 150.0, (hi - lo) * 22.0])
        job = whole_job(rows, steps=2)
        if rank == 0:
            print(json.dumps({{"rows": rows, "job": job}}))
        dist.barrier()
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["rows"] == [[1.0, 501 * 150.0, 501 * 22.0], [2.0, 500 * 150.0, 500 * 22.0]]
    assert d["job"]["seconds"] == 2.0 and d["job"]["bases"] == 1001 * 150.0 and d["job"]["tuples"] == 1001 * 22.0
    assert abs(d["job"]["gbases_per_s"] - 1001 * 150.0 * 2 / 2.0 / 1e9) < 1e-12
