"""N > 1 on the HIP path.  Two ranks share the one GPU of the test box (collectives over gloo on the CPU --
HSW_BENCH_BACKEND=gloo / HSW_BENCH_SAME_DEVICE=1 are rehearsal knobs, the driver's 8-GPU run uses RCCL):

 * each rank expands its contiguous shard WITH libhsw (closed-form chip cursor, no data-path collective);
   the all-gathered union must be bit-equal to the serial oracle streams -- gate stream, both chip column
   families (shard boundary inside a chip row included) and the seeds route;
 * `python bench.py --gpus 2` starts its two ranks itself and reports n_gpus == rccl_ranks == 2.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_blocks, bits, ncols, cursor0, q):
    try:
        sys.path.insert(0, ROOT)
        import importlib
        import torch
        import torch.distributed as dist
        from oracle import oracle as O
        hsw = importlib.import_module("halo2-dynamic-sha256_amd")
        sh = importlib.import_module("halo2-dynamic-sha256_amd.sharding")
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        rng = np.random.default_rng(31415)
        blocks = rng.integers(0, 256, (n_blocks, 64), dtype=np.uint8)
        pre = rng.integers(0, 2**32, (n_blocks, 8), dtype=np.uint64).astype(np.uint32)
        eng = hsw.WitnessEngine(0, bits, ncols)                   # both ranks on cuda:0
        starts, counts = zip(*[sh.shard_range(n_blocks, world, r) for r in range(world)])
        s, c = starts[rank], counts[rank]
        cur = sh.shard_cursor(cursor0, s, eng.limb_calls)
        out = eng.witness_blocks(torch.from_numpy(blocks[s:s + c].copy()).cuda(),
                                 torch.from_numpy(pre[s:s + c].view(np.int32).copy()).cuda(), cursor0=cur)
        eng.synchronize()
        gate = sh.allgather_gate(dist, out["gate"].cpu(), counts, eng.G)
        dense = sh.allgather_chip(dist, out["dense"].cpu(), cursor0, starts, counts, eng.limb_calls, ncols)
        spread = sh.allgather_chip(dist, out["spread"].cpu(), cursor0, starts, counts, eng.limb_calls, ncols)
        serial = O.Oracle(bits, ncols, check=True).witness_blocks(blocks, pre, cursor0=cursor0)
        ok = (np.array_equal(gate.numpy().view(np.uint64), serial["gate"])
              and np.array_equal(dense.numpy().view(np.uint64), serial["dense"])
              and np.array_equal(spread.numpy().view(np.uint64), serial["spread"]))
        # the cheaper exchange: 96-byte seeds, every rank re-expands everything on its GPU
        sb, sp = sh.allgather_seeds(dist, torch.from_numpy(blocks[s:s + c].copy()),
                                    torch.from_numpy(pre[s:s + c].view(np.int32).copy()), counts)
        allout = eng.witness_blocks(sb.cuda(), sp.cuda(), cursor0=cursor0)
        eng.synchronize()
        ok = ok and np.array_equal(allout["gate"].cpu().numpy().view(np.uint64), serial["gate"])
        ok = ok and np.array_equal(allout["dense"].cpu().numpy().view(np.uint64), serial["dense"])
        ok = ok and np.array_equal(allout["next_states"].cpu().numpy().view(np.uint32), serial["next_states"])
        eng.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok), ""))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("n_blocks,bits,ncols,cursor0", [(6, 8, 2, 0), (5, 8, 3, 1)])
def test_two_ranks_expand_their_shards_with_libhsw(n_blocks, bits, ncols, cursor0):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_blocks * 11 + ncols) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_blocks, bits, ncols, cursor0, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err)


def test_bench_gpus_2_starts_two_ranks():
    """The driver's command line.  On a 1-GPU box the two ranks share cuda:0 and reduce over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSW_BENCH_BACKEND="gloo", HSW_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--messages-per-gpu", "512", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["scaling"] == "weak"
    assert j["config"]["blocks_per_gpu"] == 512
    mg = j["extra"]["multi_gpu"]
    assert mg["allgather"]["own_shard_intact"] and mg["kernel_plus_allgather_blocks_per_s"] > 0
    assert mg["kernel_only_blocks_per_s"] == j["value"]
    assert j["roofline"]["launch"]["n_blocks"] == 512
