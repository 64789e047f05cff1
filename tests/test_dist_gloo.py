"""N > 1 path of bench.py on CPU: two gloo ranks drive the same plumbing the GPU run uses
(barrier + sync on both sides of the timed region, MAX over ranks, whole-job aggregation,
per-rank synthetic shards).  No scoring happens here (that needs a GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import numpy as np
import pytest
import bench, _gmg_pkg
gmg = _gmg_pkg.load()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# every rank owns its own shard: different seeds give different reads, same shape
packed, off = gmg.synth.packed_reads(1000, 500, bench.SEED + rank)
calls = []
def step():
    calls.append(time.perf_counter())
    time.sleep(0.02 * (1 + rank))            # rank 1 is the slow one
seconds, mine = bench.timed_region(step, 5, 2, lambda: None, dist)
out = {"rank": rank, "seconds": seconds, "mine": mine, "calls": len(calls),
       "digest": int(np.bitwise_xor.reduce(packed)), "total": int(off[-1]),
       "value": bench.aggregate(int(off[-1]) * world, 5, seconds)}
print("RESULT " + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_timed_region_and_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        res.append(json.loads([ln for ln in out.splitlines() if ln.startswith("RESULT ")][0][7:]))
    res.sort(key=lambda r: r["rank"])
    # EXACTLY warmup + steps calls per rank
    assert [r["calls"] for r in res] == [7, 7]
    # the reported time is the MAX over ranks, identical on both, and not smaller than the slow rank's own time
    assert abs(res[0]["seconds"] - res[1]["seconds"]) < 1e-9
    assert res[0]["seconds"] >= max(r["mine"] for r in res) - 1e-9
    assert res[0]["seconds"] >= 5 * 0.04 * 0.9
    # the barrier makes the fast rank wait: its own bracketed time is also about the slow rank's
    assert res[0]["mine"] >= 5 * 0.04 * 0.8
    # weak scaling: shards differ, whole-job value = all ranks' bases / max time
    assert res[0]["digest"] != res[1]["digest"]
    assert abs(res[0]["value"] - 2 * 500_000 * 5 / res[0]["seconds"]) < 1e-6


def test_shard_plan_covers_everything_once():
    import bench
    for total, world in ((10, 3), (1_000_000, 8), (7, 8), (0, 2)):
        plan = bench.shard_plan(total, world)
        assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        sizes = [hi - lo for lo, hi in plan]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_with_rccl_on_one_gpu(tmp_path):
    """the launch line the driver uses for N > 1, with N = 1 and the process group forced on: init over RCCL, the timing
    barrier, the MAX reductions and the JSON contract on a real GPU"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, GMG_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--reads", "50000", "--cpu-reads", "0"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout: RCCL's version banner and the like go to stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 1000
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
