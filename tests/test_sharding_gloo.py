"""N>1 path on CPU: two ranks over gloo. Each rank owns a contiguous env block; the concatenation of the blocks equals the
single-process result (envs are independent, no data-path collective); barrier + max-over-ranks behave as bench.py needs."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, steps, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import aircombat_selfplay_amd as pkg
    from oracle import oracle as O   # the CPU engine stands in for the GPU block in this CPU-only test
    dist = pkg.sharding.init_process_group("gloo")
    assert dist is not None and pkg.sharding.dist_env() == (rank, world, rank)
    start, count = pkg.sharding.env_block(rank, world, total_envs)
    env = O.OracleVecEnv(O.default_config(O.TASK_SINGLECOMBAT), count)
    env.reset()
    rng = np.random.default_rng(99)
    acts = [np.stack([rng.integers(0, n, size=(total_envs, 2)) for n in (41, 41, 41, 30)], axis=-1) for _ in range(steps)]
    pkg.sharding.barrier(dist)
    for a in acts:
        obs, rew, done, info = env.step(a[start:start + count])
    mx = pkg.sharding.max_over_ranks([float(rank + 1), 10.0 - rank], dist)
    assert mx == [float(world), 10.0]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), obs=obs, rew=rew, start=start, count=count)
    pkg.sharding.barrier(dist)
    dist.destroy_process_group()


def test_two_rank_env_blocks_match_single_process(tmp_path, pkg, oracle):
    total, steps, world = 6, 12, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, steps, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert [int(p["start"]) for p in parts] == [0, 3] and sum(int(p["count"]) for p in parts) == total
    env = oracle.OracleVecEnv(oracle.default_config(oracle.TASK_SINGLECOMBAT), total)
    env.reset()
    rng = np.random.default_rng(99)
    for _ in range(steps):
        a = np.stack([rng.integers(0, n, size=(total, 2)) for n in (41, 41, 41, 30)], axis=-1)
        obs, rew, done, info = env.step(a)
    assert (np.concatenate([p["obs"] for p in parts]) == obs).all()
    assert (np.concatenate([p["rew"] for p in parts]) == rew).all()


def test_env_block_partition(pkg):
    for world in (1, 2, 3, 8):
        for total in (8, 13, 4096, 32768):
            blocks = [pkg.sharding.env_block(r, world, total) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_bench_control_flow_two_ranks_under_torchrun(tmp_path):
    """bench.py's multi-rank control flow exactly as the driver launches it (python -m torch.distributed.run, two ranks, gloo instead
    of RCCL, a stand-in env handle instead of a GPU): barrier, W untimed + exactly K timed steps per rank, max-over-ranks, ONE JSON
    line from rank 0, whole-job value, and a distinct seed block per rank (env i of the job keeps seed + 1000 i)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--envs", "64", "--stub-env"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 30 and r["warmup"] == 5 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["stub"]["steps_taken"] == 35 and r["stub"]["seed"] == 1                     # rank 0: envs 0 .. 63
    assert abs(r["value"] - 2 * 64 * 2 * 30 / (r["ms_per_step"] * 1e-3 * 30)) < 1e-6 * r["value"]   # whole-job agent-steps / max-over-ranks time
    assert r["ms_per_step"] >= 0.2                                                        # the stand-in sleeps 0.2 ms per step
    assert "stub" in r["data"]
    # the line proves who took part: one SUM all-reduce on the job's group carried every rank's 1, device ordinal and work
    assert r["ranks_reporting"] == 2 and [d["rank"] for d in r["devices"]] == [0, 1] and [d["device"] for d in r["devices"]] == [0, 1]
    assert r["agent_steps_per_rank"] == [64 * 2 * 30.0] * 2 and sum(r["agent_steps_per_rank"]) == 2 * 64 * 2 * 30
    assert r["collective_backend"] == "gloo"


@pytest.mark.parametrize("per_side,world", [(2, 2), (4, 2)])
def test_bench_multi_gpu_configs_c4_c5_under_torchrun(tmp_path, per_side, world):
    """BASELINE's multi-GPU configs through the same contract: `bench.py --gpus N --task scenario_nvn --per-side 2|4 --hierarchical` (C4 =
    Scenario2_NvN 2v2 over 2 GPUs, C5 = Scenario3_NvN 4v4 over 8; as shipped: [3,5,3] + four weapon bits, act_dim 7), rehearsed with two
    gloo ranks and the stand-in handle: the action batches have the config's shape, the value counts A = 2 per_side aircraft per env,
    the workload string names the config."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5", "--envs", "32", "--stub-env",
           "--task", "scenario_nvn", "--per-side", str(per_side), "--hierarchical"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    A = 2 * per_side
    assert r["n_gpus"] == world and r["config"]["aircraft_per_env"] == A and r["config"]["hierarchical"] is True
    assert r["stub"]["agents"] == A and r["stub"]["act_dim"] == 7 and r["stub"]["steps_taken"] == 25
    assert ("C4" if per_side == 2 else "C5") in r["config"]["workload"] and "as shipped" in r["config"]["workload"]
    assert abs(r["value"] - world * 32 * A * 20 / (r["ms_per_step"] * 1e-3 * 20)) < 1e-6 * r["value"]
    assert r["ranks_reporting"] == world and len(r["devices"]) == world and r["agent_steps_per_rank"] == [32.0 * A * 20] * world


def test_bench_rank_seed_blocks_do_not_overlap(pkg):
    """Rank r of an N-rank bench owns envs [r E, (r + 1) E) of the job and seeds its handle with 1 + 1000 r E, so env i of the job keeps
    the reference's seed + 1000 i (scripts/train/train_jsbsim.py:33) whatever N is: no two ranks share a stream."""
    E = 4096
    for world in (2, 8):
        seeds = set()
        for rank in range(world):
            start, count = pkg.sharding.env_block(rank, world, world * E)
            assert (start, count) == (rank * E, E)
            block = {1 + 1000 * start + 1000 * i for i in range(count)}
            assert not (seeds & block)
            seeds |= block
