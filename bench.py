#!/usr/bin/env python3
"""bench.py — agent-steps/sec of the vectorised air-combat step() on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it with
``python -m torch.distributed.run --nproc-per-node N``: one rank per GPU, weak scaling (every rank owns its own
contiguous block of 4096 envs, no collective in the env path — envs are independent). Rank 0 prints ONE JSON line.

Workload at every N: BASELINE.json configs[1] — SingleCombat 1v1 self-play, no weapons, 4096 envs per GPU, uniform random
integer actions regenerated every step (worst case for FCS activity), auto-reset on. A "step" is one pass of the hot path
over the whole batch: 6 FDM ticks per aircraft + observation / reward / termination, one kernel launch.

``value`` is SURVEY 8(d)'s metric: agent-steps per second of ``VecEnv.step(numpy actions) -> numpy obs / rewards / dones / infos`` at
the Python boundary, every ctypes call and every byte that crosses PCIe included. The same step with the
actions and the outputs resident in HBM (``step_device``) is reported beside it as ``device_resident``; the
``roofline`` of the dominant kernel is measured on that back-to-back device-resident leg with HIP events on the launch stream.
At N = 1 the line also carries ``configs`` (BASELINE configs C3, C4, C5 at 4096 envs per GPU), a saturating-batch leg and the
CPU baseline in SURVEY 8(d)'s two shapes.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TINST = 256 * 4 * 32 * 2.4e9 / 1e12   # lane-instructions/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (157.3 TFLOP/s fp32 = 2 flop FMA)
VALU_PER_AGENT_STEP = 8796.0                   # SQ_INSTS_VALU per wave per launch / 64 lanes x 64 (profiles/round1_pmc_mix.txt)


def algorithmic_bytes(env, missiles_in_flight=0.0):
    """512 B state + action + 4 * obs_dim + reward + done per agent-step (+ 192 B per live missile-step)."""
    return 512.0 + 4.0 * env.act_dim + 4.0 * env.obs_dim + 5.0 + 192.0 * missiles_in_flight


def action_pool(np, rng, env, count):
    E, A = env.num_envs, env.num_agents
    nvec = (3, 5, 3) if env.hierarchical else (41, 41, 41, 30)
    pool = []
    for _ in range(count):
        a = np.stack([rng.integers(0, n, size=(E, A)) for n in nvec], axis=-1).astype(np.float32)
        if env.act_dim > len(nvec):   # shoot bit / the four weapon bits: Bernoulli(0.05)
            a = np.concatenate([a, (rng.random((E, A, env.act_dim - len(nvec))) < 0.05).astype(np.float32)], axis=-1)
        pool.append(a)
    return pool


def pmc_traffic(task, envs, aircraft):
    """HBM bytes per launch of the step kernel from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    runs of this same command, FETCH_SIZE calibrated on the digest kernel's known byte count; tools/pmc_traffic.py writes the
    summary). Counters cannot be read from inside the process, so this is the committed measurement for the same workload, or
    null when there is none for it."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    for r in rec.get("runs", []):
        if r.get("task") == task and r.get("envs_per_gpu") == envs and r.get("aircraft") == aircraft and r.get("round", 1) >= 2:
            return r["traffic_bytes_per_launch"]
    return None


def host_leg(env, pool, steps, warmup, sync_all=None):
    """VecEnv.step(numpy) at the Python boundary: `warmup` untimed steps, then EXACTLY `steps` timed ones."""
    n = len(pool)
    for i in range(warmup):
        env.step(pool[i % n])
    if sync_all:
        sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(pool[(warmup + i) % n])
    t1 = time.perf_counter()
    if sync_all:
        sync_all()
    return t1 - t0


def device_leg(env, dev_ptrs, steps, warmup):
    """The same steps with actions / outputs resident in HBM, launched back to back; HIP events on the launch stream."""
    n = len(dev_ptrs)
    for i in range(warmup):
        env.step_device(dev_ptrs[i % n])
    env.sync()
    env.lib.check(env.lib.ac_timing_begin(env._h), "ac_timing_begin")
    t0 = time.perf_counter()
    for i in range(steps):
        env.step_device(dev_ptrs[(warmup + i) % n])
    env.sync()
    wall = time.perf_counter() - t0
    ev = C.c_float()
    env.lib.check(env.lib.ac_timing_end(env._h, C.byref(ev)), "ac_timing_end")
    return wall, ev.value / steps      # seconds of the loop, kernel ms per launch


def host_launch_ms(env, pool, steps=200):
    """Duration of the step kernel when it is launched the way VecEnv.step launches it (actions read from, outputs also written to,
    mapped host memory): HIP events on the launch stream around every launch of a short leg of its own, outside the timed region
    (two event records per step would cost the host leg what they measure)."""
    tot = 0.0
    ev = C.c_float()
    dll = env.lib.dll
    for i in range(steps + 20):
        cur = env._cur = env._cur ^ 1
        np_copy(env._sets[cur]["actions"], pool[i % len(pool)])
        env.lib.check(env.lib.ac_timing_begin(env._h), "ac_timing_begin")
        env.lib.check(dll.ac_step_host_async(env._h, cur), "ac_step_host_async")
        env.lib.check(env.lib.ac_timing_end(env._h, C.byref(ev)), "ac_timing_end")
        if i >= 20:
            tot += ev.value
    return tot / steps


def np_copy(dst, src):
    import numpy as np
    np.copyto(dst, src.reshape(dst.shape))


def roofline(env, task, kernel_ms, hierarchical=False):
    algo = algorithmic_bytes(env) * env.num_envs * env.num_agents            # per launch, one GPU
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": pmc_traffic(task, env.num_envs, env.num_envs * env.num_agents),
            "kernel": "step kernel of the task" + (" + controller_kernel" if hierarchical else ""), "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_agent_step": algorithmic_bytes(env)}


def config_leg(pkg, np, torch, name, task, per_side, envs, device_id, steps=300, warmup=60):
    """One more BASELINE config on this GPU: host-boundary rate, device-resident rate, kernel time and roofline."""
    cfg = pkg.default_config(task) if per_side == 1 else pkg.default_nvn_config(per_side, task=task)
    cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
    env = cls(cfg, envs, device_id=device_id, seed=1)
    env.reset()
    rng = np.random.default_rng(20250321)
    pool = action_pool(np, rng, env, 16)
    dev = [torch.from_numpy(a).cuda(device_id) for a in pool]
    torch.cuda.synchronize()
    hb = host_leg(env, pool, steps, warmup)
    wall, kernel_ms = device_leg(env, [t.data_ptr() for t in dev], steps, warmup)
    n = env.num_envs * env.num_agents
    out = {"config": name, "task": task, "envs_per_gpu": envs, "aircraft_per_env": env.num_agents, "obs_dim": env.obs_dim, "act_dim": env.act_dim,
           "value": n * steps / hb, "unit": "agent-steps/s", "ms_per_step": hb / steps * 1e3, "steps": steps, "warmup": warmup,
           "device_resident": {"value": n * steps / wall, "ms_per_step": wall / steps * 1e3},
           "roofline": roofline(env, task, kernel_ms)}
    env.close()
    return out


def saturating_leg(pkg, np, torch, cfg, device_id, envs=524288, steps=40, warmup=8):
    """SURVEY 8d asks for the same path at a saturating batch (>= 2^20 aircraft) beside the BASELINE batch."""
    env = pkg.HipVecEnv(cfg, envs, device_id=device_id, seed=7)
    env.reset()
    rng = np.random.default_rng(99)
    pool = [torch.from_numpy(a).cuda(device_id) for a in action_pool(np, rng, env, 4)]
    wall, kernel_ms = device_leg(env, [t.data_ptr() for t in pool], steps, warmup)
    n = env.num_envs * env.num_agents
    bytes_ = algorithmic_bytes(env)
    env.close()
    rate = n / (kernel_ms * 1e-3)
    return {"envs": envs, "aircraft": n, "steps": steps, "value": n * steps / wall, "unit": "agent-steps/s", "kernel_ms": kernel_ms,
            "hbm": {"achieved": bytes_ * rate / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": bytes_ * rate / 1e9 / HBM_PEAK_GBPS},
            "valu": {"achieved": VALU_PER_AGENT_STEP * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s",
                     "frac": VALU_PER_AGENT_STEP * rate / 1e12 / VALU_PEAK_TINST}}


def cpu_baseline(np, cfg, envs, agents, seconds_target=8.0):
    """the CPU restatement (oracle/, f64 C: the reference's own SubprocVecEnv + jsbsim wheel cannot run anywhere in this
    pipeline) behind the VecEnv surface in the two shapes named there, same config and action distribution as the GPU run, timed on
    this box's host cores on a bounded sample: (i) every env in one process on one thread at E = 32 (the authors' setting);
    (ii) one worker process per host core, each owning a block of the 4096 envs, pipes + pickle like env_wrappers.py:182-320."""
    from oracle import oracle as O
    from oracle.subproc_vec_env import OracleBlockVecEnv, OracleSubprocVecEnv
    ocfg = O.config_from_ac(cfg)
    rng = np.random.default_rng(20250321)

    def actions(E):
        return [np.stack([rng.integers(0, n, size=(E, agents)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32) for _ in range(8)]

    def timed(env, pool, budget):
        env.reset()
        for i in range(3):
            env.step(pool[i % 8])
        t0 = time.perf_counter()
        k = 0
        while True:
            env.step(pool[k % 8])
            k += 1
            if k >= 20 and time.perf_counter() - t0 >= budget:
                break
        return k, time.perf_counter() - t0

    one = OracleBlockVecEnv(ocfg, 32)
    k1, s1 = timed(one, actions(32), seconds_target * 0.4)
    one.close()
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, 16))     # a one-GPU box's CPU share is 16 cores (os.cpu_count() reports the whole host)
    sub = OracleSubprocVecEnv(ocfg, envs, workers)
    kw, sw = timed(sub, actions(envs), seconds_target * 0.6)
    sub.close()
    return {"value": envs * agents * kw / sw, "unit": "agent-steps/s", "cores": workers, "kind": "port", "host_cpu_count": cores,
            "single_thread": {"value": 32 * agents * k1 / s1, "unit": "agent-steps/s", "envs": 32, "cores": 1,
                              "sample": f"32 envs x {agents} aircraft x {k1} env steps in {s1:.1f} s, one process, one thread"},
            "sample": f"{envs} envs x {agents} aircraft x {kw} env steps in {sw:.1f} s: {workers} worker processes (one per core of this box's "
                      f"16-core share; os.cpu_count() = {cores} is the whole host), each stepping its block of envs per ('step', actions) message over a pipe, pickled "
                      f"numpy arrays both ways, auto-reset in the worker, parent concatenates (shape of envs/env_wrappers.py:182-320); "
                      f"uniform random actions, oracle/ = f64 C restatement of the JSBSim + Python path"}


class StubVecEnv:
    """--stub-env: a stand-in handle without a GPU, for the multi-rank control-flow rehearsal under gloo (tests/test_sharding_gloo.py):
    barrier, exactly-K timed steps, max-over-ranks, rank-0 print, distinct seed blocks. Never a measurement."""
    hierarchical = False

    def __init__(self, cfg, num_envs, device_id=0, seed=0):
        self.num_envs, self.num_agents, self.obs_dim, self.act_dim = num_envs, 2, 15, 4
        self.seed_value, self.steps_taken, self.closed = seed, 0, False

    def reset(self):
        return None

    def step(self, actions):
        self.steps_taken += 1
        time.sleep(2e-4)

    def close(self):
        self.closed = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (default: the BASELINE config)")
    ap.add_argument("--task", default="singlecombat", help="any name of aircombat_selfplay_amd.config.TASK_IDS (default: BASELINE configs[1])")
    ap.add_argument("--per-side", type=int, default=None, help="aircraft per team for the NvN tasks (2 or 4)")
    ap.add_argument("--hierarchical", action="store_true", help="[3,5,3] actions through the low-level controller kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 legs (N=1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-process rehearsal on a one-GPU box: every rank uses cuda:0 and the barrier / max-over-ranks go "
                         "through gloo (RCCL refuses two ranks on one device); never a measurement")
    ap.add_argument("--stub-env", action="store_true", help="control-flow rehearsal without a GPU (gloo, stand-in handle); never a measurement")
    ap.add_argument("--no-saturating", action="store_true", help="skip the extra 2^20-aircraft leg (N=1 only)")
    ap.add_argument("--checksum-calls", type=int, default=0,
                    help="after the timed region launch the read-only state digest kernel this many times (a dispatch with a known "
                         "byte count in the step kernel's access pattern, used to calibrate FETCH_SIZE under rocprofv3 --pmc)")
    ap.add_argument("--host-only", action="store_true", help="profiling runs: the host-boundary leg alone (no device-resident leg, no further legs)")
    ap.add_argument("--device-only", action="store_true", help="profiling runs: skip the host-boundary leg (value = device-resident rate)")
    args = ap.parse_args()

    import numpy as np
    import aircombat_selfplay_amd as pkg

    rank, world, local_rank = pkg.sharding.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch = None
    if not args.stub_env:
        import torch
        if args.rehearse_on_one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
    cpu_group = args.rehearse_on_one_gpu or args.stub_env
    dist = pkg.sharding.init_process_group("gloo" if cpu_group else "nccl")   # RCCL; only the timing barrier / max-over-ranks use it
    red_dev = "cpu" if cpu_group else f"cuda:{local_rank}"

    E = args.envs
    # env i of the whole job keeps the reference's seed + 1000 i (train_jsbsim.py:33): a rank's block starts at env rank * E
    start, _ = pkg.sharding.env_block(rank, world, world * E)
    seed = 1 + 1000 * start
    if args.stub_env:
        cfg, env = None, StubVecEnv(None, E, seed=seed)
    else:
        if args.task == "heading":
            cfg = pkg.default_config("heading")
        elif args.per_side and args.per_side > 1:
            cfg = pkg.default_nvn_config(args.per_side, task=args.task, hierarchical=args.hierarchical)
        else:
            cfg = pkg.default_config(args.task, hierarchical=args.hierarchical)
        cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
        env = cls(cfg, E, device_id=local_rank, seed=seed)
    A = env.num_agents
    env.reset()

    # ---- synthetic inputs: a pool of random action batches (host arrays for the boundary leg, HBM copies for the device leg)
    rng = np.random.default_rng(20250321 + rank)
    pool = action_pool(np, rng, env, 64 if not args.stub_env else 4)

    def sync_all():
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize()

    # ---- setup, not part of W or K: bring the device to its running clocks (a run as short as the driver's --steps 20 --warmup 5 is
    # over in a millisecond, before the power state has followed), then put every env back to its initial state
    SETUP_STEPS = 0 if args.stub_env else 300
    dev = None
    if not args.stub_env:
        dev = [torch.from_numpy(a).cuda(local_rank) for a in pool]
        torch.cuda.synchronize()
    for i in range(SETUP_STEPS):
        if args.device_only:   # (profiling runs: every launch of the process is a device-resident one)
            env.step_device(dev[i % len(dev)].data_ptr())
        else:
            env.step(pool[i % len(pool)])
    if SETUP_STEPS:
        env.sync()
        env.reset()

    # ---- headline: VecEnv.step(numpy) at the Python boundary; W untimed, then exactly K timed steps between barriers
    kernel_ms = dev_wall = None
    if not args.device_only:
        elapsed = host_leg(env, pool, args.steps, args.warmup, sync_all)
    if args.host_only:   # profiling runs: every launch of the process is a host-boundary launch
        print(json.dumps({"metric": "agent-steps/sec", "value": float(world) * E * A * args.steps / elapsed, "unit": "agent-steps/s",
                          "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                          "config": {"task": args.task, "envs_per_gpu": E, "boundary": "VecEnv.step(numpy) (--host-only: no other leg)"}}))
        env.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    if not args.stub_env:
        sync_all()
        dev_wall, kernel_ms = device_leg(env, [t.data_ptr() for t in dev], args.steps, args.warmup)
        sync_all()
        if args.device_only:
            elapsed = dev_wall
    red = pkg.sharding.max_over_ranks([elapsed, dev_wall or 0.0, kernel_ms or 0.0], dist, device=red_dev)
    elapsed, dev_wall, kernel_ms = red

    if not args.stub_env:
        for _ in range(args.checksum_calls):
            env.state_checksum()
        _, _, _, _, info = env.device_tensors()
        info_h = info.cpu().numpy()

    result = None
    if rank == 0:
        agent_steps = float(world) * E * A * args.steps
        workload = ("SingleCombat 1v1 self-play (no weapons), BASELINE configs[1]" if args.task == "singlecombat" and not args.hierarchical
                    else f"{args.task}{' (hierarchical)' if args.hierarchical else ''}")
        result = {
            "metric": "agent-steps/sec", "value": agent_steps / elapsed, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.stub_env else "stub (control-flow rehearsal, not a measurement)",
            "config": {"workload": workload, "task": args.task, "envs_per_gpu": E, "aircraft_per_env": A, "fdm_ticks_per_step": 6,
                       "boundary": "VecEnv.step(numpy) -> numpy, PCIe inclusive (SURVEY 8d)" if not args.device_only else "device-resident (--device-only)",
                       "actions": "uniform random MultiDiscrete[41,41,41,30], a new host batch every step",
                       "auto_reset": True, "parallelism": f"env-block x{world}", "seed_of_rank0_block": seed,
                       "setup_steps": SETUP_STEPS},
        }
        if not args.stub_env:
            result["device_resident"] = {"value": agent_steps / dev_wall, "unit": "agent-steps/s", "ms_per_step": dev_wall / args.steps * 1e3,
                                         "note": "same steps with actions and outputs resident in HBM (step_device, SURVEY N2), launched back to back"}
            result["roofline"] = roofline(env, args.task, kernel_ms, args.hierarchical)
            if not args.device_only and not args.hierarchical and hasattr(env, "_sets"):
                result["roofline"]["kernel_ms_host_boundary"] = host_launch_ms(env, pool)
                result["roofline"]["note"] = ("kernel_ms: launches of the device-resident leg (actions and outputs in HBM), HIP events over that leg; "
                                              "kernel_ms_host_boundary: the same kernel launched by VecEnv.step (actions read from and a second copy of the "
                                              "outputs written to mapped host memory, across PCIe), HIP events around each launch of a separate 200-step leg")
            result["episode_check"] = {"max_current_step": int(info_h[:, 0].max()), "envs_reset_last_step": int(info_h[:, 3].sum())}
        else:
            result["stub"] = {"steps_taken": env.steps_taken, "seed": env.seed_value}

    if world == 1 and not args.stub_env:
        rate = E * A / (kernel_ms * 1e-3)
        result["roofline"]["valu"] = {"achieved": VALU_PER_AGENT_STEP * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s",
                                      "frac": VALU_PER_AGENT_STEP * rate / 1e12 / VALU_PEAK_TINST,
                                      "note": "lane-instructions of the one-wave kernel form (the algorithm's count) over the measured kernel time"}
        plain = args.task == "singlecombat" and not args.hierarchical
        if plain and not args.device_only:
            # SURVEY 8(d)'s second, "benign" run: the reference's straight-fly action [20, 18.6 -> 19, 20, 0] (baseline.py:168) held in
            # every env, so no aircraft crashes early and the timed mix is all level flight
            env.reset()
            hold = np.tile(np.array([20, 19, 20, 0], dtype=np.float32), (E, A, 1))
            bn = host_leg(env, [hold], 500, 50)
            result["benign_actions"] = {"value": E * A * 500 / bn, "unit": "agent-steps/s", "ms_per_step": bn / 500 * 1e3,
                                        "note": "same boundary, every aircraft holds the straight-fly action: no early crashes in the mix"}
        env.close()
        if plain and not args.no_configs:
            result["configs"] = [
                config_leg(pkg, np, torch, "C3 SingleCombat 1v1 shoot-missile", "singlecombat_shoot", 1, ENVS_PER_GPU, local_rank),
                config_leg(pkg, np, torch, "C3 Scenario1 (gun, AIM-9M, AIM-120B, chaff)", "scenario1", 1, ENVS_PER_GPU, local_rank),
                config_leg(pkg, np, torch, "C4 Scenario2_NvN 2v2 (8192 envs over 2 GPUs)", "scenario_nvn", 2, ENVS_PER_GPU, local_rank),
                config_leg(pkg, np, torch, "C4 legacy MultipleCombat 2v2", "multiplecombat", 2, ENVS_PER_GPU, local_rank),
                config_leg(pkg, np, torch, "C5 Scenario3_NvN 4v4 (32768 envs over 8 GPUs)", "scenario_nvn", 4, ENVS_PER_GPU, local_rank),
            ]
        if plain and not args.no_saturating:
            result["saturating"] = saturating_leg(pkg, np, torch, cfg, local_rank)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(np, cfg, E, A)
    if not env.closed:
        env.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
