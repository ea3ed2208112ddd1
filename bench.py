#!/usr/bin/env python3
"""bench.py — agent-steps/sec of the vectorised air-combat step() on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it with
``python -m torch.distributed.run --nproc-per-node N``: one rank per GPU, weak scaling (every rank owns its own
contiguous block of 4096 envs, no collective in the env path — envs are independent). Rank 0 prints ONE JSON line.

Workload at every N: BASELINE.json configs[1] — SingleCombat 1v1 self-play, no weapons, 4096 envs per GPU, uniform random
integer actions regenerated every step (worst case for FCS activity), auto-reset on. A "step" is one pass of the hot path
over the whole batch: 6 FDM ticks per aircraft + observation/reward/termination, one kernel launch.
``value`` is measured with the actions already resident in HBM and the outputs left in HBM (SURVEY N2 path); the
host-boundary (PCIe + ctypes + numpy) rate of the strict drop-in ``step()`` is reported beside it as ``host_boundary``.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
AGENTS = 2
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_AGENT_STEP = 593.0    # SURVEY §8(d) / BASELINE.md §3, config C2: 512 state + 16 action + 60 obs + 4 reward + 1 done


def cpu_baseline(cfg, seconds_target=12.0):
    """The CPU port (oracle) timed on this box's host cores, rank 0 at N=1 only, on a bounded sample of the same workload
    (one worker per host core, capped at the 16-core share a one-GPU box gets; the single-core rate beside it)."""
    import threading
    from oracle import oracle as O
    ocfg = O.config_from_ac(cfg)
    n0, s0, _ = O.bench_run(ocfg, 256, 20)                 # calibrate one core
    rate1 = n0 / s0
    threads = max(1, min(16, os.cpu_count() or 1))
    per = ENVS_PER_GPU // threads
    steps = max(10, int(seconds_target * rate1 / (per * AGENTS)))
    out = [None] * threads

    def work(i):
        out[i] = O.bench_run(ocfg, per, steps, seed=20250321 + i)      # ctypes releases the GIL: real threads

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(o[0] for o in out)
    return {"value": total / wall, "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "single_core_value": rate1,
            "sample": f"{per * threads} envs x {AGENTS} aircraft x {steps} env steps split over {threads} threads (one env block each), "
                      f"random actions, auto-reset, {wall:.1f} s wall on {threads} of {os.cpu_count()} host cores (oracle/: f64 C "
                      f"restatement of the JSBSim+Python path; the reference's own SubprocVecEnv+jsbsim wheel cannot run here)"}


VALU_PEAK_TINST = 256 * 4 * 32 * 2.4e9 / 1e12   # lane-instructions/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (157.3 TFLOP/s fp32 = 2 flop FMA)
VALU_PER_AGENT_STEP = 8796.0                   # SQ_INSTS_VALU per wave per launch / 64 lanes x 64 (profiles/round1_pmc_mix.txt)


def saturating_leg(pkg, cfg, local_rank, envs=524288, steps=40, warmup=8):
    """SURVEY 8d asks for the same path at a saturating batch (>= 2^20 aircraft) beside the BASELINE batch: 2 waves per SIMD on
    every CU instead of one wave on an eighth of them."""
    import ctypes as C
    import numpy as np
    import torch
    env = pkg.HipVecEnv(cfg, envs, device_id=local_rank, seed=7)
    env.reset()
    rng = np.random.default_rng(99)
    pool = []
    for _ in range(4):
        a = np.stack([rng.integers(0, n, size=(envs, AGENTS)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32)
        if env.act_dim == 5:
            a = np.concatenate([a, (rng.random((envs, AGENTS, 1)) < 0.05).astype(np.float32)], axis=-1)  # (only reached for 1v1 tasks)
        pool.append(torch.from_numpy(a).cuda(local_rank))
    ptrs = [t.data_ptr() for t in pool]
    for i in range(warmup):
        env.step_device(ptrs[i % 4])
    env.sync()
    env.lib.ac_timing_begin(env._h)
    t0 = time.perf_counter()
    for i in range(steps):
        env.step_device(ptrs[i % 4])
    env.sync()
    wall = time.perf_counter() - t0
    ev = C.c_float()
    env.lib.check(env.lib.ac_timing_end(env._h, C.byref(ev)), "ac_timing_end")
    env.close()
    kernel_s = ev.value * 1e-3 / steps
    rate = envs * AGENTS / kernel_s
    return {"envs": envs, "aircraft": envs * AGENTS, "steps": steps, "value": envs * AGENTS * steps / wall, "unit": "agent-steps/s",
            "kernel_ms": kernel_s * 1e3,
            "hbm": {"achieved": BYTES_PER_AGENT_STEP * rate / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": BYTES_PER_AGENT_STEP * rate / 1e9 / HBM_PEAK_GBPS},
            "valu": {"achieved": VALU_PER_AGENT_STEP * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s",
                     "frac": VALU_PER_AGENT_STEP * rate / 1e12 / VALU_PEAK_TINST}}


def pmc_traffic(task, envs):
    """HBM bytes per launch of the step kernel from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    runs of this same command, FETCH_SIZE calibrated on the digest kernel's known byte count; tools/pmc_traffic.py writes the
    summary). Counters cannot be read from inside the process, so this is the committed measurement for the same workload, or
    null when there is none for it."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    for r in rec.get("runs", []):
        if r.get("task") == task and r.get("envs_per_gpu") == envs:
            return r["traffic_bytes_per_launch"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (default: the BASELINE config)")
    ap.add_argument("--task", default="singlecombat", help="any name of aircombat_selfplay_amd.config.TASK_IDS (default: BASELINE configs[1])")
    ap.add_argument("--hierarchical", action="store_true", help="[3,5,3] actions through the low-level controller kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-process rehearsal on a one-GPU box: every rank uses cuda:0 and the barrier / max-over-ranks go "
                         "through gloo (RCCL refuses two ranks on one device); never a measurement")
    ap.add_argument("--no-saturating", action="store_true", help="skip the extra 2^20-aircraft leg (N=1 only)")
    ap.add_argument("--checksum-calls", type=int, default=0,
                    help="after the timed region launch the read-only state digest kernel this many times (a dispatch with a known "
                         "byte count in the step kernel's access pattern, used to calibrate FETCH_SIZE under rocprofv3 --pmc)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import aircombat_selfplay_amd as pkg

    rank, world, local_rank = pkg.sharding.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RCCL; used only for the timing barrier / max-over-ranks
    dist = pkg.sharding.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl")

    E = args.envs
    cfg = pkg.default_config(args.task, hierarchical=args.hierarchical) if args.task != "heading" else pkg.default_config("heading")
    cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
    env = cls(cfg, E, device_id=local_rank, seed=1 + 1000 * rank)
    global AGENTS, BYTES_PER_AGENT_STEP
    AGENTS = env.num_agents
    # 512 B state + action + 4 * obs_dim + reward + done
    BYTES_PER_AGENT_STEP = 512.0 + 4.0 * env.act_dim + 4.0 * env.obs_dim + 5.0
    nvec = (3, 5, 3) if env.hierarchical else (41, 41, 41, 30)
    env.reset()
    act_dim = env.act_dim

    # ---- synthetic inputs, resident in HBM before the timed region: a pool of random action batches
    rng = np.random.default_rng(20250321 + rank)
    POOL = 64
    pool = []
    for _ in range(POOL):
        a = np.stack([rng.integers(0, n, size=(E, AGENTS)) for n in nvec], axis=-1).astype(np.float32)
        if act_dim > len(nvec):   # shoot bit / the four weapon bits: Bernoulli(0.05)
            a = np.concatenate([a, (rng.random((E, AGENTS, act_dim - len(nvec))) < 0.05).astype(np.float32)], axis=-1)
        pool.append(torch.from_numpy(a).cuda(local_rank))
    ptrs = [t.data_ptr() for t in pool]
    torch.cuda.synchronize()

    def run(k, offset=0):
        for i in range(k):
            env.step_device(ptrs[(offset + i) % POOL])

    run(args.warmup)
    env.sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    env.lib.ac_timing_begin(env._h)
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    env.sync()
    t1 = time.perf_counter()
    import ctypes as C
    ev_ms = C.c_float()
    env.lib.check(env.lib.ac_timing_end(env._h, C.byref(ev_ms)), "ac_timing_end")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    kernel_ms = ev_ms.value / args.steps          # HIP events on the launch stream, average per launch
    elapsed, kernel_ms = pkg.sharding.max_over_ranks([elapsed, kernel_ms], dist,
                                                     device="cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}")

    for _ in range(args.checksum_calls):
        env.state_checksum()

    # sanity: the episode machinery really ran (steps counted, resets happened)
    _, _, _, _, info = env.device_tensors()
    info_h = info.cpu().numpy()

    result = None
    if rank == 0:
        agent_steps = float(world) * E * AGENTS * args.steps
        value = agent_steps / elapsed
        algo_bytes = BYTES_PER_AGENT_STEP * E * AGENTS            # per launch, one GPU
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SingleCombat 1v1 self-play (no weapons), BASELINE configs[1]" if args.task == "singlecombat" and not args.hierarchical
                       else f"{args.task}{' (hierarchical)' if args.hierarchical else ''}", "task": args.task,
                       "envs_per_gpu": E, "aircraft_per_env": AGENTS, "fdm_ticks_per_step": 6,
                       "actions": "uniform random MultiDiscrete[41,41,41,30], new batch every step, device-resident",
                       "auto_reset": True, "parallelism": f"env-block x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.task, E),
                         "kernel": "step kernel of the task (+ controller_kernel when hierarchical)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes},
            "episode_check": {"max_current_step": int(info_h[:, 0].max()), "envs_reset_last_step": int(info_h[:, 3].sum())},
        }

    # ---- the strict drop-in boundary (host numpy in/out, PCIe inclusive) for DESIGN.md; never the headline value
    if world == 1:
        a_host = [p.cpu().numpy() for p in pool[:8]]
        for i in range(20):
            env.step(a_host[i % 8])
        t0 = time.perf_counter()
        HB = 200
        for i in range(HB):
            env.step(a_host[i % 8])
        hb = time.perf_counter() - t0
        # SURVEY 8(d)'s second, "benign" run: the reference's straight-fly action [20, 18.6 -> 19, 20, 0] (baseline.py:168) held in
        # every env, so no aircraft crashes early and the timed mix is all level flight (device-resident, like the headline run)
        if args.task == "singlecombat" and not args.hierarchical:
            env.reset()
            hold = torch.from_numpy(np.tile(np.array([20, 19, 20, 0], dtype=np.float32), (E, AGENTS, 1))).cuda(local_rank)
            for _ in range(50):
                env.step_device(hold.data_ptr())
            env.sync()
            t0 = time.perf_counter()
            BN = 500
            for _ in range(BN):
                env.step_device(hold.data_ptr())
            env.sync()
            bn = time.perf_counter() - t0
            result["benign_actions"] = {"value": E * AGENTS * BN / bn, "unit": "agent-steps/s", "ms_per_step": bn / BN * 1e3,
                                        "note": "same batch, every aircraft holds the straight-fly action: no early crashes in the mix"}
        result["host_boundary"] = {"value": E * AGENTS * HB / hb, "unit": "agent-steps/s", "ms_per_step": hb / HB * 1e3,
                                   "note": "VecEnv.step(numpy) incl. H2D actions, kernel, D2H obs/reward/done and the info codes (dicts are built when read)"}
        rate = E * AGENTS / (kernel_ms * 1e-3)
        result["roofline"]["valu"] = {"achieved": VALU_PER_AGENT_STEP * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s",
                                      "frac": VALU_PER_AGENT_STEP * rate / 1e12 / VALU_PEAK_TINST,
                                      "note": "lane-instructions of the one-wave kernel form (the algorithm's count) over the measured time; a lone "
                                              "wave issues one instruction per ~4 cycles, so at this batch the step time is the length of a lane's "
                                              "instruction stream: %d workgroups x %d wave(s) on 1024 SIMDs" % (
                                                  (E * AGENTS + 63) // 64, 3 if (args.task == "singlecombat" and not args.hierarchical and (E * AGENTS + 63) // 64 <= 512
                                                                                 and os.environ.get("AIRCOMBAT_SPLIT", "1") != "0") else 1)}
        if not args.no_saturating and args.task == "singlecombat" and not args.hierarchical:
            env.close()
            result["saturating"] = saturating_leg(pkg, cfg, local_rank)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg)
    if not env.closed:
        env.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
