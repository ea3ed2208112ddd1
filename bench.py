#!/usr/bin/env python3
"""bench.py — agent-steps/sec of the vectorised air-combat step() on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it with
``python -m torch.distributed.run --nproc-per-node N``: one rank per GPU, weak scaling (every rank owns its own
contiguous block of 4096 envs, no collective in the env path — envs are independent). Rank 0 prints ONE JSON line.

Workload at every N by default: BASELINE.json configs[1] — SingleCombat 1v1 self-play, no weapons, 4096 envs per GPU, uniform random
integer actions regenerated every step (worst case for FCS activity), auto-reset on. A "step" is one pass of the hot path
over the whole batch: 6 FDM ticks per aircraft + observation / reward / termination, one kernel launch.
``--task scenario_nvn --per-side 2|4 --hierarchical`` runs BASELINE's multi-GPU configs instead (C4: Scenario2_NvN 2v2, C5: Scenario3_NvN
4v4, as shipped: [3,5,3] + four weapon bits through the low-level controller kernel), same contract, 4096 envs per GPU.

``value`` is SURVEY 8(d)'s metric: agent-steps per second of ``VecEnv.step(numpy actions) -> numpy obs / rewards / dones / infos`` at
the Python boundary, every ctypes call and every byte that crosses PCIe included, with the VecEnv in its DEFAULT mode (``copy=True``:
step() returns arrays the caller owns, like the reference's np.stack -- handed out without a copy from a ring of page-locked result
sets that is only ever stepped into where the caller has dropped the previous arrays; ``value_mode`` says so in the line, and
``zero_copy_views_mode`` reports the same leg with ``copy=False``, the unchecked views of two alternating sets). The same step with the actions and the outputs resident in HBM (``step_device``) is reported beside it as
``device_resident``; the ``roofline`` of the dominant kernel is measured on that back-to-back device-resident leg with HIP events on
the launch stream. At N = 1 the line also carries ``configs`` (BASELINE C3 / C4 / C5 as shipped = hierarchical, and in the
control-index form, one GPU's 4096-env shard each), a steady-state leg, a saturating-batch leg and the CPU baseline in SURVEY 8(d)'s
two shapes -- which is measured FIRST, before torch or the HIP runtime are touched (its worker processes are forked from a GPU-free parent).
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
# SQ_INSTS_VALU per aircraft-step of the SingleCombat kernel (profiles/round4_pmc_mix.txt, tools/pmc_mix.sh): the one-wave form executes
# the algorithm once per lane; the three-wave form of the BASELINE batch executes the same tick cut in three plus the mailbox traffic
VALU_PER_AGENT_STEP = {"one_wave": 8325.0, "three_wave": 9343.0}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def action_bytes(env):
    """SURVEY 8(d) prices the action row at 16-28 B: the four control indices (16 B: C2 593, C3 617, C4 689, C5 785 B per agent-step are
    quoted with that; the weapon bits of the control-index form are not priced), or the as-shipped hierarchical row of 4 * act_dim
    ([3,5,3] + up to four weapon bits: 12-28 B, floored at 16)."""
    return float(min(28, max(16, 4 * env.act_dim))) if env.hierarchical else 16.0


def algorithmic_bytes(env, missiles_in_flight=0.0):
    """512 B state + action (16-28 B) + 4 * obs_dim + reward + done per agent-step (+ 192 B per live missile-step)."""
    return 512.0 + action_bytes(env) + 4.0 * env.obs_dim + 5.0 + 192.0 * missiles_in_flight


def boundary_bytes(env):
    """Bytes one VecEnv.step moves across PCIe: the action batch in; observation rows, rewards, done flags and one info word per env out."""
    n = env.num_envs * env.num_agents
    return {"actions_in": 4 * env.act_dim * n, "outputs": (4 * env.obs_dim + 4 + 1) * n + 4 * env.num_envs}


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


def pmc_traffic(task, envs, aircraft, hierarchical=False):
    """HBM bytes per launch of the step kernel from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    runs, FETCH_SIZE calibrated on the digest kernel's known byte count; tools/pmc_traffic.py writes the summary). Counters cannot
    be read from inside the process: this is the COMMITTED measurement for the same workload (newest round), or null when there is none."""
    try:
        rec = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None
    best = None
    for r in rec.get("runs", []):
        if (r.get("task") == task and r.get("envs_per_gpu") == envs and r.get("aircraft") == aircraft and bool(r.get("hierarchical", False)) == bool(hierarchical)
                and r.get("round", 1) >= 2 and (best is None or r.get("round", 1) >= best.get("round", 1))):
            best = r
    return best["traffic_bytes_per_launch"] if best else None


def host_leg(env, pool, steps, warmup, sync_all=None):
    """VecEnv.step(numpy) at the Python boundary: `warmup` untimed steps, then EXACTLY `steps` timed ones."""
    n = len(pool)
    for i in range(warmup):
        env.step(pool[i % n])
    if sync_all:
        sync_all()
    res = None
    t0 = time.perf_counter()
    for i in range(steps):
        res = env.step(pool[(warmup + i) % n])     # (kept in a name like a rollout loop does: step t's arrays are alive while step t + 1 runs)
    t1 = time.perf_counter()
    del res
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
    return wall, ev.value / steps      # seconds of the loop, kernel ms per launch (controller + step kernel for a hierarchical handle)


def per_kernel_ms(env, dev_ptrs, steps=200):
    """Controller kernel and step kernel of a device-resident step separately: HIP events around each on the launch stream
    (ac_step_timed_device), one step at a time on a short leg of its own."""
    ctl, stp = C.c_float(), C.c_float()
    tc = ts = 0.0
    for i in range(steps + 20):
        env.lib.check(env.lib.ac_step_timed_device(env._h, dev_ptrs[i % len(dev_ptrs)], C.byref(ctl), C.byref(stp)), "ac_step_timed_device")
        if i >= 20:
            tc += ctl.value
            ts += stp.value
    return tc / steps, ts / steps


def host_launch_ms(env, pool, steps=200):
    """Duration of the step's kernels when they are launched the way VecEnv.step launches them (actions read from, outputs also written
    to, mapped host memory): HIP events on the launch stream around every launch of a short leg of its own, outside the timed region
    (two event records per step would cost the host leg what they measure)."""
    import numpy as np
    tot = 0.0
    ev = C.c_float()
    dll = env.lib.dll
    for i in range(steps + 20):
        st, _ = env._hand_over(pool[i % len(pool)])      # the set VecEnv.step would pick, actions copied into its mapped buffer
        env.lib.check(env.lib.ac_timing_begin(env._h), "ac_timing_begin")
        env.lib.check(dll.ac_step_host_async(env._h, st["index"]), "ac_step_host_async")
        env.lib.check(env.lib.ac_timing_end(env._h, C.byref(ev)), "ac_timing_end")
        if i >= 20:
            tot += ev.value
    return tot / steps


def munitions_per_aircraft(env, dev_ptrs, steps=100, every=10):
    """Average number of munitions in flight per aircraft over a short device-resident leg (ac_munitions_in_flight, sampled every
    `every` steps): SURVEY 8(d) prices each live missile-step at 192 algorithmic bytes."""
    if not hasattr(env, "munitions_in_flight") or env.config.task in (0, 1, 4, 7, 8):     # heading, singlecombat, multiplecombat, wvr, maneuver: no munitions
        return 0.0
    tot = cnt = 0
    for i in range(steps):
        env.step_device(dev_ptrs[i % len(dev_ptrs)])
        if i % every == every - 1:
            tot += env.munitions_in_flight()
            cnt += 1
    return tot / max(cnt, 1) / (env.num_envs * env.num_agents)


def roofline(env, task, kernel_ms, hierarchical=False, controller_ms=None, step_ms=None, missiles=0.0):
    """HBM roofline of the launch: algorithmic bytes (SURVEY 8d accounting, incl. 192 B per live missile-step when `missiles` = munitions in
    flight per aircraft is given) over the measured kernel time. For a hierarchical handle the launch is controller kernel + step kernel;
    `step_kernel` prices the step kernel alone and `controller` reports the controller's matrix rate (its bound is the matrix pipe, not
    HBM: DESIGN.md section 5)."""
    algo = algorithmic_bytes(env, missiles) * env.num_envs * env.num_agents  # per launch, one GPU
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    traffic = pmc_traffic(task, env.num_envs, env.num_envs * env.num_agents, hierarchical)
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "traffic": traffic, "traffic_committed_pmc": traffic,
           "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this workload (profiles/pmc_traffic.json); NOT measured in this run",
           "kernel": "step kernel of the task" + (" + controller8_kernel" if hierarchical else ""), "kernel_ms": kernel_ms,
           "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_agent_step": algorithmic_bytes(env, missiles),
           "munitions_in_flight_per_aircraft": missiles}
    if traffic:
        out["traffic_over_algorithmic"] = traffic / algo
        if hierarchical:
            # SURVEY 8(d)'s per-unit figure has no term for the controller; what it adds per aircraft-step: the GRU state 512 B in + 512 B out,
            # and the 544 KB weight pack that each of the 8 XCDs' L2 fetches once per launch (profiles/pmc_traffic.json)
            n = env.num_envs * env.num_agents
            extra = 1024.0 * n + 8 * 544.0 * 1024.0
            out["traffic_over_algorithmic_with_controller_state"] = traffic / (algo + extra)
            out["traffic_note"] += "; as shipped the launch includes controller8_kernel, whose GRU state (1024 B per aircraft-step) and weight pack (544 KB per XCD and launch) SURVEY 8(d)'s figure does not price"
    if step_ms is not None:
        a = algo / (step_ms * 1e-3) / 1e9
        out["step_kernel"] = {"kernel_ms": step_ms, "achieved": a, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS}
    if hierarchical and controller_ms:
        flop = 2.0 * 137753.0 * env.num_envs * env.num_agents       # BaselineActor: 137 753 multiply-adds per aircraft (weights in baseline_actor.f32)
        out["controller"] = {"kernel_ms": controller_ms, "bound": "L1 fill of the weight stream (fp32 products as two fp16 pieces on the f16 MFMA)", "achieved": flop / (controller_ms * 1e-3) / 1e12,
                             "unit": "TFLOP/s fp32-equivalent", "flop_per_call": flop}
    return out


def config_leg(pkg, np, torch, name, task, per_side, envs, device_id, hierarchical=False, steps=300, warmup=60):
    """One more BASELINE config on this GPU (one GPU's 4096-env shard of it): host-boundary rate, device-resident rate, kernel times,
    roofline, and how much of the boundary step is the step's bytes crossing PCIe."""
    cfg = pkg.default_config(task, hierarchical=hierarchical) if per_side == 1 else pkg.default_nvn_config(per_side, task=task, hierarchical=hierarchical)
    cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
    env = cls(cfg, envs, device_id=device_id, seed=1)       # default mode: arrays the caller owns (ring of page-locked result sets)
    env.reset()
    rng = np.random.default_rng(20250321)
    pool = action_pool(np, rng, env, 16)
    dev = [torch.from_numpy(a).cuda(device_id) for a in pool]
    ptrs = [t.data_ptr() for t in dev]
    torch.cuda.synchronize()
    hb = host_leg(env, pool, steps, warmup)
    wall, kernel_ms = device_leg(env, ptrs, steps, warmup)
    missiles = munitions_per_aircraft(env, ptrs)       # (right after the device-resident leg: the mix that leg ran with)
    ctl_ms, stp_ms = per_kernel_ms(env, ptrs, 100)
    hb_kernel_ms = host_launch_ms(env, pool, 100)
    n = env.num_envs * env.num_agents
    bb = boundary_bytes(env)
    out = {"config": name, "task": task, "action_form": "hierarchical [3,5,3] (+ weapon bits) through the controller kernel (as shipped)" if hierarchical else "control indices [41,41,41,30] (+ weapon bits), no controller launch",
           "envs_per_gpu": envs, "aircraft_per_env": env.num_agents, "obs_dim": env.obs_dim, "act_dim": env.act_dim,
           "value": n * steps / hb, "unit": "agent-steps/s", "ms_per_step": hb / steps * 1e3, "steps": steps, "warmup": warmup,
           "device_resident": {"value": n * steps / wall, "ms_per_step": wall / steps * 1e3},
           "controller_ms": ctl_ms if hierarchical else None, "step_ms": stp_ms,
           "output_bytes": bb["outputs"], "action_bytes": bb["actions_in"],
           "kernel_ms_host_boundary": hb_kernel_ms,
           "pcie_bound_frac": max(0.0, hb_kernel_ms - kernel_ms) / (hb / steps * 1e3),
           "pcie_note": "share of the host-boundary step spent with the step's action and output bytes crossing PCIe = (kernel time with mapped host buffers - "
                        "kernel time with everything in HBM) / ms_per_step; the host-boundary `value` of this config is a PCIe number to that extent, not a kernel number",
           "roofline": roofline(env, task, kernel_ms, hierarchical, ctl_ms, stp_ms, missiles)}
    env.close()
    return out


def saturating_leg(pkg, np, torch, cfg, device_id, envs=524288, steps=40, warmup=8):
    """SURVEY 8d asks for the same path at a saturating batch (>= 2^20 aircraft) beside the BASELINE batch."""
    env = pkg.HipVecEnv(cfg, envs, device_id=device_id, seed=7, copy=False)
    env.reset()
    rng = np.random.default_rng(99)
    pool = [torch.from_numpy(a).cuda(device_id) for a in action_pool(np, rng, env, 4)]
    wall, kernel_ms = device_leg(env, [t.data_ptr() for t in pool], steps, warmup)
    n = env.num_envs * env.num_agents
    bytes_ = algorithmic_bytes(env)
    env.close()
    rate = n / (kernel_ms * 1e-3)
    valu = VALU_PER_AGENT_STEP["one_wave"]
    return {"envs": envs, "aircraft": n, "steps": steps, "value": n * steps / wall, "unit": "agent-steps/s", "kernel_ms": kernel_ms,
            "hbm": {"achieved": bytes_ * rate / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": bytes_ * rate / 1e9 / HBM_PEAK_GBPS},
            "valu": {"achieved": valu * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s", "frac": valu * rate / 1e12 / VALU_PEAK_TINST}}


def host_cores():
    """CPU share of this process: the affinity mask, narrowed by a cgroup CPU quota when there is one (a one-GPU box of the pool is a
    16-core share of a 256-core host)."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max") and txt[0] != "max":
                quota = max(1, int(int(txt[0]) / int(txt[1])))
            elif path.endswith("cfs_quota_us") and int(txt[0]) > 0:
                quota = max(1, int(int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return affinity, quota


def cpu_baseline(np, cfg, envs, agents, seconds_target=8.0):
    """the CPU restatement (oracle/, f64 C: the reference's own SubprocVecEnv + jsbsim wheel cannot run anywhere in this
    pipeline) behind the VecEnv surface in the two shapes named there, same config and action distribution as the GPU run, timed on
    this box's host cores on a bounded sample: (i) every env in one process on one thread at E = 32 (the authors' setting);
    (ii) one worker process per host core, each owning a block of the 4096 envs, pipes + pickle like env_wrappers.py:182-320.
    Called BEFORE torch / the HIP runtime are imported: the workers are forked from a GPU-free parent."""
    assert "torch" not in sys.modules, "cpu_baseline must run before torch / HIP are initialised (its workers are forked)"
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
    affinity, quota = host_cores()
    workers = max(1, min(affinity, quota or affinity, 64))
    sub = OracleSubprocVecEnv(ocfg, envs, workers)
    kw, sw = timed(sub, actions(envs), seconds_target * 0.6)
    sub.close()
    return {"value": envs * agents * kw / sw, "unit": "agent-steps/s", "cores": workers, "kind": "port",
            "host_cpu_count": os.cpu_count(), "sched_affinity_cores": affinity, "cgroup_cpu_quota_cores": quota,
            "measured": "first, before torch / the HIP runtime were imported (workers forked from a GPU-free parent)",
            "single_thread": {"value": 32 * agents * k1 / s1, "unit": "agent-steps/s", "envs": 32, "cores": 1,
                              "sample": f"32 envs x {agents} aircraft x {k1} env steps in {s1:.1f} s, one process, one thread"},
            "sample": f"{envs} envs x {agents} aircraft x {kw} env steps in {sw:.1f} s: {workers} worker processes (sched_getaffinity = {affinity} cores, "
                      f"cgroup quota = {quota}, os.cpu_count() = {os.cpu_count()}), each stepping its block of envs per ('step', actions) message over a pipe, pickled "
                      f"numpy arrays both ways, auto-reset in the worker, parent concatenates (shape of envs/env_wrappers.py:182-320); "
                      f"uniform random actions, oracle/ = f64 C restatement of the JSBSim + Python path"}


class StubVecEnv:
    """--stub-env: a stand-in handle without a GPU, for the multi-rank control-flow rehearsal under gloo (tests/test_sharding_gloo.py):
    barrier, exactly-K timed steps, max-over-ranks, rank-0 print, distinct seed blocks. Never a measurement."""

    def __init__(self, cfg, num_envs, device_id=0, seed=0, agents=2, hierarchical=False):
        self.num_envs, self.num_agents, self.obs_dim = num_envs, agents, (15 if agents == 2 else 9 + 6 * agents + 6)
        self.hierarchical = hierarchical
        self.act_dim = (3 if agents == 2 else 7) if hierarchical else (4 if agents == 2 else 8)
        self.seed_value, self.steps_taken, self.closed = seed, 0, False

    def reset(self):
        return None

    def step(self, actions):
        assert actions.shape == (self.num_envs, self.num_agents, self.act_dim)
        self.steps_taken += 1
        time.sleep(2e-4)

    def close(self):
        self.closed = True


def build_config(pkg, args):
    if args.task == "heading":
        return pkg.default_config("heading")
    if args.per_side and args.per_side > 1:
        return pkg.default_nvn_config(args.per_side, task=args.task, hierarchical=args.hierarchical)
    return pkg.default_config(args.task, hierarchical=args.hierarchical)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (default: the BASELINE config)")
    ap.add_argument("--task", default="singlecombat", help="any name of aircombat_selfplay_amd.config.TASK_IDS (default: BASELINE configs[1])")
    ap.add_argument("--per-side", type=int, default=None, help="aircraft per team for the NvN tasks (2 = BASELINE C4, 4 = C5)")
    ap.add_argument("--hierarchical", action="store_true", help="[3,5,3] (+ weapon bits) actions through the low-level controller kernel: the scenario tasks as shipped")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 legs (N=1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-process rehearsal on a one-GPU box: every rank uses cuda:0 and the barrier / max-over-ranks go "
                         "through gloo (RCCL refuses two ranks on one device); never a measurement")
    ap.add_argument("--stub-env", action="store_true", help="control-flow rehearsal without a GPU (gloo, stand-in handle); never a measurement")
    ap.add_argument("--no-saturating", action="store_true", help="skip the extra 2^20-aircraft leg (N=1 only)")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the steady-state leg (N=1 only)")
    ap.add_argument("--checksum-calls", type=int, default=0,
                    help="after the timed region launch the read-only state digest kernel this many times (a dispatch with a known "
                         "byte count in the step kernel's access pattern, used to calibrate FETCH_SIZE under rocprofv3 --pmc)")
    ap.add_argument("--host-only", action="store_true", help="profiling runs: the host-boundary leg alone (no device-resident leg, no further legs)")
    ap.add_argument("--device-only", action="store_true", help="profiling runs: skip the host-boundary leg (value = device-resident rate)")
    args = ap.parse_args()

    import numpy as np
    import aircombat_selfplay_amd as pkg     # (ctypes structs and host code only: neither torch nor the HIP library is loaded by the import)

    rank, world, local_rank = pkg.sharding.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    E = args.envs
    cfg = None if args.stub_env else build_config(pkg, args)
    profiling = args.host_only or args.device_only
    plain = args.task == "singlecombat" and not args.hierarchical

    # ---- the CPU baseline comes FIRST (rank 0 of a one-GPU run): nothing GPU-side exists yet in this process
    cpu = None
    if world == 1 and not args.stub_env and not args.no_cpu_baseline and not profiling:
        base_cfg = cfg if plain else pkg.default_config("singlecombat")   # the CPU baseline is quoted on the headline config
        cpu = cpu_baseline(np, base_cfg, ENVS_PER_GPU, 2)

    torch = None
    if not args.stub_env:
        import torch
        if args.rehearse_on_one_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
    cpu_group = args.rehearse_on_one_gpu or args.stub_env
    dist = pkg.sharding.init_process_group("gloo" if cpu_group else "nccl")   # RCCL; only the timing barrier / max-over-ranks use it
    red_dev = "cpu" if cpu_group else f"cuda:{local_rank}"

    # env i of the whole job keeps the reference's seed + 1000 i (train_jsbsim.py:33): a rank's block starts at env rank * E
    start, _ = pkg.sharding.env_block(rank, world, world * E)
    seed = 1 + 1000 * start
    if args.stub_env:
        env = StubVecEnv(None, E, seed=seed, agents=2 * (args.per_side or 1), hierarchical=args.hierarchical)
    else:
        cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
        env = cls(cfg, E, device_id=local_rank, seed=seed)     # default mode (copy=True): what a caller who reads INTEGRATION.md section 3 gets
    A = env.num_agents
    env.reset()

    # ---- synthetic inputs: a pool of random action batches (host arrays for the boundary leg, HBM copies for the device leg)
    rng = np.random.default_rng(20250321 + rank)
    pool = action_pool(np, rng, env, (64 if A <= 2 else 16) if not args.stub_env else 4)

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
    ptrs = None
    if not args.stub_env:
        ptrs = [t.data_ptr() for t in dev]
        sync_all()
        dev_wall, kernel_ms = device_leg(env, ptrs, args.steps, args.warmup)
        sync_all()
        if args.device_only:
            elapsed = dev_wall
    red = pkg.sharding.max_over_ranks([elapsed, dev_wall or 0.0, kernel_ms or 0.0], dist, device=red_dev)
    elapsed, dev_wall, kernel_ms = red
    # who took part: one SUM all-reduce on the job's group (RCCL for N > 1 on GPUs) with every rank's device ordinal, PCI bus id and work
    pci = 0
    if torch is not None:
        try:
            pci = int(torch.cuda.get_device_properties(local_rank).pci_bus_id)
        except (AttributeError, RuntimeError):
            pci = 0
    census = pkg.sharding.rank_census(dist, device=red_dev, device_ordinal=local_rank, pci_bus_id=pci, agent_steps=float(E) * env.num_agents * args.steps)

    if not args.stub_env:
        for _ in range(args.checksum_calls):
            env.state_checksum()
        _, _, _, _, info = env.device_tensors()
        info_h = info.cpu().numpy()

    result = None
    if rank == 0:
        agent_steps = float(world) * E * A * args.steps
        if plain:
            workload = "SingleCombat 1v1 self-play (no weapons), BASELINE configs[1]"
        elif args.task == "scenario_nvn" and args.per_side in (2, 4):
            workload = (f"{'Scenario2_NvN 2v2 (BASELINE configs[3], C4)' if args.per_side == 2 else 'Scenario3_NvN 4v4 (BASELINE configs[4], C5)'}, "
                        f"{'as shipped: hierarchical [3,5,3] + four weapon bits' if args.hierarchical else 'control-index action form'}, {E} envs per GPU")
        else:
            workload = f"{args.task}{' (hierarchical)' if args.hierarchical else ''}"
        result = {
            "metric": "agent-steps/sec", "value": agent_steps / elapsed, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.stub_env else "stub (control-flow rehearsal, not a measurement)",
            "value_mode": "default VecEnv (copy=True): step() returns arrays the caller owns, handed out without a copy from a ring of page-locked result sets "
                          "that is only stepped into where the caller has dropped the previous arrays; zero_copy_views_mode = the same leg with copy=False",
            "ranks_reporting": census["ranks_reporting"], "devices": census["devices"], "agent_steps_per_rank": census["agent_steps_per_rank"],
            "collective_backend": None if dist is None else ("gloo" if cpu_group else "nccl (RCCL)"),
            "config": {"workload": workload, "task": args.task, "envs_per_gpu": E, "aircraft_per_env": A, "fdm_ticks_per_step": 6,
                       "hierarchical": bool(args.hierarchical),
                       "boundary": "VecEnv.step(numpy) -> numpy arrays the caller owns (default copy=True), PCIe inclusive (SURVEY 8d)" if not args.device_only else "device-resident (--device-only)",
                       "actions": ("uniform random MultiDiscrete[3,5,3] (+ Bernoulli(0.05) weapon bits)" if args.hierarchical else "uniform random MultiDiscrete[41,41,41,30]") + ", a new host batch every step",
                       "auto_reset": True, "parallelism": f"env-block x{world}", "seed_of_rank0_block": seed,
                       "setup_steps": SETUP_STEPS},
        }
        if not args.stub_env:
            result["device_resident"] = {"value": agent_steps / dev_wall, "unit": "agent-steps/s", "ms_per_step": dev_wall / args.steps * 1e3,
                                         "note": "same steps with actions and outputs resident in HBM (step_device, SURVEY N2), launched back to back"}
            missiles = munitions_per_aircraft(env, ptrs) if world == 1 else 0.0
            ctl_ms, stp_ms = per_kernel_ms(env, ptrs) if world == 1 else (None, None)
            result["roofline"] = roofline(env, args.task, kernel_ms, args.hierarchical, ctl_ms, stp_ms, missiles)
            if not args.device_only and hasattr(env, "_sets") and world == 1:
                hb_ms = host_launch_ms(env, pool)
                bb = boundary_bytes(env)
                result["roofline"]["kernel_ms_host_boundary"] = hb_ms
                result["roofline"]["note"] = ("kernel_ms: launches of the device-resident leg (actions and outputs in HBM), HIP events over that leg; "
                                              "kernel_ms_host_boundary: the same kernel launched by VecEnv.step (actions read from and a second copy of the "
                                              "outputs written to mapped host memory, across PCIe), HIP events around each launch of a separate 200-step leg")
                result["output_bytes"], result["action_bytes"] = bb["outputs"], bb["actions_in"]
                result["pcie_bound_frac"] = max(0.0, hb_ms - kernel_ms) / (elapsed / args.steps * 1e3)
            result["episode_check"] = {"max_current_step": int(info_h[:, 0].max()), "envs_reset_last_step": int(info_h[:, 3].sum())}
        else:
            result["stub"] = {"steps_taken": env.steps_taken, "seed": env.seed_value, "agents": env.num_agents, "act_dim": env.act_dim}

    if world == 1 and not args.stub_env:
        rate = E * A / (kernel_ms * 1e-3)
        if plain:
            valu = VALU_PER_AGENT_STEP["three_wave" if (E * A + 63) // 64 <= 512 else "one_wave"]
            result["roofline"]["valu"] = {"achieved": valu * rate / 1e12, "peak": VALU_PEAK_TINST, "unit": "T lane-inst/s",
                                          "frac": valu * rate / 1e12 / VALU_PEAK_TINST,
                                          "note": "SQ_INSTS_VALU per aircraft-step of the kernel form this batch runs (profiles/round4_pmc_mix.txt) over the measured kernel time"}
        if not args.device_only and args.steps < 200:
            # a K as short as the driver's (20 steps = 0.8 ms) is one sample of a noisy quantity: the same K-step region repeated, median reported
            reps = sorted(host_leg(env, pool, args.steps, 0, sync_all) for _ in range(31))
            med = reps[len(reps) // 2]
            result["repeat_median"] = {"value": E * A * args.steps / med, "ms_per_step": med / args.steps * 1e3, "repeats": len(reps),
                                       "min_ms_per_step": reps[0] / args.steps * 1e3, "max_ms_per_step": reps[-1] / args.steps * 1e3,
                                       "note": f"median over {len(reps)} repeats of the same {args.steps}-step timed region (`value` above is the contract's single region)"}
        if not args.device_only and not args.no_steady_state:
            # steady state: run on from wherever the batch is (no reset) until episode ages are spread out -- with random actions aircraft crash
            # and envs restart at different times -- then time the mix: auto-resets, dead aircraft waiting for their env to end, all ages
            for i in range(max(0, 700 - args.steps - args.warmup)):
                env.step(pool[i % len(pool)])
            resets = dead = 0
            t0 = time.perf_counter()
            SS = 1000
            for i in range(SS):
                out = env.step(pool[i % len(pool)])
                if i % 50 == 0:
                    resets += int((out[-1]._codes >> 31 & 1).sum()) if out[-1]._codes.ndim == 1 else 0
                    dead += int(out[-2].sum())
            ss = time.perf_counter() - t0
            steps_h = (out[-1]._codes & 0xFFFF)
            result["steady_state"] = {"value": E * A * SS / ss, "unit": "agent-steps/s", "ms_per_step": ss / SS * 1e3, "steps": SS,
                                      "envs_reset_per_step": resets / (SS / 50), "done_flags_per_step": dead / (SS / 50),
                                      "episode_age_steps": {"min": int(steps_h.min()), "median": int(np.median(steps_h)), "max": int(steps_h.max())},
                                      "note": "same boundary after >= 700 steps without a reset: episode ages staggered, auto-resets and dead aircraft in the mix (sampled every 50th step)"}
            if plain:
                # SURVEY 8(d)'s second, "benign" run: the reference's straight-fly action [20, 18.6 -> 19, 20, 0] (baseline.py:168) held in
                # every env, so no aircraft crashes early and the timed mix is all level flight
                env.reset()
                hold = np.tile(np.array([20, 19, 20, 0], dtype=np.float32), (E, A, 1))
                bn = host_leg(env, [hold], 500, 50)
                result["benign_actions"] = {"value": E * A * 500 / bn, "unit": "agent-steps/s", "ms_per_step": bn / 500 * 1e3,
                                            "note": "same boundary, every aircraft holds the straight-fly action: no early crashes in the mix"}
                # the same boundary with copy=False: unchecked views of two alternating sets (what `value` was measured with up to round 3)
                vw = pkg.HipVecEnv(cfg, E, device_id=local_rank, seed=seed, copy=False)
                vw.reset()
                zc = host_leg(vw, pool, 500, 50)
                result["zero_copy_views_mode"] = {"value": E * A * 500 / zc, "unit": "agent-steps/s", "ms_per_step": zc / 500 * 1e3,
                                                  "note": "VecEnv(copy=False): step() returns live views of two alternating page-locked sets, no ownership check"}
                vw.close()
                # and a caller that hoards every result (more live result sets than the ring has): step() falls back to fresh copies
                env.reset()
                hoard = []
                for i in range(50):
                    hoard.append(env.step(pool[i % len(pool)]))
                t0 = time.perf_counter()
                for i in range(300):
                    hoard.append(env.step(pool[i % len(pool)]))
                hd = time.perf_counter() - t0
                del hoard
                result["hoarding_caller_copy_fallback"] = {"value": E * A * 300 / hd, "unit": "agent-steps/s", "ms_per_step": hd / 300 * 1e3,
                                                           "note": "a caller that keeps EVERY step's arrays alive: the ring has no free set, step() copies the results out of a staging set "
                                                                   "(the outputs were just written by the GPU, so the copy reads them cache-cold: tools/diag/host_copy_bench.py)"}
        env.close()
        if plain and not args.no_configs and not profiling:
            legs = [("C3 SingleCombat 1v1 shoot-missile", "singlecombat_shoot", 1), ("C3 Scenario1 (gun, AIM-9M, AIM-120B, chaff)", "scenario1", 1),
                    ("C4 Scenario2_NvN 2v2: one GPU's 4096-env shard of the 8192-env config", "scenario_nvn", 2),
                    ("C4 legacy MultipleCombat 2v2", "multiplecombat", 2),
                    ("C5 Scenario3_NvN 4v4: one GPU's 4096-env shard of the 32768-env config", "scenario_nvn", 4)]
            result["configs"] = []
            for name, task, per_side in legs:
                if task in ("scenario1", "scenario_nvn"):    # the scenario tasks as the reference ships them (scenario2_task.py:14,225)
                    result["configs"].append(config_leg(pkg, np, torch, name + " -- AS SHIPPED (hierarchical)", task, per_side, ENVS_PER_GPU, local_rank, hierarchical=True))
                result["configs"].append(config_leg(pkg, np, torch, name + (" -- control-index form" if task in ("scenario1", "scenario_nvn") else ""), task, per_side, ENVS_PER_GPU, local_rank))
        if plain and not args.no_saturating and not profiling:
            result["saturating"] = saturating_leg(pkg, np, torch, cfg, local_rank)
        if cpu is not None:
            result["cpu_baseline"] = cpu
    if not env.closed:
        env.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
