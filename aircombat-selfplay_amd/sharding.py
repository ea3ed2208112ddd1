"""Env-block sharding across GPUs: one process per GPU, a contiguous block of envs per rank, no data-path collective.

The env instances of the reference are independent (per-env curriculum counters, RNG and reward memory), so the
N-GPU path is N replicas of the single-GPU path over disjoint env blocks. torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for the timing barrier and the max-over-ranks of the bench contract.
"""
import os


def env_block(rank, world, n_envs_total):
    """Contiguous block [start, start+count) of rank ``rank``; the first ``n_envs_total % world`` ranks take one extra env."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_envs_total), int(world))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def dist_env():
    """(rank, world, local_rank) from the torch.distributed.run environment; single process when unset."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank, world, _ = dist_env()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist if world > 1 else None


def max_over_ranks(values, dist=None, device="cpu"):
    """Element-wise MAX of a list of floats over all ranks (identity when not distributed)."""
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def barrier(dist=None):
    if dist is not None:
        dist.barrier()


def rank_census(dist=None, device="cpu", device_ordinal=0, pci_bus_id=0, agent_steps=0.0):
    """What makes an N > 1 bench line self-proving: every rank adds a 1, its device ordinal, its GPU's PCI bus id and the agent-steps
    it ran into its own slot of one SUM all-reduce on the job's process group (RCCL on GPUs). Returns {"ranks_reporting": how many
    ranks took part in the collective, "devices": [{"rank", "device", "pci_bus_id"} ...], "agent_steps_per_rank": [...]};
    a rank that never joined leaves its slot empty (and the collective would not have completed)."""
    rank, world, _ = dist_env()
    if dist is None:
        return {"ranks_reporting": 1, "devices": [{"rank": 0, "device": int(device_ordinal), "pci_bus_id": int(pci_bus_id)}],
                "agent_steps_per_rank": [float(agent_steps)]}
    import torch
    t = torch.zeros(world, 4, dtype=torch.float64, device=device)
    t[rank, 0], t[rank, 1], t[rank, 2], t[rank, 3] = 1.0, float(device_ordinal), float(pci_bus_id), float(agent_steps)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu()
    return {"ranks_reporting": int(t[:, 0].sum().item()),
            "devices": [{"rank": r, "device": int(t[r, 1].item()), "pci_bus_id": int(t[r, 2].item())} for r in range(world) if t[r, 0] > 0],
            "agent_steps_per_rank": [float(t[r, 3].item()) for r in range(world)]}
