"""Per-element comparison rules shared by the GPU parity tests: fp32 kernel outputs against the float64 oracle, every element held
to its own bound (conditioning of acos for the relative angles, of the potential-form PostureReward for the rewards)."""
import numpy as np

TASK_FIELDS = ("bloods", "pre_posture", "pre_altitude", "pre_event", "pre_shoot", "status", "die_flag", "remaining", "pre_remaining",
               "shoot_action", "last_missile", "last_shoot_time", "lock_bits", "lock_pos", "cur_step")


def geometry_blocks(obs_dim):
    """Start columns of the [du, dh, AO, TA, R / 1e4, side] blocks of an observation layout (others ..., then the missile block)."""
    return [9 + 6 * k for k in range((obs_dim - 9) // 6)]


def obs_bounds(want, scale=1.0):
    """Per-element bound of an fp32 observation against the float64 oracle: 2e-4 + 2e-4 |x| (x scale), with the two angles of every
    relative-geometry block weighted by the conditioning of acos (an fp32 rounding of its argument moves the angle by
    eps / sin(angle)), and the side flag free where the cross product that defines it vanishes (dead ahead / astern)."""
    tol = scale * (2e-4 + 2e-4 * np.abs(want))
    free = np.zeros(want.shape, dtype=bool)
    for o in geometry_blocks(want.shape[-1]):
        for col in (o + 2, o + 3):
            tol[..., col] = scale * (2e-4 + 3e-7 / np.maximum(np.sin(want[..., col]), 1e-4))
        free[..., o + 5] = np.sin(want[..., o + 2]) < 2e-3 * scale
    return tol, free


USED = {}   # worst fraction of the bound used, per context label (printed by the tests that pass one: how much room the scale leaves)


def assert_obs(got, want, scale, ctx, label=None):
    tol, free = obs_bounds(want, scale)
    bad = (np.abs(got - want) > tol) & ~free
    assert not bad.any(), (ctx, np.argwhere(bad)[:6].tolist(), got[bad][:6], want[bad][:6], tol[bad][:6])
    if label is not None and got.size:
        frac = np.where(free, 0.0, np.abs(got - want) / tol)
        USED[label] = max(USED.get(label, 0.0), float(frac.max()))


class RewardBound:
    """Per-element reward bound: 5e-3 + 1e-3 |r| (x scale), widened by what the potential-form PostureReward (x posture_scale,
    differenced between consecutive steps) does with an fp32 target angle: d/dTA of atanh(1 - 2 TA / pi) / 2 pi is
    1 / (pi^2 (1 - x^2)), unbounded at TA = pi, and TA itself carries eps / sin(TA). The term is a difference r_t - r_(t-1) whose
    r_(t-1) is the device's own value of the step before, so the widening of the previous step's geometry counts as well."""

    def __init__(self, posture_scale, enemy_blocks_from, n_enemies, scale=1.0):
        self.k, self.o, self.n, self.scale = posture_scale, enemy_blocks_from, n_enemies, scale
        self.prev = None

    def __call__(self, want, robs):
        tol = self.scale * (5e-3 + 1e-3 * np.abs(want))
        extra = np.zeros(want.shape[:-1])
        for k in range(self.n):
            TA = robs[..., self.o + 6 * k + 3]
            x = np.clip(1.0 - 2.0 * TA / np.pi, -0.9999999, 0.9999999)
            dTA = 2e-4 + 3e-7 / np.maximum(np.sin(TA), 1e-4)
            extra += np.where(x < 0, self.k * (2.0 / np.pi) / (2 * np.pi * (1 - x * x)) * dTA * 2.0, 0.0)
        both = extra + (self.prev if self.prev is not None and self.prev.shape == extra.shape else 0.0)
        self.prev = extra
        return tol + self.scale * both[..., None]


def team_max(rt, A):
    """NvN tasks hand every agent its team's mean reward: the widest bound of a team applies to its members."""
    E = rt.shape[0]
    return np.broadcast_to(rt.reshape(E, 2, A // 2, 1).max(axis=2, keepdims=True), (E, 2, A // 2, 1)).reshape(E, A, 1)
