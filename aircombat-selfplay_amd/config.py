"""Scenario YAML -> ac_config_t.

Mirrors ``parse_config`` (reference envs/JSBSim/utils/utils.py:7-23) and the ``getattr(config, f'{Class}_{param}', default)``
convention of the reward / termination classes (reward_function_base.py:14-15, altitude_reward.py:14-16, low_altitude.py:13,
overload.py:18-20, timeout.py:16), for the tasks the HIP path implements.
"""
import yaml

from .capi import (AcConfig, AC_MAX_AGENTS, AC_TASK_HEADING, AC_TASK_SINGLECOMBAT, AC_TASK_DODGE_MISSILE, AC_TASK_WVR, AC_TASK_MANEUVER,
                   AC_TASK_SHOOT_MISSILE, AC_TASK_MULTICOMBAT, AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN)

TASK_IDS = {
    "heading": AC_TASK_HEADING,
    "approach": AC_TASK_HEADING,                     # ApproachTask: the heading env with AltitudeReward alone and no UnreachHeading
    "singlecombat": AC_TASK_SINGLECOMBAT,            # SingleCombatTask (1v1, no weapons)
    "singlecombat_dodge_missile": AC_TASK_DODGE_MISSILE,
    "singlecombat_shoot": AC_TASK_SHOOT_MISSILE,     # SingleCombatShootMissileTask
    "multiplecombat": AC_TASK_MULTICOMBAT,           # MultipleCombatTask (NvN, no weapons) under MultipleCombatEnv.step
    "scenario1": AC_TASK_SCENARIO1,                  # Scenario1 weapon rules + 11 rewards, 1v1, low-level control
    "scenario_nvn": AC_TASK_SCENARIO_NVN,            # Scenario2_NvN / Scenario3_NvN, low-level control
    "scenario2_nvn": AC_TASK_SCENARIO_NVN,
    "scenario3_nvn": AC_TASK_SCENARIO_NVN,
    "maneuver_lowlevel": AC_TASK_MANEUVER,           # Maneuver_curriculum rules with explicit control indices and the YAML's own spawn
    "wvr_lowlevel": AC_TASK_WVR,                     # WVRTask rules with explicit control indices and the YAML's own spawn
    "hierarchical_singlecombat": AC_TASK_SINGLECOMBAT,        # HierarchicalSingleCombatTask: [3,5,3] through the low-level controller
    "hierarchical_multiplecombat": AC_TASK_MULTICOMBAT,       # HierarchicalMultipleCombatTask
    # HierarchicalMultipleCombatShootTask (multiplecombat_env.py:31-32 -> multiplecombat_with_missile_task.py:206-238): the only
    # missile variant of MultipleCombat an env can select. Its step() launches nothing, so it is MultipleCombat with the 21-value
    # paired-enemy observation and a [3,5,3] + shoot-bit action whose bit is ignored.
    "hierarchical_multiplecombat_shoot": AC_TASK_MULTICOMBAT,
    # MultipleCombatShootMissileTask (multiplecombat_with_missile_task.py:165-216), its parent: the same task with the control-index
    # action [41,41,41,30] + shoot bit. No env of the reference constructs it (multiplecombat_env.py:25-33); the name is this package's own.
    "multiplecombat_shoot": AC_TASK_MULTICOMBAT,
    # MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-145): rule-based launches of the base-class missile at
    # enemies[0] under MultipleCombatEnv.step, the same 21-value observation with a live missile block. Equally unconstructible in the
    # reference; here AC_TASK_DODGE_MISSILE with four or eight aircraft.
    "multiplecombat_dodge_missile": AC_TASK_DODGE_MISSILE,
    "hierarchical_multiplecombat_dodge_missile": AC_TASK_DODGE_MISSILE,   # HierarchicalSMultipleCombatDodgeMissileTask (:148-181): [3,5,3] through the controller
    # HierarchicalSingleCombatShootTask / HierarchicalSingleCombatDodgeMissileTask (singlecombat_with_missile_task.py:126-145,206-238):
    # the 1v1 missile tasks behind the low-level controller. No env of the reference selects them (singlecombat_env.py:19-36); the
    # names are this package's own.
    "hierarchical_singlecombat_shoot": AC_TASK_SHOOT_MISSILE,
    "hierarchical_singlecombat_dodge_missile": AC_TASK_DODGE_MISSILE,
}
# task names whose reference class takes the [3,5,3] (+ weapon bits) action through the low-level controller. The scenario tasks
# are hierarchical in the reference (scenario1_task.py:11, scenario2_task.py:14); config_from_yaml follows that, while
# default_config() keeps the control-index form unless asked (tests drive the weapon rules with explicit controls).
ALWAYS_HIERARCHICAL = ("hierarchical_singlecombat", "hierarchical_multiplecombat", "hierarchical_multiplecombat_shoot",
                       "hierarchical_singlecombat_shoot", "hierarchical_singlecombat_dodge_missile", "hierarchical_multiplecombat_dodge_missile")
HIERARCHICAL_IN_REFERENCE = ALWAYS_HIERARCHICAL + ("scenario1", "scenario2_nvn", "scenario3_nvn", "scenario1_curriculum",
                                                   "scenario2_nvn_curriculum", "scenario3_nvn_curriculum", "scenario1_rwr",
                                                   "scenario2_rwr", "scenario3_rwr", "scenario1_rwr_curriculum",
                                                   "scenario2_rwr_curriculum", "scenario3_rwr_curriculum", "wvr", "maneuver_curriculum",
                                                   "scenario2", "scenario3", "scenario2_curriculum", "scenario3_curriculum")
# The *_curriculum tasks (scenario1_task.py:147-195, scenario2_task.py:318-383) respawn the aircraft from
# env.reset_simulators_curriculum(curriculum_angle) at every reset. The angle is meant to grow with the ego win rate, but it never
# does in the reference: the 1v1 tasks AND `success` over conditions that start with LowAltitude (always False), and every task
# pops its record back to 20 entries while the advance needs len(record) > 20. So these tasks are the base task with the
# angle-0 spawn; `curriculum_angle` in the scenario dict selects another fixed angle for users who want one.
CURRICULUM_BASE = {"maneuver_curriculum": "maneuver_lowlevel",
                   "wvr": "wvr_lowlevel",   # WVRTask.reset always respawns from reset_simulators_curriculum (WVR_task.py:41-46)
                   "scenario1_curriculum": "scenario1", "scenario2_nvn_curriculum": "scenario2_nvn",
                   "scenario3_nvn_curriculum": "scenario3_nvn", "scenario2_curriculum": "scenario2", "scenario3_curriculum": "scenario3",
                   "scenario1_rwr_curriculum": "scenario1_rwr",
                   "scenario2_rwr_curriculum": "scenario2_rwr", "scenario3_rwr_curriculum": "scenario3_rwr"}
# *_RWR variants (scenario1_task.py:197-314, scenario2_task.py:385-476): the base task with two reserved observation slots
RWR_BASE = {"scenario1_rwr": "scenario1", "scenario2_rwr": "scenario2_nvn", "scenario3_rwr": "scenario3_nvn"}
# Scenario2 / Scenario3 themselves (scenario2_task.py:14-157): the NvN rules with the older 21-value paired-enemy observation
LEGACY_OBS_BASE = {"scenario2": "scenario2_nvn", "scenario3": "scenario3_nvn"}


def curriculum_spawn(center_lat, center_lon, radius_km, angle_deg):
    """calculate_coordinates_heading_by_curriculum (utils/utils.py:126-156) for one angle: a point on a circle about the centre
    (0 deg = due south, counter-clockwise) and the heading to fly there; spherical Earth of 6371 km."""
    import math
    d = radius_km / 6371.0
    la, lo = math.radians(center_lat), math.radians(center_lon)
    th = math.radians(180 - angle_deg)
    nla = math.asin(math.sin(la) * math.cos(d) + math.cos(la) * math.sin(d) * math.cos(th))
    nlo = lo + math.atan2(math.sin(th) * math.sin(d) * math.cos(la), math.cos(d) - math.sin(la) * math.sin(nla))
    hdg = 2 * angle_deg if 0 <= angle_deg < 90 else 360 - 2 * angle_deg
    return math.degrees(nla), math.degrees(nlo), hdg


def apply_curriculum_spawn(cfg, angle=0):
    """reset_simulators_curriculum (singlecombat_env.py:87-122 for two aircraft, multiplecombat_env.py:185-248 otherwise, which
    rewrites list entries 0..3 whatever the team sizes): 20 000 ft, 800 ft/s."""
    def put(i, lat, lon, psi):
        ic = cfg.init[i]
        ic.lat_geod_deg, ic.lon_deg, ic.psi_deg, ic.h_sl_ft, ic.u_fps = lat, lon, float(psi), 20000.0, 800.0
    if cfg.n_agents == 2:
        put(0, *curriculum_spawn(60.1, 120.0, 11.119, angle))
        put(1, 60.1, 120.0, 0)
    else:
        put(0, *curriculum_spawn(60.1, 120.0, 11.119, angle))
        put(1, *curriculum_spawn(60.1, 120.01, 11.119, angle))
        put(2, 60.1, 120.0, 0)
        put(3, 60.1, 120.01, 0)
    return cfg

# defaults of AircraftSimulator.clear_defalut_condition (simulatior.py:192-208)
_IC_DEFAULT = dict(lon_deg=120.0, lat_geod_deg=60.0, h_sl_ft=20000.0, psi_deg=0.0, u_fps=800.0, v_fps=0.0, w_fps=0.0,
                   p_rad_sec=0.0, q_rad_sec=0.0, r_rad_sec=0.0)
_IC_KEYS = dict(ic_long_gc_deg="lon_deg", ic_lat_geod_deg="lat_geod_deg", ic_h_sl_ft="h_sl_ft", ic_psi_true_deg="psi_deg",
                ic_u_fps="u_fps", ic_v_fps="v_fps", ic_w_fps="w_fps", ic_p_rad_sec="p_rad_sec", ic_q_rad_sec="q_rad_sec",
                ic_r_rad_sec="r_rad_sec")


def _clip(v, lo, hi):
    return min(max(v, lo), hi)


def config_from_dict(data, task=None, hierarchical=None):
    """Build an AcConfig from a parsed scenario dict; ``task`` overrides the YAML's task name with one of TASK_IDS;
    ``hierarchical`` (None = what the task name implies) selects the [3,5,3] action through the low-level controller."""
    cfg = AcConfig()
    name = task or data.get("task")
    curriculum = name in CURRICULUM_BASE
    if curriculum:
        if hierarchical is None:
            hierarchical = False
        name = CURRICULUM_BASE[name]
    rwr = name in RWR_BASE
    if rwr:
        name = RWR_BASE[name]
    legacy = name in LEGACY_OBS_BASE
    if legacy:
        name = LEGACY_OBS_BASE[name]
    if name not in TASK_IDS:
        raise NotImplementedError(f"Unknown taskname: {name} (available: {sorted(TASK_IDS)})")
    cfg.task = TASK_IDS[name]
    cfg.rwr = int(rwr)
    cfg.legacy_obs = int(legacy or name in ("hierarchical_multiplecombat_shoot", "multiplecombat_shoot", "multiplecombat_dodge_missile",
                                            "hierarchical_multiplecombat_dodge_missile"))
    cfg.approach = int(name == "approach")
    cfg.hierarchical = int(name in ALWAYS_HIERARCHICAL if hierarchical is None else bool(hierarchical))
    acs = data["aircraft_configs"]
    uids = list(acs.keys())
    if len(uids) > AC_MAX_AGENTS:
        raise ValueError("too many aircraft")
    team0 = uids[0][0]
    ego = [u for u in uids if u[0] == team0]
    enm = [u for u in uids if u[0] != team0]
    if uids != ego + enm:
        raise ValueError("aircraft_configs must list the ego team first (BaseEnv._pack order)")
    cfg.n_agents = len(uids)
    cfg.n_ego = len(ego)
    cfg.sim_freq = int(data.get("sim_freq", 60))
    cfg.agent_interaction_steps = int(data.get("agent_interaction_steps", 12))   # env_base.py:27
    cfg.max_steps = int(data.get("max_steps", 100))                              # env_base.py:25
    lon, lat, alt = data.get("battle_field_center", (120.0, 60.0, 0.0))
    cfg.center_lon, cfg.center_lat, cfg.center_alt = float(lon), float(lat), float(alt)
    cfg.altitude_limit = float(data.get("altitude_limit", 2500))
    cfg.acc_limit_x = float(data.get("acceleration_limit_x", 10.0))
    cfg.acc_limit_y = float(data.get("acceleration_limit_y", 10.0))
    cfg.acc_limit_z = float(data.get("acceleration_limit_z", 10.0))
    for i, uid in enumerate(uids):
        ic = dict(_IC_DEFAULT)
        for k, v in (acs[uid].get("init_state") or {}).items():
            if k in _IC_KEYS:
                ic[_IC_KEYS[k]] = float(v)
        # catalogue bounds applied by set_property_value (catalog.py:237,247)
        ic["h_sl_ft"] = _clip(ic["h_sl_ft"], -1400, 85000)
        ic["psi_deg"] = _clip(ic["psi_deg"], 0, 360)
        for k, v in ic.items():
            setattr(cfg.init[i], k, v)
        cfg.num_missiles[i] = int(acs[uid].get("missile", 0))
    g = data.get
    cfg.posture_scale = float(g("PostureReward_scale", 1.0)); cfg.posture_potential = int(bool(g("PostureReward_potential", False)))
    cfg.altitude_scale = float(g("AltitudeReward_scale", 1.0)); cfg.altitude_potential = int(bool(g("AltitudeReward_potential", False)))
    cfg.event_scale = float(g("EventDrivenReward_scale", 1.0)); cfg.event_potential = int(bool(g("EventDrivenReward_potential", False)))
    cfg.missile_posture_scale = float(g("MissilePostureReward_scale", 1.0))
    cfg.shoot_penalty_scale = float(g("ShootPenaltyReward_scale", 1.0)); cfg.shoot_penalty_potential = int(bool(g("ShootPenaltyReward_potential", False)))
    for ver, want in (("PostureReward_orientation_version", "v2"), ("PostureReward_range_version", "v3")):
        if g(ver, want) != want:
            raise NotImplementedError(f"{ver}={g(ver)}: only {want} (the one every shipped YAML selects) is implemented")
    cfg.alt_safe = float(g("AltitudeReward_safe_altitude", 4.0))
    cfg.alt_danger = float(g("AltitudeReward_danger_altitude", 3.5))
    cfg.alt_kv = float(g("AltitudeReward_Kv", 0.2))
    cfg.max_attack_angle = float(g("max_attack_angle", 180))
    cfg.max_attack_distance = float(g("max_attack_distance", float("inf")))
    cfg.min_attack_interval = int(g("min_attack_interval", 125))
    cfg.use_artillery = int(bool(g("use_artillery", False)))
    # scripted opponents (singlecombat_task.py:19-27, load_agent :197-207): 'pursue' and 'maneuver' feed the low-level controller;
    # 'loiter' is named by some shipped YAMLs but the reference has no such agent (load_agent raises NotImplementedError)
    cfg.use_baseline = 0
    if g("use_baseline", False):
        kind = g("baseline_type", "pursue")
        if kind not in ("pursue", "maneuver"):
            raise NotImplementedError(f"baseline_type={kind}: the reference's load_agent only knows the controller-driven 'pursue' and 'maneuver' here")
        cfg.use_baseline = 1 if kind == "pursue" else 2
    # HeadingTask: HeadingReward scale / potential; UnreachHeading reads its limits from the first aircraft's block
    # (unreach_heading.py:14-19)
    cfg.heading_scale = float(g("HeadingReward_scale", 1.0)); cfg.heading_potential = int(bool(g("HeadingReward_potential", False)))
    first = acs[uids[0]]
    cfg.max_heading_increment = float(first.get("max_heading_increment", 180))
    cfg.max_altitude_increment = float(first.get("max_altitude_increment", 7000))
    cfg.max_velocities_u_increment = float(first.get("max_velocities_u_increment", 100))
    cfg.check_interval = float(first.get("check_interval", 30))
    if curriculum:
        apply_curriculum_spawn(cfg, int(data.get("curriculum_angle", 0)))
    return cfg


def config_from_yaml(path, task=None, hierarchical=None):
    with open(path, "r", encoding="utf-8") as f:
        data = yaml.load(f, Loader=yaml.FullLoader)
    if hierarchical is None:   # a shipped YAML means the reference's own task class
        hierarchical = (task or data.get("task")) in HIERARCHICAL_IN_REFERENCE
    return config_from_dict(data, task=task, hierarchical=hierarchical)


def default_nvn_config(n_per_side=2, task="multiplecombat", hierarchical=False):
    """The aircraft block of reference configs/scenario2/scenario2_nvn.yaml (2v2) or scenario3/scenario3_nvn.yaml (4v4)
    with the MultipleCombatTask semantics (BASELINE configs C4 / C5 without the weapon rules)."""
    acs = {}
    for team, lat, psi, color in (("A", 60.0, 0.0, "Blue"), ("B", 60.1, 180.0, "Red")):
        for k in range(n_per_side):
            acs[f"{team}0{k + 1}00"] = {"color": color, "model": "f16", "missile": 2,
                                      "init_state": {"ic_long_gc_deg": 120.0 + 0.01 * k, "ic_lat_geod_deg": lat, "ic_h_sl_ft": 20000,
                                                     "ic_psi_true_deg": psi, "ic_u_fps": 800.0}}
    data = {"task": task, "sim_freq": 60, "agent_interaction_steps": 6, "max_steps": 9000, "altitude_limit": 2500,
            "acceleration_limit_x": 10.0, "acceleration_limit_y": 10.0, "acceleration_limit_z": 10.0, "aircraft_configs": acs,
            "max_attack_angle": 45, "max_attack_distance": 14000, "min_attack_interval": 125, "battle_field_center": [120.0, 60.0, 0.0],
            "PostureReward_scale": 15.0, "PostureReward_potential": True, "PostureReward_orientation_version": "v2",
            "PostureReward_range_version": "v3", "AltitudeReward_safe_altitude": 4.0, "AltitudeReward_danger_altitude": 3.5,
            "AltitudeReward_Kv": 0.2}
    return config_from_dict(data, hierarchical=hierarchical)


def default_heading_config():
    """reference configs/singlecontrol/heading.yaml (BASELINE config C1)."""
    data = {"task": "heading", "sim_freq": 60, "agent_interaction_steps": 6, "max_steps": 10000, "altitude_limit": 2500,
            "acceleration_limit_x": 10.0, "acceleration_limit_y": 10.0, "acceleration_limit_z": 10.0,
            "aircraft_configs": {"A0100": {"color": "Blue", "model": "f16", "max_heading_increment": 180, "max_altitude_increment": 7000,
                                           "max_velocities_u_increment": 100, "check_interval": 30,
                                           "init_state": {"ic_long_gc_deg": 120.0, "ic_lat_geod_deg": 60.0, "ic_h_sl_ft": 20000,
                                                          "ic_psi_true_deg": 0.0, "ic_u_fps": 800.0}}},
            "battle_field_center": [120.0, 60.0, 0.0],
            "AltitudeReward_safe_altitude": 4.0, "AltitudeReward_danger_altitude": 3.5, "AltitudeReward_Kv": 0.2}
    return config_from_dict(data, hierarchical=False)


def default_config(task="singlecombat", hierarchical=False):
    """The 1v1 block of reference configs/scenario1/WVR_selfplay.yaml (BASELINE configs C2 / C3)."""
    if task == "heading":
        return default_heading_config()
    if task == "approach":                           # configs/singlecontrol/approach.yaml: the heading aircraft block, task approach
        cfg = default_heading_config()
        cfg.approach = 1
        return cfg
    if task in ALWAYS_HIERARCHICAL:
        hierarchical = True
    if task in ("multiplecombat", "hierarchical_multiplecombat", "hierarchical_multiplecombat_shoot", "multiplecombat_shoot", "multiplecombat_dodge_missile",
                "hierarchical_multiplecombat_dodge_missile"):
        return default_nvn_config(2, task=task if task.endswith(("_shoot", "_dodge_missile")) else "multiplecombat", hierarchical=hierarchical)
    if task in ("scenario_nvn", "scenario2_nvn"):
        return default_nvn_config(2, task="scenario_nvn", hierarchical=hierarchical)
    if task == "scenario3_nvn":
        return default_nvn_config(4, task="scenario_nvn", hierarchical=hierarchical)
    data = {
        "task": task, "sim_freq": 60, "agent_interaction_steps": 6, "max_steps": 9000, "altitude_limit": 2500,
        "acceleration_limit_x": 10.0, "acceleration_limit_y": 10.0, "acceleration_limit_z": 10.0,
        "aircraft_configs": {
            "A0100": {"color": "Blue", "model": "f16", "missile": 2,
                      "init_state": {"ic_long_gc_deg": 120.0, "ic_lat_geod_deg": 60.0, "ic_h_sl_ft": 20000,
                                     "ic_psi_true_deg": 0, "ic_u_fps": 800.0}},
            "B0100": {"color": "Red", "model": "f16", "missile": 2,
                      "init_state": {"ic_h_sl_ft": 20000, "ic_lat_geod_deg": 60.1, "ic_long_gc_deg": 120.5,
                                     "ic_psi_true_deg": 180.0, "ic_u_fps": 800.0}},
        },
        "max_attack_angle": 45, "max_attack_distance": 14000, "min_attack_interval": 25,
        "battle_field_center": [120.0, 60.0, 0.0],
        "MissilePostureReward_scale": 30, "PostureReward_scale": 15.0, "PostureReward_potential": True,
        "PostureReward_orientation_version": "v2", "PostureReward_range_version": "v3",
        "AltitudeReward_safe_altitude": 4.0, "AltitudeReward_danger_altitude": 3.5, "AltitudeReward_Kv": 0.2,
        "EventDrivenReward_scale": 1, "EventDrivenReward_potential": True,
    }
    return config_from_dict(data, hierarchical=hierarchical)
