"""Tacview ACMI text records of one env: the format written by BaseEnv.render (envs/JSBSim/envs/env_base.py:207-250)
from BaseSimulator.log / MissileSimulator.log (core/simulatior.py:73-79, 535-551). Host-side only; not on the step() path."""
import math

HEADER = "FileType=text/acmi/tacview\nFileVersion=2.1\n0,ReferenceTime=2020-04-01T00:00:00Z\n"
_A, _F = 6378137.0, 1 / 298.257223563
_B = _A * (1 - _F)


def neu_to_lla(n, e, u, lon0, lat0, alt0):
    """NEU2LLA (utils/utils.py:44-55 -> pymap3d.ned2geodetic): ENU offset about the battle-field centre -> lon, lat (deg), height (m).
    Geodetic -> ECEF, rotate the offset into ECEF, then the closed-form inverse (Bowring start + two Newton steps: < 1e-9 deg)."""
    la0, lo0 = math.radians(lat0), math.radians(lon0)
    N0 = _A ** 2 / math.hypot(_A * math.cos(la0), _B * math.sin(la0))
    x0 = (N0 + alt0) * math.cos(la0) * math.cos(lo0)
    y0 = (N0 + alt0) * math.cos(la0) * math.sin(lo0)
    z0 = (N0 * (_B / _A) ** 2 + alt0) * math.sin(la0)
    t = math.cos(la0) * u - math.sin(la0) * n
    x = x0 + math.cos(lo0) * t - math.sin(lo0) * e
    y = y0 + math.sin(lo0) * t + math.cos(lo0) * e
    z = z0 + math.sin(la0) * u + math.cos(la0) * n
    e2 = 1 - (_B / _A) ** 2
    p = math.hypot(x, y)
    lat = math.atan2(z, p * (1 - e2))
    for _ in range(4):
        Nn = _A / math.sqrt(1 - e2 * math.sin(lat) ** 2)
        h = p / math.cos(lat) - Nn
        lat = math.atan2(z, p * (1 - e2 * Nn / (Nn + h)))
    Nn = _A / math.sqrt(1 - e2 * math.sin(lat) ** 2)
    return math.degrees(math.atan2(y, x)), math.degrees(lat), p / math.cos(lat) - Nn


MISSILE_MODELS = {0: "AIM-9L", 1: "AIM-120B", 2: "AIM-9M"}   # ac_get_missile's model code -> MissileSimulator.model


def aircraft_record(uid, color, entity, model="f16"):
    """BaseSimulator.log (simulatior.py:73-79): `entity` = ac_get_entity's lon, lat (deg), alt (m), roll, pitch, yaw (rad), ..."""
    lon, lat, alt, roll, pitch, yaw = entity[:6]
    deg = lambda x: x * 180 / math.pi           # get_rpy() * 180 / np.pi, in the reference's operation order
    return (f"{uid},T={lon}|{lat}|{alt}|{deg(roll)}|{deg(pitch)}|{deg(yaw)},"
            f"Name={model.upper()},Color={color}")


def missile_records(uid, color, status, neu, theta, psi, center, exploded, radius, model="AIM-9L"):
    """MissileSimulator.log (simulatior.py:535-551): alive -> a position record; first frame after it is done -> removal + explosion;
    later -> removal (the reference's removal message carries its own newline, so the file shows an empty line after it).
    Returns (text, exploded_flag)."""
    lon, lat, alt = neu_to_lla(neu[0], neu[1], neu[2], *center)
    pose = f"T={lon}|{lat}|{alt}|0.0|{theta * 180 / math.pi}|{psi * 180 / math.pi}"
    if status == 0:
        return f"{uid},{pose},Name={model.upper()},Color={color}", exploded
    if not exploded:
        return f"-{uid}\n{uid}F,{pose},Type=Misc+Explosion,Color={color},Radius={radius}", True
    return f"-{uid}\n", exploded


def chaff_record(uid, color, alive, pose, model="CHF"):
    """ChaffSimulator.log (simulatior.py:383-388): the cloud keeps the geodetic position and attitude its parent had at the release
    (`pose` = lon, lat, alt, roll, pitch, yaw) while it is effective; afterwards the removal message."""
    if alive:
        return aircraft_record(uid, color, pose, model)
    return f"-{uid}\n"
