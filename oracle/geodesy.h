/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * WGS84 geodetic <-> local NED, restating the two pymap3d entry points the reference calls
 * (envs/JSBSim/utils/utils.py:30-55: pymap3d.geodetic2ned / pymap3d.ned2geodetic; pymap3d is an
 * un-pinned third-party dependency absent from /root/reference). "parity unpinned" at this boundary:
 * the reference's tests hold no geodesy vectors; tests/ guard it with round trips and closed forms. */
#ifndef ORACLE_GEODESY_H
#define ORACLE_GEODESY_H
#ifdef __cplusplus
extern "C" {
#endif
void or_geodetic2ecef(double lat_deg, double lon_deg, double alt, double* x, double* y, double* z);
void or_ecef2geodetic(double x, double y, double z, double* lat_deg, double* lon_deg, double* alt);
/* LLA2NEU(lon, lat, alt, lon0, lat0, alt0) -> (n, e, u), utils.py:30-41 */
void or_lla2neu(double lon, double lat, double alt, double lon0, double lat0, double alt0, double neu[3]);
/* NEU2LLA(n, e, u, lon0, lat0, alt0) -> (lon, lat, alt), utils.py:44-55 */
void or_neu2lla(double n, double e, double u, double lon0, double lat0, double alt0, double lla[3]);
#ifdef __cplusplus
}
#endif
#endif
