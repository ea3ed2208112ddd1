/* ORACLE — TEST INFRASTRUCTURE ONLY (see geodesy.h). */
#include "geodesy.h"
#include <math.h>

#define WGS84_A 6378137.0
#define WGS84_F (1.0 / 298.257223563)
#define WGS84_B (WGS84_A * (1.0 - WGS84_F))
#define D2R (M_PI / 180.0)
#define R2D (180.0 / M_PI)

/* pymap3d.ecef.geodetic2ecef: prime-vertical radius N = a^2 / hypot(a cos, b sin) */
void or_geodetic2ecef(double lat_deg, double lon_deg, double alt, double* x, double* y, double* z) {
  double lat = lat_deg * D2R, lon = lon_deg * D2R;
  double N = WGS84_A * WGS84_A / hypot(WGS84_A * cos(lat), WGS84_B * sin(lat));
  *x = (N + alt) * cos(lat) * cos(lon);
  *y = (N + alt) * cos(lat) * sin(lon);
  *z = (N * (WGS84_B / WGS84_A) * (WGS84_B / WGS84_A) + alt) * sin(lat);
}

/* pymap3d.ecef.ecef2geodetic: You (2000) closed form with one correction step */
void or_ecef2geodetic(double x, double y, double z, double* lat_deg, double* lon_deg, double* alt) {
  const double a = WGS84_A, b = WGS84_B;
  double r = sqrt(x * x + y * y + z * z);
  double E = sqrt(a * a - b * b);
  double u = sqrt(0.5 * (r * r - E * E) + 0.5 * hypot(r * r - E * E, 2 * E * z));
  double hxy = hypot(x, y);
  double huE = hypot(u, E);
  double Beta = atan(huE / u * z / hxy);
  double dBeta = ((b * u - a * huE + E * E) * sin(Beta)) / (a * huE * 1 / cos(Beta) - E * E * cos(Beta));
  Beta += dBeta;
  double lat = atan(a / b * tan(Beta));
  double lon = atan2(y, x);
  double cosBeta = cos(Beta);
  double h = hypot(z - b * sin(Beta), hxy - a * cosBeta);
  int inside = (x * x / (a * a) + y * y / (a * a) + z * z / (b * b)) < 1.0;
  if (inside) h = -h;
  *lat_deg = lat * R2D;
  *lon_deg = lon * R2D;
  *alt = h;
}

void or_lla2neu(double lon, double lat, double alt, double lon0, double lat0, double alt0, double neu[3]) {
  double x, y, z, x0, y0, z0;
  or_geodetic2ecef(lat, lon, alt, &x, &y, &z);
  or_geodetic2ecef(lat0, lon0, alt0, &x0, &y0, &z0);
  double du = x - x0, dv = y - y0, dw = z - z0;
  double la = lat0 * D2R, lo = lon0 * D2R;
  double t = cos(lo) * du + sin(lo) * dv;
  double east = -sin(lo) * du + cos(lo) * dv;
  double up = cos(la) * t + sin(la) * dw;
  double north = -sin(la) * t + cos(la) * dw;
  neu[0] = north; neu[1] = east; neu[2] = up; /* utils.py:41 returns [n, e, -d] */
}

void or_neu2lla(double n, double e, double u, double lon0, double lat0, double alt0, double lla[3]) {
  double x0, y0, z0;
  or_geodetic2ecef(lat0, lon0, alt0, &x0, &y0, &z0);
  double la = lat0 * D2R, lo = lon0 * D2R;
  /* pymap3d.enu2uvw */
  double t = cos(la) * u - sin(la) * n;
  double dw = sin(la) * u + cos(la) * n;
  double du = cos(lo) * t - sin(lo) * e;
  double dv = sin(lo) * t + cos(lo) * e;
  double lat, lon, alt;
  or_ecef2geodetic(x0 + du, y0 + dv, z0 + dw, &lat, &lon, &alt);
  lla[0] = lon; lla[1] = lat; lla[2] = alt;
}
