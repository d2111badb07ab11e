/* A plain-C host of libcosmo_pol_hip.so: includes the public header as C, links the
 * library, exercises the entry points that need no GPU (per-ray tables) and checks that
 * context creation reports a clean error code when no device is usable.
 * Built and run by tests/test_cabi_cpu.py::test_c_host_links_and_runs. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cosmo_pol_amd.h"

int main(void)
{
    cpol_sweep_params p;
    memset(&p, 0, sizeof p);
    p.n_rays = 2; p.n_hnodes = 1; p.n_vnodes = 1;
    p.sin_u1 = sin(0.8); p.cos_u1 = cos(0.8);
    const double az[2] = {0.0, 90.0}, el[2] = {1.0, 45.0}, off[1] = {0.0};
    /* sized exactly as the header documents: [n_rays][n_vnodes][CPOL_TRAJ_STRIDE] and
       [n_rays][n_hnodes][CPOL_GEO_STRIDE], with one guard element behind each */
    enum { NT = 2 * 1 * CPOL_TRAJ_STRIDE, NG = 2 * 1 * CPOL_GEO_STRIDE };
    double traj[NT + 1], geo[NG + 1];
    traj[NT] = -777.0; geo[NG] = -777.0;
    if (CPOL_TRAJ_STRIDE != 4 || CPOL_GEO_STRIDE != 8 || CPOL_SITE_STRIDE != 8) { printf("stride constants changed\n"); return 5; }
    if (cpol_ray_tables(&p, az, el, off, off, traj, geo) != CPOL_OK) { printf("ray_tables failed\n"); return 1; }
    if (traj[NT] != -777.0 || geo[NG] != -777.0) { printf("ray_tables wrote past the documented size\n"); return 6; }
    /* traj = (el_rad, sin, cos, el_deg) ; geo[0..1] = sin / cos of the azimuth */
    if (fabs(traj[3] - 1.0) > 1e-15 || fabs(traj[CPOL_TRAJ_STRIDE + 1] - sin(45.0 * 3.14159265358979323846 / 180.0)) > 1e-15 ||
        fabs(traj[CPOL_TRAJ_STRIDE + 3] - 45.0) > 1e-13 ||
        fabs(geo[0]) > 1e-15 || fabs(geo[CPOL_GEO_STRIDE + 0] - 1.0) > 1e-15) { printf("ray_tables values wrong\n"); return 2; }
    cpol_ctx *ctx = NULL;
    int rc = cpol_create(0, &ctx);
    if (rc == CPOL_OK) {            /* a GPU is present: the handle must work and go away */
        if (!ctx || cpol_synchronize(ctx) != CPOL_OK) { printf("context unusable\n"); return 3; }
        cpol_destroy(ctx);
        printf("C_HOST_OK gpu\n");
    } else {
        if (ctx != NULL || rc >= 0) { printf("bad error convention rc=%d\n", rc); return 4; }
        printf("C_HOST_OK nogpu rc=%d\n", rc);
    }
    return 0;
}
