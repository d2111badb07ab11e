// cpol_device.h -- device-side data structures and small helpers shared by the
// HIP kernels of libcosmo_pol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cosmo_pol_amd.h"

#define CPOL_WAVE 64

// WGS84 (Vincenty direct; mirrors oracle/cosmo_pol_oracle/geodesy.py)
#define CPOL_WGS84_F (1.0 / 298.257223563)
#define CPOL_VINCENTY_ITERS 5
#define CPOL_DEG (3.14159265358979323846 / 180.0)

// Model cube resident in HBM, re-laid-out at staging time so that ONE grid
// column is contiguous:  H[ny][nx][nz]  and  V[ny][nx][nz][n_vars]
// (the reference layout [nz][ny][nx] strides a column over planes of
//  ny*nx*4 B = 3.6 MB; see DESIGN.md "data layout in HBM").
struct ModelDev {
    const float *H;        // z-levels, column-major per cell, descending with k
    const float *V;        // variables interleaved per (cell, level)
    int n_vars, nz, ny, nx;
    float llc0, llc1;      // Lo1 (lon), La1 (lat)
    float urc0, urc1;
    float res0, res1;      // dlon, dlat
    // rotated-pole rotation: products of sin/cos evaluated on the host
    double ctcp, ctsp, st, nsp, cp, nstcp, stsp, ct;
};

// Per-hydrometeor staged data
struct HydroDev {
    cpol_hydro_desc d;
    const double *table;   // [n_e][n_t][n_d][12]
    const double *pre;     // [n_d]
    const double *dnu;     // [n_d]
    const double *aux;     // family specific
    const double *rcsw;    // [n_e][n_t][n_d][2] Doppler scheme 2 weights or NULL
    int key_base;          // first bucket id of this hydrometeor
    int n_par;             // per-item parameter count
};

struct HydroSet {
    int n_hydro;
    int n_keys;
    HydroDev h[CPOL_MAX_HYDRO];
};

// Doppler scheme 3 (spectrum): float32 radar cross sections per table bin and the float32
// diameter grid of get_doppler_spectrum with its float32 powers.  Kept OUT of HydroDev:
// growing that struct (a by-value kernel argument of the PSD kernels) by 24 bytes per
// slot made the dominant PSD kernel 25 % slower in isolation.
struct SpecDev {
    const float *rcs32;    // [n_e][n_t][n_d] or NULL
    const float *dgrid;    // [3][n_d]: D, D^mu, D^nu (float32) or NULL
    float step32;          // D[1] - D[0] in float32
    int pad_spec;
};

struct SpecSet {
    SpecDev s[CPOL_MAX_HYDRO];
};

struct WorkUnit {          // one wave of the PSD kernel
    int key;               // bucket id (hydrometeor, e bin, t bin)
    int start;             // first position in perm[]
    int count;             // 1..64 items
    int pad;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (CPOL_WAVE - 1); }

__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}
