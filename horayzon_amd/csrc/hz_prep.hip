// hz_prep.hip -- the steps immediately before / after the ray-casting path,
// as streaming HIP kernels (one lane per element; HBM bound):
//   slope_plane_meth / slope_vector_meth        topo_param.pyx:84-225, :284-372
//   lonlat2ecef, ecef2enu, ecef2enu_vector      transform.pyx:60-103, :152-189, :231-261
//   surf_norm, north_dir                        direction.pyx:48-70, :125-178
// Arithmetic types follow the reference statement by statement (float32 state for the slope
// routines, float64 for the coordinate transforms, float32 outputs where the reference has them).
// The reference builds these modules with -ffast-math (setup.py:24), so agreement is to rounding,
// not bit for bit; tests/golden holds vectors produced by the reference modules themselves.
#include "hz_internal.h"
#include <cmath>

namespace hz {

__device__ __forceinline__ double deg2rad_d(double a) { return a * (3.14159265358979323846 / 180.0); }

// --------------------------------------------------------------------------------------------
// slope: plane fit to the 3 x 3 neighbourhood (least squares in z), normal of the plane
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slope_plane(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ z, int len_0, int len_1,
                                                    const float *__restrict__ rot_mat, int output_rot,
                                                    float *__restrict__ vec_tilt) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (size_t)len_0 * len_1) return;
    const int i = (int)(c / len_1), j = (int)(c - (size_t)i * len_1);
    const float nanv = __int_as_float(0x7fc00000);
    if (i < 1 || j < 1 || i >= len_0 - 1 || j >= len_1 - 1) {       // vec_tilt[:] = NAN, :129
        vec_tilt[3 * c] = nanv; vec_tilt[3 * c + 1] = nanv; vec_tilt[3 * c + 2] = nanv;
        return;
    }
    float r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (rot_mat)
        for (int k = 0; k < 9; k++) r[k] = rot_mat[9 * c + k];
    const float xc = x[c], yc = y[c], zc = z[c];
    float x_l_sum = 0, y_l_sum = 0, z_l_sum = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0;
    for (int k = i - 1; k <= i + 1; k++)
        for (int l = j - 1; l <= j + 1; l++) {                    // translate, rotate: :136-154
            const size_t q = (size_t)k * len_1 + l;
            const float cx = x[q] - xc, cy = y[q] - yc, cz = z[q] - zc;
            const float vx = r[0] * cx + r[1] * cy + r[2] * cz;
            const float vy = r[3] * cx + r[4] * cy + r[5] * cz;
            const float vz = r[6] * cx + r[7] * cy + r[8] * cz;
            x_l_sum += vx; y_l_sum += vy; z_l_sum += vz;          // :165-173
            xx += vx * vx; xy += vx * vy; xz += vx * vz; yy += vy * vy; yz += vy * vz;
        }
    // 3 x 3 system (:175-186) solved as LAPACK sgesv does: LU with partial pivoting, float32
    float A[3][3] = {{xx, xy, x_l_sum}, {xy, yy, y_l_sum}, {x_l_sum, y_l_sum, 9.0f}};
    float b[3] = {xz, yz, z_l_sum};
    for (int col = 0; col < 3; col++) {
        int piv = col;
        float best = fabsf(A[col][col]);
        for (int row = col + 1; row < 3; row++)
            if (fabsf(A[row][col]) > best) { best = fabsf(A[row][col]); piv = row; }
        if (piv != col) {
            for (int k = 0; k < 3; k++) { const float t = A[col][k]; A[col][k] = A[piv][k]; A[piv][k] = t; }
            const float t = b[col]; b[col] = b[piv]; b[piv] = t;
        }
        const float inv = 1.0f / A[col][col];
        for (int row = col + 1; row < 3; row++) {
            const float f = A[row][col] * inv;
            for (int k = col + 1; k < 3; k++) A[row][k] -= f * A[col][k];
            b[row] -= f * b[col];
        }
    }
    float s1 = (b[1] - A[1][2] * (b[2] / A[2][2])) / A[1][1];
    float s0 = (b[0] - A[0][1] * s1 - A[0][2] * (b[2] / A[2][2])) / A[0][0];
    float vx = s0, vy = s1, vz = -1.0f;                           // vec[2] = -1.0, :189
    const float mag = sqrtf(vx * vx + vy * vy + vz * vz);         // :196-199
    vx /= mag; vy /= mag; vz /= mag;
    if (vz < 0.0f) { vx = -vx; vy = -vy; vz = -vz; }              // :202-205
    if (!output_rot) {                                            // rotate back with the transpose, :218-231
        const float ox = r[0] * vx + r[3] * vy + r[6] * vz;
        const float oy = r[1] * vx + r[4] * vy + r[7] * vz;
        const float oz = r[2] * vx + r[5] * vy + r[8] * vz;
        vx = ox; vy = oy; vz = oz;
    }
    vec_tilt[3 * c] = vx; vec_tilt[3 * c + 1] = vy; vec_tilt[3 * c + 2] = vz;
}

// slope: average of the normals of the 4 adjacent triangles (Corripio 2003), :308-343
__global__ __launch_bounds__(256) void k_slope_vector(const float *__restrict__ x, const float *__restrict__ y,
                                                     const float *__restrict__ z, int len_0, int len_1,
                                                     const float *__restrict__ rot_mat, int output_rot,
                                                     float *__restrict__ vec_tilt) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (size_t)len_0 * len_1) return;
    const int i = (int)(c / len_1), j = (int)(c - (size_t)i * len_1);
    const float nanv = __int_as_float(0x7fc00000);
    if (i < 1 || j < 1 || i >= len_0 - 1 || j >= len_1 - 1) {
        vec_tilt[3 * c] = nanv; vec_tilt[3 * c + 1] = nanv; vec_tilt[3 * c + 2] = nanv;
        return;
    }
    const size_t w = c - 1, e = c + 1, s = c + len_1, n = c - len_1;
    const float a_x = x[w] - x[c], a_y = y[w] - y[c], a_z = z[w] - z[c];
    const float b_x = x[s] - x[c], b_y = y[s] - y[c], b_z = z[s] - z[c];
    const float c_x = x[e] - x[c], c_y = y[e] - y[c], c_z = z[e] - z[c];
    const float d_x = x[n] - x[c], d_y = y[n] - y[c], d_z = z[n] - z[c];
    float vx = ((a_y * b_z - a_z * b_y) + (b_y * c_z - b_z * c_y) + (c_y * d_z - c_z * d_y) + (d_y * a_z - d_z * a_y)) / 4.0f;
    float vy = ((a_z * b_x - a_x * b_z) + (b_z * c_x - b_x * c_z) + (c_z * d_x - c_x * d_z) + (d_z * a_x - d_x * a_z)) / 4.0f;
    float vz = ((a_x * b_y - a_y * b_x) + (b_x * c_y - b_y * c_x) + (c_x * d_y - c_y * d_x) + (d_x * a_y - d_y * a_x)) / 4.0f;
    const float mag = sqrtf(vx * vx + vy * vy + vz * vz);
    vx /= mag; vy /= mag; vz /= mag;
    if (vz < 0.0f) { vx = -vx; vy = -vy; vz = -vz; }
    if (output_rot && rot_mat) {                                  // :355-370
        const float *r = rot_mat + 9 * c;
        const float ox = r[0] * vx + r[1] * vy + r[2] * vz;
        const float oy = r[3] * vx + r[4] * vy + r[5] * vz;
        const float oz = r[6] * vx + r[7] * vy + r[8] * vz;
        vx = ox; vy = oy; vz = oz;
    }
    vec_tilt[3 * c] = vx; vec_tilt[3 * c + 1] = vy; vec_tilt[3 * c + 2] = vz;
}

// --------------------------------------------------------------------------------------------
// coordinate transforms (float64)
// --------------------------------------------------------------------------------------------
struct Ellps { int sphere; double r, a, b, e_2; };

static Ellps make_ellps(int kind) {   // 0 sphere, 1 GRS80, 2 WGS84 (transform.pyx:73-87)
    Ellps e;
    e.sphere = (kind == 0);
    e.r = 6370997.0;
    e.a = 6378137.0;
    const double f = (kind == 1) ? (1.0 / 298.257222101) : (1.0 / 298.257223563);
    e.b = e.a * (1.0 - f);
    e.e_2 = 1.0 - ((e.b * e.b) / (e.a * e.a));
    return e;
}

__device__ __forceinline__ void lonlat2ecef_one(const Ellps &e, double lon, double lat, float h, double &X,
                                                double &Y, double &Z) {
    const double sl = sin(deg2rad_d(lat)), cl = cos(deg2rad_d(lat));
    const double so = sin(deg2rad_d(lon)), co = cos(deg2rad_d(lon));
    if (e.sphere) {                                               // transform.pyx:74-82
        X = (e.r + (double)h) * cl * co;
        Y = (e.r + (double)h) * cl * so;
        Z = (e.r + (double)h) * sl;
    } else {                                                      // :93-101
        const double n = e.a / sqrt(1.0 - e.e_2 * (sl * sl));
        X = (n + (double)h) * cl * co;
        Y = (n + (double)h) * cl * so;
        Z = ((e.b * e.b) / (e.a * e.a) * n + (double)h) * sl;
    }
}

__global__ __launch_bounds__(256) void k_lonlat2ecef(Ellps e, const double *__restrict__ lon,
                                                    const double *__restrict__ lat, const float *__restrict__ h,
                                                    size_t n, double *__restrict__ X, double *__restrict__ Y,
                                                    double *__restrict__ Z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lonlat2ecef_one(e, lon[i], lat[i], h[i], X[i], Y[i], Z[i]);
}

struct EnuFrame { double sin_lon, cos_lon, sin_lat, cos_lat, x_or, y_or, z_or; };

__global__ __launch_bounds__(256) void k_ecef2enu(EnuFrame f, const double *__restrict__ X, const double *__restrict__ Y,
                                                 const double *__restrict__ Z, size_t n, float *__restrict__ xe,
                                                 float *__restrict__ ye, float *__restrict__ ze) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dx = X[i] - f.x_or, dy = Y[i] - f.y_or, dz = Z[i] - f.z_or;     // transform.pyx:178-187
    xe[i] = (float)(-f.sin_lon * dx + f.cos_lon * dy);
    ye[i] = (float)(-f.sin_lat * f.cos_lon * dx - f.sin_lat * f.sin_lon * dy + f.cos_lat * dz);
    ze[i] = (float)(+f.cos_lat * f.cos_lon * dx + f.cos_lat * f.sin_lon * dy + f.sin_lat * dz);
}

__global__ __launch_bounds__(256) void k_ecef2enu_vector(EnuFrame f, const float *__restrict__ v, size_t n,
                                                        float *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = v[3 * i], b = v[3 * i + 1], c = v[3 * i + 2];                 // transform.pyx:252-259
    o[3 * i] = (float)(-f.sin_lon * a + f.cos_lon * b);
    o[3 * i + 1] = (float)(-f.sin_lat * f.cos_lon * a - f.sin_lat * f.sin_lon * b + f.cos_lat * c);
    o[3 * i + 2] = (float)(+f.cos_lat * f.cos_lon * a + f.cos_lat * f.sin_lon * b + f.sin_lat * c);
}

__global__ __launch_bounds__(256) void k_surf_norm(const double *__restrict__ lon, const double *__restrict__ lat,
                                                  size_t n, float *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double so = sin(deg2rad_d(lon[i])), co = cos(deg2rad_d(lon[i]));        // direction.pyx:61-68
    const double sl = sin(deg2rad_d(lat[i])), cl = cos(deg2rad_d(lat[i]));
    o[3 * i] = (float)(cl * co); o[3 * i + 1] = (float)(cl * so); o[3 * i + 2] = (float)sl;
}

__global__ __launch_bounds__(256) void k_north_dir(double np_z, const double *__restrict__ X,
                                                  const double *__restrict__ Y, const double *__restrict__ Z,
                                                  const float *__restrict__ vn, size_t n, float *__restrict__ o) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double vx = 0.0 - X[i], vy = 0.0 - Y[i], vz = np_z - Z[i];              // direction.pyx:157-176
    const double nx = vn[3 * i], ny = vn[3 * i + 1], nz = vn[3 * i + 2];
    const double dot = (vx * nx) + (vy * ny) + (vz * nz);
    const double px = vx - dot * nx, py = vy - dot * ny, pz = vz - dot * nz;
    const double norm = sqrt(px * px + py * py + pz * pz);
    o[3 * i] = (float)(px / norm); o[3 * i + 1] = (float)(py / norm); o[3 * i + 2] = (float)(pz / norm);
}

static EnuFrame make_frame(const Ellps &e, double lon_or, double lat_or) {       // transform.pyx:455-487
    EnuFrame f;
    const double lo = lon_or * (M_PI / 180.0), la = lat_or * (M_PI / 180.0);
    f.sin_lon = sin(lo); f.cos_lon = cos(lo); f.sin_lat = sin(la); f.cos_lat = cos(la);
    if (e.sphere) {
        f.x_or = e.r * cos(la) * cos(lo); f.y_or = e.r * cos(la) * sin(lo); f.z_or = e.r * sin(la);
    } else {
        const double n = e.a / sqrt(1.0 - e.e_2 * (sin(la) * sin(la)));
        f.x_or = n * cos(la) * cos(lo); f.y_or = n * cos(la) * sin(lo);
        f.z_or = ((e.b * e.b) / (e.a * e.a) * n) * sin(la);
    }
    return f;
}

static inline unsigned grid_of(size_t n) { return (unsigned)((n + 255) / 256); }

int prep_slope(int which, const float *x, const float *y, const float *z, int len_0, int len_1,
               const float *rot_mat, int output_rot, float *vec_tilt, hipStream_t st) {
    const size_t n = (size_t)len_0 * len_1;
    if (n == 0) return HZ_OK;
    if (which == 0)
        hipLaunchKernelGGL(k_slope_plane, dim3(grid_of(n)), dim3(256), 0, st, x, y, z, len_0, len_1, rot_mat, output_rot, vec_tilt);
    else
        hipLaunchKernelGGL(k_slope_vector, dim3(grid_of(n)), dim3(256), 0, st, x, y, z, len_0, len_1, rot_mat, output_rot, vec_tilt);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_lonlat2ecef(int ellps, const double *lon, const double *lat, const float *h, size_t n, double *X,
                     double *Y, double *Z, hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_lonlat2ecef, dim3(grid_of(n)), dim3(256), 0, st, make_ellps(ellps), lon, lat, h, n, X, Y, Z);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_ecef2enu(int ellps, double lon_or, double lat_or, const double *X, const double *Y, const double *Z,
                  size_t n, float *xe, float *ye, float *ze, hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_ecef2enu, dim3(grid_of(n)), dim3(256), 0, st, make_frame(make_ellps(ellps), lon_or, lat_or),
                       X, Y, Z, n, xe, ye, ze);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_ecef2enu_vector(int ellps, double lon_or, double lat_or, const float *v, size_t n, float *o, hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_ecef2enu_vector, dim3(grid_of(n)), dim3(256), 0, st,
                       make_frame(make_ellps(ellps), lon_or, lat_or), v, n, o);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// Swiss projection coordinates (LV95) <-> WGS84, swisstopo's approximate formulas as the reference evaluates them
// (transform.pyx:306-345, :390-432): float64 arithmetic, the height in float32
__global__ __launch_bounds__(256) void k_wgs2swiss(const double *__restrict__ lon, const double *__restrict__ lat,
                                                  const float *__restrict__ h_wgs, size_t n, double *__restrict__ e,
                                                  double *__restrict__ no, float *__restrict__ h_ch) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lo = ((lon[i] * 3600.0) - 26782.5) / 10000.0;       // arc seconds relative to Bern, in 10000"
    const double la = ((lat[i] * 3600.0) - 169028.66) / 10000.0;
    e[i] = 2600072.37 + 211455.93 * lo - 10938.51 * lo * la - 0.36 * lo * (la * la) - 44.54 * (lo * lo * lo);
    no[i] = 1200147.07 + 308807.95 * la + 3745.25 * (lo * lo) + 76.63 * (la * la) - 194.56 * (lo * lo) * la
            + 119.79 * (la * la * la);
    h_ch[i] = (float)((double)h_wgs[i] - 49.55 + 2.73 * lo + 6.94 * la);
}

__global__ __launch_bounds__(256) void k_swiss2wgs(const double *__restrict__ e, const double *__restrict__ no,
                                                  const float *__restrict__ h_ch, size_t n, double *__restrict__ lon,
                                                  double *__restrict__ lat, float *__restrict__ h_wgs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ep = (e[i] - 2600000.0) / 1000000.0, np_ = (no[i] - 1200000.0) / 1000000.0;   // civilian system, 1000 km
    const double lo = 2.6779094 + 4.728982 * ep + 0.791484 * ep * np_ + 0.1306 * ep * (np_ * np_) - 0.0436 * (ep * ep * ep);
    const double la = 16.9023892 + 3.238272 * np_ - 0.270978 * (ep * ep) - 0.002528 * (np_ * np_) - 0.0447 * (ep * ep) * np_
                      - 0.0140 * (np_ * np_ * np_);
    h_wgs[i] = (float)((double)h_ch[i] + 49.55 - 12.60 * ep - 22.64 * np_);
    lon[i] = lo * (100.0 / 36.0);                                      // 10000" -> degree
    lat[i] = la * (100.0 / 36.0);
}

int prep_wgs2swiss(const double *lon, const double *lat, const float *h_wgs, size_t n, double *e, double *no, float *h_ch,
                   hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_wgs2swiss, dim3(grid_of(n)), dim3(256), 0, st, lon, lat, h_wgs, n, e, no, h_ch);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_swiss2wgs(const double *e, const double *no, const float *h_ch, size_t n, double *lon, double *lat, float *h_wgs,
                   hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_swiss2wgs, dim3(grid_of(n)), dim3(256), 0, st, e, no, h_ch, n, lon, lat, h_wgs);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_surf_norm(const double *lon, const double *lat, size_t n, float *o, hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_surf_norm, dim3(grid_of(n)), dim3(256), 0, st, lon, lat, n, o);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int prep_north_dir(int ellps, const double *X, const double *Y, const double *Z, const float *vn, size_t n,
                   float *o, hipStream_t st) {
    if (n == 0) return HZ_OK;
    const Ellps e = make_ellps(ellps);
    hipLaunchKernelGGL(k_north_dir, dim3(grid_of(n)), dim3(256), 0, st, e.sphere ? e.r : e.b, X, Y, Z, vn, n, o);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// x, y, z planes -> interleaved xyz vertex buffer + zero padding (the vert_grid layout of the boundary:
// reference auxiliary.py:49-95, rearrange_pad_buffer / pad_buffer).  One lane per output float4 where the
// buffer allows it would need a 3:4 shuffle; a lane per vertex with three 4 B stores is HBM-bound enough
// (24 B moved per vertex, once per DEM).
__global__ __launch_bounds__(256) void k_pack_vertices(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ z, size_t n, size_t n_total,
                                                      float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[3 * i] = x[i]; out[3 * i + 1] = y[i]; out[3 * i + 2] = z[i]; }
    const size_t pad = n_total - 3 * n;          // trailing zeros (>= 16 floats)
    if (i < pad) out[3 * n + i] = 0.0f;
}

int prep_pack_vertices(const float *x, const float *y, const float *z, size_t n, size_t n_total, float *out,
                       hipStream_t st) {
    if (n_total == 0) return HZ_OK;
    const size_t work = std::max(n, n_total - 3 * n);
    hipLaunchKernelGGL(k_pack_vertices, dim3(grid_of(work)), dim3(256), 0, st, x, y, z, n, n_total, out);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

}  // namespace hz
