// TEST INFRASTRUCTURE (never part of libmg_hip.so): the device's closest-point search -- the very statements of
// morphablegraphs_amd/csrc/mg_traj_device.h, compiled for the HOST -- behind a C entry point, so that the CPU suite can hold them to
// the reference's golden vectors and to the oracle without a GPU (tests/test_oracle_golden.py).  Build: tests/native/Makefile.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../morphablegraphs_amd/csrc/mg_traj_device.h"

// the segment polynomials as mg_trajectory_create makes them (csrc/mg_trajectory.hip; catmull_rom_spline.py:66-71, :118-168)
static std::vector<double> make_poly(const double *cp, int n_points) {
    const int n_seg = n_points - 1;
    std::vector<double> pad((size_t)(n_points + 3) * 3);
    for (int d = 0; d < 3; d++) {
        pad[d] = cp[d];
        for (int i = 0; i < n_points; i++) pad[(size_t)(i + 1) * 3 + d] = cp[(size_t)i * 3 + d];
        pad[(size_t)(n_points + 1) * 3 + d] = pad[(size_t)(n_points + 2) * 3 + d] = cp[(size_t)(n_points - 1) * 3 + d];
    }
    static const double M[4][4] = {{-1.0, 3.0, -3.0, 1.0}, {2.0, -5.0, 4.0, -1.0}, {-1.0, 0.0, 1.0, 0.0}, {0.0, 2.0, 0.0, 0.0}};
    std::vector<double> poly((size_t)n_seg * 12 + 3);
    for (int s = 0; s < n_seg; s++)
        for (int r = 0; r < 4; r++)
            for (int d = 0; d < 3; d++) {
                double v = 0.0;
                for (int j = 0; j < 4; j++) v += M[r][j] * pad[(size_t)(s + j) * 3 + d];
                poly[(size_t)s * 12 + r * 3 + d] = 0.5 * v;
            }
    for (int d = 0; d < 3; d++) poly[(size_t)n_seg * 12 + d] = cp[(size_t)(n_points - 1) * 3 + d];
    return poly;
}

// one chain: points (T, 3) -> the parameter and the distance of every frame's point, each search bounded below by the one before
extern "C" int mg_test_lbfgsb_chain(const double *control_points, int32_t n_points, const double *points, int32_t T, double min_u, double *params,
                                    double *distances) {
    if (!control_points || n_points < 2 || !points || T < 0) return -1;
    const std::vector<double> poly = make_poly(control_points, n_points);
    for (int f = 0; f < T; f++) {
        const double q[3] = {points[3 * f], points[3 * f + 1], points[3 * f + 2]};
        const double dist = mg_traj_closest_lbfgsb(poly.data(), n_points - 1, &min_u, q);
        if (params) params[f] = min_u;
        if (distances) distances[f] = dist;
    }
    return 0;
}
