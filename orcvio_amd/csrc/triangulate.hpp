// triangulate.hpp -- feature triangulation on the device-resident window.
//
// Reference: Feature::checkMotion, ::initializePosition, ::triangulate_position, ::generateInitialGuess, ::cost,
// ::jacobian (include/orcvio/feat/feature.hpp:270-449, 583-719), called per lost feature from
// OrcVIO::removeLostFeatures (src/orcvio.cpp:2258-2270).  One wavefront per track, lane t <-> listed observation t:
// the three-parameter (alpha, beta, rho) Levenberg-Marquardt in the last camera's frame is a handful of wave
// reductions per iteration; every lane carries the same (alpha, beta, rho), lambda and loop counters, so the
// reference's do-while control flow is wave-uniform.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "msckf_math.hpp"

namespace orcvio_amd {

struct TriArgs {
    const double* poses;      // [N][POSE_STRIDE]
    const int* obs_ptr;       // [F+1]
    const int* obs_clone;     // [nobs]
    const double* obs_z;      // [nobs][2]
    const int* is_init;       // [F] or nullptr: start from p_w[j] (feature.hpp:604-606), no motion check
    double* p_w;              // [F][3] in: prior (if is_init), out: triangulated position of valid tracks
    int* valid;               // [F]
    int* flags;               // [F] 1 no motion / too few observations, 2 negative depth, 4 big projection error
    double* solution;         // [F][3] (alpha, beta, rho) in the last camera's frame
    double* cost;             // [F]
    int* skip;                // [F] or nullptr: 1 where the track must not enter the update
    double translation_threshold, huber_epsilon, estimation_precision, initial_damping, cost_threshold, init_final_dist_threshold;
    int outer_max, inner_max;
    int F;
};

// sum over the wavefront by DPP moves (two v_mov_b32_dpp + one v_add_f64 per step, then v_readlane of lane 63) instead of six
// ds_bpermute shuffle pairs: the Levenberg-Marquardt loop below is a chain of these reductions
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double tri_dpp_move(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tri_wave_sum(double x) {
    x += tri_dpp_move<0xB1, 0xF>(x);    // quad_perm [1,0,3,2]
    x += tri_dpp_move<0x4E, 0xF>(x);    // quad_perm [2,3,0,1]
    x += tri_dpp_move<0x141, 0xF>(x);   // row_half_mirror
    x += tri_dpp_move<0x140, 0xF>(x);   // row_mirror
    x += tri_dpp_move<0x142, 0xA>(x);   // row_bcast:15 -> rows 1, 3
    x += tri_dpp_move<0x143, 0xC>(x);   // row_bcast:31 -> rows 2, 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void k_triangulate(TriArgs p) {
    const int j = blockIdx.x, t = threadIdx.x;
    const int lo = p.obs_ptr[j];
    const int M = p.obs_ptr[j + 1] - lo;
    const bool init = p.is_init && p.is_init[j] != 0;
    if (M < 2) {   // nothing to triangulate from (the reference never gets here: least_Obs_Num, src/orcvio.cpp:2252)
        if (t == 0) {
            p.valid[j] = 0; p.flags[j] = 1; p.cost[j] = NAN;
            p.solution[3 * j] = p.solution[3 * j + 1] = p.solution[3 * j + 2] = NAN;
            if (p.skip) p.skip[j] = 1;
        }
        return;
    }
    const bool live = t < M;
    // camera pose of this lane's observation: R_c2w = R_b2w R_b2c^T, t_c_w = t_b_w + R_b2w t_c_b (src/orcvio.cpp:954-961)
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tc[3] = {0, 0, 0}, z[2] = {0, 0};
    if (live) {
        const int o = lo + t;
        const double* ps = p.poses + (size_t)p.obs_clone[o] * POSE_STRIDE;
        const double* Rbw = ps + POSE_R_B2W;
        const double* Rbc = ps + POSE_R_B2C;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) R[3 * a + b] = Rbw[3 * a] * Rbc[3 * b] + Rbw[3 * a + 1] * Rbc[3 * b + 1] + Rbw[3 * a + 2] * Rbc[3 * b + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            tc[a] = ps[POSE_T_B_W + a] + Rbw[3 * a] * ps[POSE_T_C_B] + Rbw[3 * a + 1] * ps[POSE_T_C_B + 1] + Rbw[3 * a + 2] * ps[POSE_T_C_B + 2];
        z[0] = p.obs_z[2 * o];
        z[1] = p.obs_z[2 * o + 1];
    }
    // last listed camera (the anchor)
    double Rl[9], tl[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) Rl[a] = __shfl(R[a], M - 1);
#pragma unroll
    for (int a = 0; a < 3; ++a) tl[a] = __shfl(tc[a], M - 1);
    const double zl0 = __shfl(z[0], M - 1), zl1 = __shfl(z[1], M - 1);

    // ---- checkMotion (feature.hpp:354-397): first and last listed observation ------------------------
    if (!init) {
        double mo = 0.0;
        if (t == 0) {
            const double nz = 1.0 / sqrt(z[0] * z[0] + z[1] * z[1] + 1.0);
            const double d0 = z[0] * nz, d1 = z[1] * nz, d2 = nz;
            double d[3], tr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { d[a] = R[3 * a] * d0 + R[3 * a + 1] * d1 + R[3 * a + 2] * d2; tr[a] = tl[a] - tc[a]; }
            const double par = tr[0] * d[0] + tr[1] * d[1] + tr[2] * d[2];
            const double o0 = tr[0] - par * d[0], o1 = tr[1] - par * d[1], o2 = tr[2] - par * d[2];
            mo = sqrt(o0 * o0 + o1 * o1 + o2 * o2);
        }
        mo = __shfl(mo, 0);
        if (!(mo > p.translation_threshold)) {
            if (t == 0) {
                p.valid[j] = 0; p.flags[j] = 1; p.cost[j] = NAN;
                p.solution[3 * j] = p.solution[3 * j + 1] = p.solution[3 * j + 2] = NAN;
                if (p.skip) p.skip[j] = 1;
            }
            return;
        }
    }

    // ---- pose_i^-1 * pose_last: last camera frame -> camera i frame (feature.hpp:590-592) --------------
    double Rr[9], tr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) Rr[3 * a + b] = R[a] * Rl[b] + R[3 + a] * Rl[3 + b] + R[6 + a] * Rl[6 + b];
        tr[a] = R[a] * (tl[0] - tc[0]) + R[3 + a] * (tl[1] - tc[1]) + R[6 + a] * (tl[2] - tc[2]);
    }
    // ---- initial guess ------------------------------------------------------------------------------
    double ip[3];
    if (!init) {   // generateInitialGuess(cam_poses[0], z_last, z_0) (feature.hpp:332-352, 597-599): on lane 0
        double g0 = 0, g1 = 0, g2 = 0;
        if (t == 0) {
            const double m0 = Rr[0] * zl0 + Rr[1] * zl1 + Rr[2], m1 = Rr[3] * zl0 + Rr[4] * zl1 + Rr[5], m2 = Rr[6] * zl0 + Rr[7] * zl1 + Rr[8];
            const double A0 = m0 - z[0] * m2, A1 = m1 - z[1] * m2;
            const double b0 = z[0] * tr[2] - tr[0], b1 = z[1] * tr[2] - tr[1];
            const double depth = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1);
            g0 = zl0 * depth; g1 = zl1 * depth; g2 = depth;
        }
        ip[0] = __shfl(g0, 0); ip[1] = __shfl(g1, 0); ip[2] = __shfl(g2, 0);
    } else {       // T_c_w_last^-1 * position (feature.hpp:604-606)
        const double d0 = p.p_w[3 * j] - tl[0], d1 = p.p_w[3 * j + 1] - tl[1], d2 = p.p_w[3 * j + 2] - tl[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) ip[a] = Rl[a] * d0 + Rl[3 + a] * d1 + Rl[6 + a] * d2;
    }
    double x0 = ip[0] / ip[2], x1 = ip[1] / ip[2], x2 = 1.0 / ip[2];

    auto cost_of = [&](double a, double b, double r) -> double {   // feature.hpp:270-290, summed over the track
        double e = 0.0;
        if (live) {
            const double h0 = Rr[0] * a + Rr[1] * b + Rr[2] + r * tr[0];
            const double h1 = Rr[3] * a + Rr[4] * b + Rr[5] + r * tr[1];
            const double h2 = Rr[6] * a + Rr[7] * b + Rr[8] + r * tr[2];
            const double e0 = h0 / h2 - z[0], e1 = h1 / h2 - z[1];
            e = e0 * e0 + e1 * e1;
        }
        return tri_wave_sum(e);
    };

    // ---- Levenberg-Marquardt (feature.hpp:608-679) ---------------------------------------------------
    double lambda = p.initial_damping;
    int inner = 0, outer = 0;
    bool reduced = false;
    double delta_norm = 0.0;
    double total = cost_of(x0, x1, x2);
    bool go_outer = true;
    while (go_outer) {
        double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (live) {   // feature.hpp:292-330
            const double h0 = Rr[0] * x0 + Rr[1] * x1 + Rr[2] + x2 * tr[0];
            const double h1 = Rr[3] * x0 + Rr[4] * x1 + Rr[5] + x2 * tr[1];
            const double h2 = Rr[6] * x0 + Rr[7] * x1 + Rr[8] + x2 * tr[2];
            const double ih = 1.0 / h2, ih2 = 1.0 / (h2 * h2);
            // W = [R(:,0) R(:,1) t]
            const double J00 = ih * Rr[0] - h0 * ih2 * Rr[6], J01 = ih * Rr[1] - h0 * ih2 * Rr[7], J02 = ih * tr[0] - h0 * ih2 * tr[2];
            const double J10 = ih * Rr[3] - h1 * ih2 * Rr[6], J11 = ih * Rr[4] - h1 * ih2 * Rr[7], J12 = ih * tr[1] - h1 * ih2 * tr[2];
            const double r0 = h0 / h2 - z[0], r1 = h1 / h2 - z[1];
            const double e = sqrt(r0 * r0 + r1 * r1);
            double w2 = 1.0;
            if (!(e <= p.huber_epsilon)) { const double w = sqrt(2.0 * p.huber_epsilon / e); w2 = w * w; }
            a00 = w2 * (J00 * J00 + J10 * J10); a01 = w2 * (J00 * J01 + J10 * J11); a02 = w2 * (J00 * J02 + J10 * J12);
            a11 = w2 * (J01 * J01 + J11 * J11); a12 = w2 * (J01 * J02 + J11 * J12); a22 = w2 * (J02 * J02 + J12 * J12);
            b0 = w2 * (J00 * r0 + J10 * r1); b1 = w2 * (J01 * r0 + J11 * r1); b2 = w2 * (J02 * r0 + J12 * r1);
        }
        a00 = tri_wave_sum(a00); a01 = tri_wave_sum(a01); a02 = tri_wave_sum(a02);
        a11 = tri_wave_sum(a11); a12 = tri_wave_sum(a12); a22 = tri_wave_sum(a22);
        b0 = tri_wave_sum(b0); b1 = tri_wave_sum(b1); b2 = tri_wave_sum(b2);
        bool go_inner = true;
        while (go_inner) {
            // (A + lambda I) delta = b, symmetric 3x3: L D L^T without pivoting (the reference's Eigen::LDLT pivots; the
            // matrix is SPD plus a positive shift, so both are backward stable and agree to rounding)
            const double m00 = a00 + lambda, m11 = a11 + lambda, m22 = a22 + lambda;
            const double l10 = a01 / m00, l20 = a02 / m00;
            const double d1 = m11 - l10 * a01;
            const double l21 = (a12 - l20 * a01) / d1;
            const double d2 = m22 - l20 * a02 - l21 * l21 * d1;
            const double y0 = b0, y1 = b1 - l10 * y0, y2 = b2 - l20 * y0 - l21 * y1;
            const double q2 = y2 / d2;
            const double q1 = y1 / d1 - l21 * q2;
            const double q0 = y0 / m00 - l10 * q1 - l20 * q2;
            const double n0 = x0 - q0, n1 = x1 - q1, n2 = x2 - q2;
            delta_norm = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
            const double nc = cost_of(n0, n1, n2);
            if (nc < total) {
                reduced = true;
                x0 = n0; x1 = n1; x2 = n2;
                total = nc;
                lambda = lambda / 10 > 1e-10 ? lambda / 10 : 1e-10;
            } else {
                reduced = false;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
            go_inner = (inner < p.inner_max) && !reduced;
            ++inner;
        }
        inner = 0;
        go_outer = (outer < p.outer_max) && (delta_norm > p.estimation_precision);
        ++outer;
    }

    // ---- validity (feature.hpp:681-716) ----------------------------------------------------------------
    const double f0 = x0 / x2, f1 = x1 / x2, f2 = 1.0 / x2;
    const bool behind = live && !((Rr[6] * f0 + Rr[7] * f1 + Rr[8] * f2 + tr[2]) > 0.0);
    int fl = 0;
    if (__any(behind)) fl |= 2;
    const double dd0 = f0 - ip[0], dd1 = f1 - ip[1], dd2 = f2 - ip[2];
    if (sqrt(dd0 * dd0 + dd1 * dd1 + dd2 * dd2) > p.init_final_dist_threshold) fl |= 4;
    if (total / (2.0 * M * M) > p.cost_threshold) fl |= 4;
    if (t == 0) {
        const bool ok = fl == 0;
        p.valid[j] = ok ? 1 : 0;
        p.flags[j] = fl;
        p.cost[j] = total;
        p.solution[3 * j] = x0; p.solution[3 * j + 1] = x1; p.solution[3 * j + 2] = x2;
        if (ok || !init) {   // position = T_c_w_last * final_position (feature.hpp:433); an initialised feature keeps its prior on failure
#pragma unroll
            for (int a = 0; a < 3; ++a) p.p_w[3 * j + a] = ok ? (Rl[3 * a] * f0 + Rl[3 * a + 1] * f1 + Rl[3 * a + 2] * f2 + tl[a]) : NAN;
        }
        if (p.skip) p.skip[j] = ok ? 0 : 1;
    }
}

}  // namespace orcvio_amd
