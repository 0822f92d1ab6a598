// Per-environment single-step maps as device functions, templated on the arithmetic type
// R (float for production, double for exactness checks against the fp64 reference).
//
// The reference computes the *wrapped control* in float32 (NumPy keeps
// `hover + hover*np.clip(action,-1,1)` in the float32 dtype of the policy's action) and the
// rest in float64.  That float32 control path is reproduced with individually rounded
// rn_mul/rn_add so the double build agrees with the reference to rounding noise.
//
// Citations: environments/cartpole_env.py, environments/quadrotor_env.py of the reference.
#pragma once
#include "tg_common.hpp"

namespace tg {

// div_/inv_sqrt_/norm2_: the double build keeps the reference's exact operation order (division, sqrt then
// square); the float build uses the 1-ulp hardware reciprocal / rsqrt (v_rcp_f32, v_rsq_f32): a few fp32 ulp
// per step, far inside the fp32-vs-fp64 trajectory noise, and ~half the VALU work of the kernel.
template <typename R> struct Math;
template <> struct Math<float> {
    static constexpr bool kFast = true;
    __device__ static inline float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    __device__ static inline float inv_sqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
    __device__ static inline float sqrt_(float x) { return sqrtf(x); }
    __device__ static inline float atan2_(float y, float x) { return atan2f(y, x); }
    __device__ static inline void sincos_(float x, float* s, float* c) { sincosf(x, s, c); }
    __device__ static inline float abs_(float x) { return fabsf(x); }
    __device__ static inline float tan_(float x) { return tanf(x); }
};
template <> struct Math<double> {
    static constexpr bool kFast = false;
    __device__ static inline double div_(double a, double b) { return a / b; }
    __device__ static inline double inv_sqrt_(double x) { return 1.0 / sqrt(x); }
    __device__ static inline double sqrt_(double x) { return sqrt(x); }
    __device__ static inline double atan2_(double y, double x) { return atan2(y, x); }
    __device__ static inline void sincos_(double x, double* s, double* c) { sincos(x, s, c); }
    __device__ static inline double abs_(double x) { return fabs(x); }
    __device__ static inline double tan_(double x) { return tan(x); }
};

__device__ static inline float clip1(float a) { return fminf(fmaxf(a, -1.0f), 1.0f); }

struct StepOut {
    bool truncated;   // env-level `truncated`
    bool balanced;    // bonus condition held on the new state (drives info['time_balanced'])
};
// `terminated` is always False except for Pendulum, whose episode ends once it has been balanced for more than 5 s
// (pendulum_env.py:151).  Such an env sets kBalanceTerminates and carries `term_steps` in its constants: the number
// of CONSECUTIVE balanced steps after which the fp64-accumulated `_time_balanced` first exceeds the limit.  The
// rollout kernels keep the running count (as -len[env] while the episode runs).

// ---------------------------------------------------------------------------
// CartPole swing-up.  cartpole_env.py:48-49, 51-92, 138-182.
// ---------------------------------------------------------------------------
template <typename R> struct CartPoleEnv {
    static constexpr int S = 5, A = 1;
    static constexpr bool kBalanceTerminates = false;
    struct C {
        R mc, mp, l, g, dt;
        int max_steps, time_trunc_step;
        __host__ static C make(const tg_env_params& p) {
            C c;
            c.mc = (R)p.p[0]; c.mp = (R)p.p[1]; c.l = (R)p.p[2]; c.g = (R)p.p[3]; c.dt = (R)p.timestep;
            c.max_steps = p.max_steps; c.time_trunc_step = p.time_trunc_step;
            return c;
        }
    };

    __device__ static inline StepOut step(const R (&s)[S], const float (&a)[A], const C& c, int steps_after,
                                          R (&o)[S], R& reward) {
#pragma clang fp contract(off)   // every kernel that instantiates this gets the same (unfused) arithmetic: bit-identical states
        using M = Math<R>;
        const float u32 = rn_mul(5.0f, clip1(a[0]));                       // :49
        const R u = (R)u32;                                                   // :60
        const R x = s[0], xd = s[1], sn = s[2], cs = s[3];
        const R thd = fmin(fmax(s[4], (R)-10), (R)10);                        // :58
        const R theta = M::atan2_(sn, cs);                                    // :68
        const R msum = c.mc + c.mp;
        const R alpha = (c.g * sn + cs * ((-u - c.mp * c.l * (thd * thd) * sn) / msum)) /
                        (c.l * ((R)4 / (R)3 - (c.mp * (cs * cs)) / msum));    // :71-73
        const R acc = (u + c.mp * c.l * ((thd * thd) * sn - alpha * cs)) / msum;  // :76
        const R xd_n = xd + acc * c.dt;                                       // :79
        const R x_n = x + xd_n * c.dt;                                        // :80
        const R thd_n = thd + alpha * c.dt;                                   // :82
        const R th_n = theta + thd_n * c.dt;                                  // :83
        R sn_n, cs_n;
        M::sincos_(th_n, &sn_n, &cs_n);
        o[0] = x_n; o[1] = xd_n; o[2] = sn_n; o[3] = cs_n; o[4] = thd_n;     // :85-91

        // reward on the new state; three summands (missing comma at :164-165)
        const R theta_cost = -(cs_n * cs_n * cs_n);
        const R thd_cost = thd_n * thd_n;
        const float energy32 = rn_mul(0.001f, rn_mul(u32, u32));        // float32 in the reference
        const R e1 = (R)-5 * (x_n * x_n);
        const R e2 = (R)-0.5 * (xd_n * xd_n);
        const R e3 = -((R)20 * theta_cost - (R)20) * ((R)1 / ((R)1 + (R)2 * thd_cost)) - (R)energy32;
        R r = c.dt * ((e1 + e2) + e3);                                        // :158-166
        const R ax = M::abs_(x_n);
        StepOut out;
        out.truncated = (ax > (R)1) || (steps_after >= c.time_trunc_step);    // :168
        out.balanced = (ax < (R)0.1) && (cs_n > (R)0.95) && (M::abs_(thd_n) < (R)0.1);  // :173
        if (out.balanced) r += (R)100 * c.dt;                                 // :174
        if (ax > (R)1) r -= (R)50;                                            // :179-180
        reward = r;
        return out;
    }

    // reset: theta0 ~ U(-pi, pi); state [0, 0, sin, cos, 0].  :102-119
    __device__ static inline void reset(const uint32_t (&rnd)[4], R (&o)[S]) {
        const double th = -3.141592653589793 + 6.283185307179586 * Philox::u01d(rnd[0], rnd[1]);
        R sn, cs;
        Math<R>::sincos_((R)th, &sn, &cs);
        o[0] = 0; o[1] = 0; o[2] = sn; o[3] = cs; o[4] = 0;
    }
};

// ---------------------------------------------------------------------------
// QuadPole2D: planar quadrotor + rigid pendulum payload.  quadrotor_env.py:928, 1044-1130, 1132-1223.
// ---------------------------------------------------------------------------
template <typename R> struct QuadPole2DEnv {
    static constexpr int S = 10, A = 2;
    static constexpr bool kBalanceTerminates = false;
    struct C {
        R mq_Lp, mpLp, M, g, dt, bound, balance_radius;
        float hover32, Lq_over_I32, dt32;
        int max_steps;
        __host__ static C make(const tg_env_params& p) {
            C c;
            const double mq = p.p[0], mp = p.p[1], I = p.p[2], Lq = p.p[3], Lp = p.p[4], g = p.p[5];
            c.mq_Lp = (R)(mq * Lp); c.mpLp = (R)(mp * Lp); c.M = (R)(mq + mp); c.g = (R)g;
            c.dt = (R)p.timestep; c.bound = (R)p.p[6]; c.balance_radius = (R)p.p[7];
            c.hover32 = (float)((mq + mp) * g / 2);                           // :895
            c.Lq_over_I32 = (float)(Lq / I);
            c.dt32 = (float)p.timestep;
            c.max_steps = p.max_steps;
            return c;
        }
    };

    __device__ static inline StepOut step(const R (&s)[S], const float (&a)[A], const C& c, int steps_after,
                                          R (&o)[S], R& reward) {
#pragma clang fp contract(off)   // every kernel that instantiates this gets the same (unfused) arithmetic: bit-identical states
        using M = Math<R>;
        const float u1 = rn_add(c.hover32, rn_mul(c.hover32, clip1(a[0])));   // :928 (float32)
        const float u2 = rn_add(c.hover32, rn_mul(c.hover32, clip1(a[1])));
        const R x = s[0], z = s[1], vx = s[2], vz = s[3], sth = s[4], cth = s[5], thd = s[6],
                sph = s[7], cph = s[8], phd = s[9];
        const R F = (R)rn_add(u2, u1);                                     // :1085
        const float ddtheta32 = rn_mul(c.Lq_over_I32, rn_sub(u2, u1));  // :1090 (float32)
        const R ddphi = -F * (sph * cth - sth * cph) / c.mq_Lp;               // :1094
        const R phd2 = phd * phd;
        const R ddx = (-sth * F - c.mpLp * cph * ddphi + c.mpLp * sph * phd2) / c.M;             // :1098
        const R ddz = (cth * F - c.M * c.g - c.mpLp * sph * ddphi - c.mpLp * cph * phd2) / c.M;  // :1101
        const R vx_n = vx + ddx * c.dt;                                       // :1105-1108
        const R vz_n = vz + ddz * c.dt;
        const R thd_n = thd + (R)rn_mul(ddtheta32, c.dt32);
        const R phd_n = phd + ddphi * c.dt;
        const R x_n = x + vx_n * c.dt;                                        // :1111-1112
        const R z_n = z + vz_n * c.dt;
        R sth_n, cth_n, sph_n, cph_n;
        M::sincos_(M::atan2_(sth, cth) + thd * c.dt, &sth_n, &cth_n);         // :1116-1118 (OLD rate)
        M::sincos_(M::atan2_(sph, cph) + phd * c.dt, &sph_n, &cph_n);         // :1122-1124
        o[0] = x_n; o[1] = z_n; o[2] = vx_n; o[3] = vz_n; o[4] = sth_n; o[5] = cth_n; o[6] = thd_n;
        o[7] = sph_n; o[8] = cph_n; o[9] = phd_n;

        const R r2 = x_n * x_n + z_n * z_n;
        const R pos_cost = (M::abs_(x_n) + M::abs_(z_n)) + r2;                // :1186
        const R vel_cost = vx_n * vx_n + vz_n * vz_n;
        const R theta_cost = (R)1 - M::abs_(cth_n);
        const R omega_cost = thd_n * thd_n;
        const R phi_cost = cph_n * cph_n * cph_n;
        const R phid_cost = phd_n * phd_n;
        R acc = -(R)15 * pos_cost;                                            // :1195-1201
        acc = acc + (-(R)0.5 * vel_cost);
        acc = acc + (-(R)5 * theta_cost);
        acc = acc + (-(R)5 * omega_cost);
        acc = acc + (-((R)25 * phi_cost - (R)25) * ((R)1 / ((R)1 + (R)5 * phid_cost)));
        R r = c.dt * acc;
        StepOut out;
        out.balanced = (M::sqrt_(r2) < c.balance_radius) && (cph_n < (R)-0.95) && (M::abs_(phd_n) < (R)0.1);  // :1204
        if (out.balanced) r += (R)100 * c.dt;
        const bool oob = (x_n < -c.bound) || (x_n > c.bound) || (z_n < -c.bound) || (z_n > c.bound);  // :1020-1022
        if (oob) r -= (R)1000 * c.dt;                                         // :1215-1217
        out.truncated = (steps_after >= c.max_steps) || oob;                  // :1220
        reward = r;
        return out;
    }

    // reset: phi0 ~ U(-pi, pi); quad [0,0,0,0,0,1,0], pend [sin, cos, 0].  :930-961
    __device__ static inline void reset(const uint32_t (&rnd)[4], R (&o)[S]) {
        const double ph = -3.141592653589793 + 6.283185307179586 * Philox::u01d(rnd[0], rnd[1]);
        R sn, cs;
        Math<R>::sincos_((R)ph, &sn, &cs);
        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 1; o[6] = 0; o[7] = sn; o[8] = cs; o[9] = 0;
    }
};

// ---------------------------------------------------------------------------
// QuadPole: 3-D quadrotor (quaternion attitude) + tethered payload (quaternion).
// quadrotor_env.py:190-228 (quaternion helpers), 409-413, 417-528, 625-713.
// ---------------------------------------------------------------------------
template <typename R> __device__ static inline void quat_mult(const R (&q)[4], const R (&r)[4], R (&o)[4]) {
#pragma clang fp contract(off)
    o[0] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];             // :196-201
    o[1] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
    o[2] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
    o[3] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
}

template <typename R> struct QuadPoleEnv {
    static constexpr int S = 20, A = 4;
    static constexpr bool kBalanceTerminates = false;
    struct C {
        R m0, m_p, g, L, Ixx, Iyy, Izz, arm, dt, bound, tension_k, m0L, inv_m0, s22, mpL2;
        float hover32, tc32;
        int max_steps;
        __host__ static C make(const tg_env_params& p) {
            C c;
            const double m0 = p.p[0], mp = p.p[1], g = p.p[2], L = p.p[3];
            c.m0 = (R)m0; c.m_p = (R)mp; c.g = (R)g; c.L = (R)L;
            c.Ixx = (R)p.p[4]; c.Iyy = (R)p.p[5]; c.Izz = (R)p.p[6];
            c.tc32 = (float)p.p[7]; c.arm = (R)p.p[8]; c.bound = (R)p.p[9];
            c.dt = (R)p.timestep;
            c.tension_k = (R)(mp / (m0 + mp)); c.m0L = (R)(m0 * L); c.inv_m0 = (R)(1.0 / m0);
            c.s22 = (R)(1.4142135623730951 / 2.0);
            c.mpL2 = (R)(mp * (L * L));   // divisor of :511 (kept as a divisor below)
            c.hover32 = (float)((m0 + mp) * g / 4);                           // :382
            c.max_steps = p.max_steps;
            return c;
        }
    };

    __device__ static inline StepOut step(const R (&s)[S], const float (&a)[A], const C& c, int steps_after,
                                          R (&o)[S], R& reward) {
#pragma clang fp contract(off)   // every kernel that instantiates this gets the same (unfused) arithmetic: bit-identical states
        using M = Math<R>;
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = rn_add(c.hover32, rn_mul(c.hover32, clip1(a[i])));  // :413
        const R u_tot = (R)rn_add(rn_add(rn_add(u[0], u[1]), u[2]), u[3]);                  // :445

        const R q[4] = {s[6], s[7], s[8], s[9]};
        const R om[3] = {s[10], s[11], s[12]};
        const R qp[4] = {s[13], s[14], s[15], s[16]};
        const R omp[3] = {s[17], s[18], s[19]};

        // thrust in the inertial frame = third column of R(q) * u_tot.  :210-219, 466
        const R F[3] = {((R)2 * (q[1] * q[3] + q[0] * q[2])) * u_tot,
                        ((R)2 * (q[2] * q[3] - q[0] * q[1])) * u_tot,
                        ((R)1 - (R)2 * (q[1] * q[1] + q[2] * q[2])) * u_tot};
        // tether direction: rotate [0,0,-1] by q_p.  :221-228, 470
        const R qv[4] = {0, 0, 0, -1};
        const R qpc[4] = {qp[0], -qp[1], -qp[2], -qp[3]};
        R tmp[4], rot[4];
        quat_mult(qp, qv, tmp);
        quat_mult(tmp, qpc, rot);
        const R ut[3] = {rot[1], rot[2], rot[3]};
        const R ud[3] = {omp[1] * ut[2] - omp[2] * ut[1], omp[2] * ut[0] - omp[0] * ut[2],
                         omp[0] * ut[1] - omp[1] * ut[0]};                    // :473
        const R ud2 = ud[0] * ud[0] + ud[1] * ud[1] + ud[2] * ud[2];
        R udn2 = ud2;                                                         // |u_dot|^2
        if constexpr (!M::kFast) { const R udn = M::sqrt_(ud2); udn2 = udn * udn; }   // norm()**2 as written
        const R Fdot = F[0] * ut[0] + F[1] * ut[1] + F[2] * ut[2];
        const R T = c.tension_k * (Fdot - c.m0L * udn2);                      // :476
        const R mg[3] = {0, 0, c.m0 * -c.g};
        R vel_n[3], pos_n[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const R acc = c.inv_m0 * ((mg[i] + F[i]) - T * ut[i]);            // :480
            vel_n[i] = s[3 + i] + acc * c.dt;                                 // :483
            pos_n[i] = s[i] + vel_n[i] * c.dt;                                // :484
        }
        // torques; the gyroscopic term is applied twice in the reference (kept).  :487-500
        const R d_x = (R)rn_sub(rn_sub(rn_add(u[0], u[2]), u[1]), u[3]);
        const R d_y = (R)rn_sub(rn_sub(rn_add(u[2], u[3]), u[0]), u[1]);
        const float d_z32 = rn_sub(rn_sub(rn_add(u[0], u[3]), u[1]), u[2]);
        const R tau_x = c.s22 * d_x * c.arm - (c.Izz - c.Iyy) * om[1] * om[2];
        const R tau_y = c.s22 * d_y * c.arm - (c.Izz - c.Ixx) * om[0] * om[2];
        const R tau_z = (R)rn_mul(c.tc32, d_z32);
        const R Jo[3] = {c.Ixx * om[0], c.Iyy * om[1], c.Izz * om[2]};
        const R cr[3] = {om[1] * Jo[2] - om[2] * Jo[1], om[2] * Jo[0] - om[0] * Jo[2], om[0] * Jo[1] - om[1] * Jo[0]};
        R om_n[3];
        om_n[0] = om[0] + M::div_(tau_x - cr[0], c.Ixx) * c.dt;
        om_n[1] = om[1] + M::div_(tau_y - cr[1], c.Iyy) * c.dt;
        om_n[2] = om[2] + M::div_(tau_z - cr[2], c.Izz) * c.dt;
        // q' = normalise(q + 0.5 (q (x) [0, om']) dt)   (NEW omega).  :504-506
        const R om4[4] = {0, om_n[0], om_n[1], om_n[2]};
        R qd[4], q_n[4];
        quat_mult(q, om4, qd);
        R nn = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { q_n[i] = q[i] + ((R)0.5 * qd[i]) * c.dt; nn += q_n[i] * q_n[i]; }
        if constexpr (M::kFast) {
            const R inv = M::inv_sqrt_(nn);
#pragma unroll
            for (int i = 0; i < 4; ++i) q_n[i] = q_n[i] * inv;
        } else {
            nn = M::sqrt_(nn);
#pragma unroll
            for (int i = 0; i < 4; ++i) q_n[i] = q_n[i] / nn;
        }
        // payload: omega_p' and q_p' = normalise(q_p + 0.5 ([0, omega_p'] (x) q_p) dt).  :511-517
        const R arm_v[3] = {c.L * ut[0], c.L * ut[1], c.L * ut[2]};
        const R frc[3] = {T * ut[0] + (R)0, T * ut[1] + (R)0, T * ut[2] + (-c.g * c.m_p)};
        const R ompd[3] = {M::div_(arm_v[1] * frc[2] - arm_v[2] * frc[1], c.mpL2),
                           M::div_(arm_v[2] * frc[0] - arm_v[0] * frc[2], c.mpL2),
                           M::div_(arm_v[0] * frc[1] - arm_v[1] * frc[0], c.mpL2)};
        R omp_n[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) omp_n[i] = omp[i] + ompd[i] * c.dt;
        const R omp4[4] = {0, omp_n[0], omp_n[1], omp_n[2]};
        R qpd[4], qp_n[4];
        quat_mult(omp4, qp, qpd);
        R np_ = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { qp_n[i] = qp[i] + ((R)0.5 * qpd[i]) * c.dt; np_ += qp_n[i] * qp_n[i]; }
        if constexpr (M::kFast) {
            const R inv = M::inv_sqrt_(np_);
#pragma unroll
            for (int i = 0; i < 4; ++i) qp_n[i] = qp_n[i] * inv;
        } else {
            np_ = M::sqrt_(np_);
#pragma unroll
            for (int i = 0; i < 4; ++i) qp_n[i] = qp_n[i] / np_;
        }

#pragma unroll
        for (int i = 0; i < 3; ++i) { o[i] = pos_n[i]; o[3 + i] = vel_n[i]; o[10 + i] = om_n[i]; o[17 + i] = omp_n[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[6 + i] = q_n[i]; o[13 + i] = qp_n[i]; }

        // reward on the new state.  :669-699
        const R th_q = (R)1 - M::abs_(q_n[0]);
        const R th_p = (R)1 - M::abs_(qp_n[0]);
        const R c_pos = (pos_n[0] * pos_n[0] + pos_n[1] * pos_n[1]) + pos_n[2] * pos_n[2];
        const R c_vel = (vel_n[0] * vel_n[0] + vel_n[1] * vel_n[1]) + vel_n[2] * vel_n[2];
        const R c_rate = (om_n[0] * om_n[0] + om_n[1] * om_n[1]) + om_n[2] * om_n[2];
        const R c_prate = (omp_n[0] * omp_n[0] + omp_n[1] * omp_n[1]) + omp_n[2] * omp_n[2];
        R acc = (R)1;
        acc = acc + M::div_((R)5, (R)1 + (R)10 * c_pos);
        acc = acc + M::div_((R)10, (R)1 + (R)10 * c_vel);
        acc = acc + M::div_((R)0.1, (R)1 + th_q * th_q);
        acc = acc + M::div_((R)5, (R)1 + c_rate);
        acc = acc + M::div_((R)10, (R)1 + (R)10 * (th_p * th_p));
        acc = acc + M::div_((R)1, (R)1 + (R)10 * c_prate);
        R r = c.dt * acc;
        bool oob = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) oob = oob || (pos_n[i] < -c.bound) || (pos_n[i] > c.bound);   // :614-622
        if (oob) r -= (R)10000 * c.dt;                                        // :705-706
        StepOut out;
        out.truncated = (steps_after >= c.max_steps) || oob;                  // :710
        out.balanced = false;                                                 // never updated by QuadPole.step
        reward = r;
        return out;
    }

    // reset: alpha, beta ~ U(-1,1); q_p = normalise(q_y (x) q_x); everything else 0 / identity.  :530-576
    __device__ static inline void reset(const uint32_t (&rnd)[4], R (&o)[S]) {
        const double al = -1.0 + 2.0 * Philox::u01d(rnd[0], rnd[1]);
        const double be = -1.0 + 2.0 * Philox::u01d(rnd[2], rnd[3]);
        R sa, ca, sb, cb;
        Math<R>::sincos_((R)(al / 2), &sa, &ca);
        Math<R>::sincos_((R)(be / 2), &sb, &cb);
        const R qx[4] = {ca, sa, 0, 0}, qy[4] = {cb, 0, sb, 0};
        R qp[4];
        quat_mult(qy, qx, qp);
        const R n = Math<R>::sqrt_(qp[0] * qp[0] + qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
#pragma unroll
        for (int i = 0; i < S; ++i) o[i] = 0;
        o[6] = 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[13 + i] = qp[i] / n;
    }
};

// ---------------------------------------------------------------------------
// Pendulum.  pendulum_env.py:45-46, 48-74, 124-158.
// ---------------------------------------------------------------------------
template <typename R> struct PendulumEnv {
    static constexpr int S = 3, A = 1;
    static constexpr bool kBalanceTerminates = true;
    struct C {
        R mgl, inv_ml2, dt;
        int max_steps, time_trunc_step, term_steps, swingup;
        __host__ static C make(const tg_env_params& p) {
            C c;
            const double mass = p.p[0], length = p.p[1], gravity = p.p[2];
            c.mgl = (R)(mass * gravity * length);                             // python-float product, :61
            c.inv_ml2 = (R)(1.0 / (mass * (length * length)));
            c.dt = (R)p.timestep;
            c.max_steps = p.max_steps; c.time_trunc_step = p.time_trunc_step;
            c.swingup = p.p[3] != 0.0; c.term_steps = (int)p.p[4];
            return c;
        }
    };

    __device__ static inline StepOut step(const R (&s)[S], const float (&a)[A], const C& c, int steps_after,
                                          R (&o)[S], R& reward) {
#pragma clang fp contract(off)   // every kernel that instantiates this gets the same (unfused) arithmetic: bit-identical states
        using M = Math<R>;
        const float a32 = clip1(a[0]);                                        // :46 (float32, no scaling)
        const R u = (R)a32;                                                   // promoted by the float64 scalar at :61
        const R thd = fmin(fmax(s[2], (R)-10), (R)10);                        // :57
        const R theta = M::atan2_(s[0], s[1]);                                // :59
        R sn, cs;
        M::sincos_(theta, &sn, &cs);
        const R alpha = c.inv_ml2 * (u - c.mgl * sn);                         // :61
        const R thd_n = thd + alpha * c.dt;                                   // :63
        const R th_n = theta + thd_n * c.dt;                                  // :64
        R sn_n, cs_n;
        M::sincos_(th_n, &sn_n, &cs_n);
        o[0] = sn_n; o[1] = cs_n; o[2] = thd_n;                               // :68-72
        StepOut out;
        out.balanced = cs_n <= (R)-0.99;                                      // :135
        out.truncated = steps_after >= c.time_trunc_step;                     // :150, float-accumulated time
        const float energy32 = rn_mul(-0.001f, rn_mul(a32, a32));           // float32 in the reference, :145
        const R e1 = (R)-10 * M::sqrt_(M::abs_((R)-1 - cs_n));                // :143 (x ** 0.5)
        const R e2 = (R)-0.1 * (thd_n * thd_n);                               // :144
        R r = c.dt * ((e1 + e2) + (R)energy32);                               // :142
        if (out.balanced) r += (R)1;                                          // :148-149 (time_balanced > 0 <=> balanced now)
        reward = r;
        return out;
    }

    // reset: theta0 ~ U(pi - 0.05, pi + 0.05), or U(-pi, pi) with swingup; state [sin, cos, 0].  :86-106
    __device__ static inline void reset(const uint32_t (&rnd)[4], R (&o)[S]) { reset(rnd, o, 0); }
    __device__ static inline void reset(const uint32_t (&rnd)[4], R (&o)[S], int swingup) {
        const double u = Philox::u01d(rnd[0], rnd[1]);
        const double th = swingup ? -3.141592653589793 + 6.283185307179586 * u : (3.141592653589793 - 0.05) + 0.1 * u;
        R sn, cs;
        Math<R>::sincos_((R)th, &sn, &cs);
        o[0] = sn; o[1] = cs; o[2] = 0;
    }
};

}  // namespace tg
