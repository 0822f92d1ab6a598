"""CPU oracle: batched NumPy restatement of the reference environments' step/reset.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All citations are
`path:line` in the reference checkout (Dyllon-Preston/trajopt-grpo @ 2025-06-13).

Every function steps N independent environments at once; arrays are
`(N, S)` states, `(N, A)` float32 actions (the reference's policies emit
float32 actions, policies/actor_critic.py:138,289).  The reference computes the
*wrapped control* in float32 (NumPy weak-scalar promotion keeps
`hover + hover*np.clip(action, -1, 1)` in the action's dtype) and everything
downstream in float64; that mixed flow is restated literally so the fp64 oracle
agrees with the reference to rounding noise (tests pin <= 1e-12).

Pass `dtype=np.float32` to run the same arithmetic in float32 (used only to
size tolerances for the fp32 GPU kernels; goldens are fp64).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# ---------------------------------------------------------------------------
# CartPole  (environments/cartpole_env.py)
# ---------------------------------------------------------------------------
CARTPOLE_DEFAULTS = dict(masscart=1.0, masspole=1.0, length=0.5,
                         gravity=9.80665, timestep=0.02)  # cartpole_env.py:10-14


def cartpole_time_trunc_step(max_steps: int, timestep: float = 0.02) -> int:
    """Smallest step count k at which `_time > max_time` first fires.

    The reference accumulates `_time += timestep` in fp64 and compares with
    `max_steps*timestep` (cartpole_env.py:27,151,168).  Reproduced literally.
    """
    max_time = max_steps * timestep
    t = 0
    k = 0
    while True:
        k += 1
        t += timestep
        if t > max_time:
            return k


def cartpole_step(state, action, steps, time_balanced, *, max_steps=500,
                  masscart=1.0, masspole=1.0, length=0.5, gravity=9.80665,
                  timestep=0.02, dtype=np.float64):
    """One CartPole.step for N envs.  cartpole_env.py:48-49 (_wrap_action),
    :51-92 (_dynamics), :138-182 (step).

    Returns (next_state (N,5), reward (N,), truncated (N,) bool,
             steps_next (N,) int, time_balanced_next (N,))."""
    R = dtype
    state = np.asarray(state, dtype=R)
    a32 = np.asarray(action, dtype=F32).reshape(len(state), -1)
    u32 = F32(5) * np.clip(a32, F32(-1), F32(1))            # :49, float32
    u = u32[:, 0].astype(R)                                  # :60 control[0]

    x, xdot, s, c, thd = (state[:, i] for i in range(5))    # :56
    thd = np.clip(thd, R(-10), R(10))                        # :58
    mc, mp, l, g, dt = (R(v) for v in (masscart, masspole, length, gravity, timestep))
    theta = np.arctan2(s, c)                                 # :68
    alpha = (g * s + c * ((-u - mp * l * thd ** 2 * s) / (mc + mp))) / (
        l * (R(4) / R(3) - (mp * c ** 2) / (mc + mp)))       # :71-73
    acc = (u + mp * l * (thd ** 2 * s - alpha * c)) / (mc + mp)  # :76
    xdot_n = xdot + acc * dt                                 # :79
    x_n = x + xdot_n * dt                                    # :80
    thd_n = thd + alpha * dt                                 # :82
    theta_n = theta + thd_n * dt                             # :83
    s_n, c_n = np.sin(theta_n), np.cos(theta_n)              # :85-91
    nxt = np.stack([x_n, xdot_n, s_n, c_n, thd_n], axis=1)

    # reward on the new state, :155-166 (three list entries: the missing comma
    # at :164-165 folds the energy term into the balancing term)
    theta_cost = -c_n ** 3
    thd_cost = thd_n ** 2
    energy32 = F32(0.001) * (u32 ** 2).sum(axis=1, dtype=F32)    # float32 (weak scalar)
    e1 = R(-5) * x_n ** 2
    e2 = R(-0.5) * xdot_n ** 2
    e3 = -(R(20.0) * theta_cost - R(20.0)) * (R(1) / (R(1) + R(2) * thd_cost)) - energy32.astype(R)
    reward = dt * ((e1 + e2) + e3)

    steps_n = np.asarray(steps).astype(np.int64) + 1         # :150
    # :168  truncated = |x|>1 or _time > max_time  (float-accumulated time)
    truncated = (np.abs(x_n) > 1) | (steps_n >= cartpole_time_trunc_step(max_steps, timestep))
    balanced = (np.abs(x_n) < R(0.1)) & (c_n > R(0.95)) & (np.abs(thd_n) < R(0.1))  # :173
    reward = np.where(balanced, reward + R(100) * dt, reward)        # :174
    tb = np.where(balanced, np.asarray(time_balanced, dtype=R) + dt, R(0))  # :175-177
    reward = np.where(np.abs(x_n) > 1, reward - R(50), reward)       # :179-180
    return nxt, reward, truncated, steps_n, tb


def cartpole_reset(theta0, dtype=np.float64):
    """cartpole_env.py:102-119: state [0,0,sin t0,cos t0,0], t0 ~ U(-pi,pi)."""
    theta0 = np.asarray(theta0, dtype=dtype)
    z = np.zeros_like(theta0)
    return np.stack([z, z, np.sin(theta0), np.cos(theta0), z], axis=1)


# ---------------------------------------------------------------------------
# QuadPole2D  (environments/quadrotor_env.py:867-1223)
# ---------------------------------------------------------------------------
QP2D = dict(mq=1.5, mp=0.5, I=4e-1, Lq=0.5, Lp=0.75, gravity=9.80665,
            bound=2.0, balance_radius=0.25)                  # :875-895


def quadpole2d_step(state, action, steps, time_balanced, *, max_steps=500,
                    timestep=0.02, dtype=np.float64):
    """One QuadPole2D.step for N envs.  quadrotor_env.py:928 (_wrap_action),
    :1044-1130 (_dynamics), :1132-1223 (step).

    Returns (next_state (N,10), reward, truncated, steps_next, time_balanced_next)."""
    R = dtype
    state = np.asarray(state, dtype=R)
    a32 = np.asarray(action, dtype=F32).reshape(len(state), 2)
    mq, mp, I, Lq, Lp, g = (QP2D[k] for k in ("mq", "mp", "I", "Lq", "Lp", "gravity"))
    hover32 = F32((mq + mp) * g / 2)                          # :895 (python float, weak)
    u32 = hover32 + hover32 * np.clip(a32, F32(-1), F32(1))   # :928 float32
    u1_32, u2_32 = u32[:, 0], u32[:, 1]

    x, z, vx, vz, sth, cth, thd, sph, cph, phd = (state[:, i] for i in range(10))
    dt = R(timestep)
    F32sum = u2_32 + u1_32                                    # :1085 float32
    F = F32sum.astype(R)
    M = R(mq + mp)                                            # :1087
    ddtheta32 = F32(Lq / I) * (u2_32 - u1_32)                 # :1090 float32 (weak scalar)
    ddphi = -F * (sph * cth - sth * cph) / R(mq * Lp)         # :1094
    mpLp = R(mp * Lp)
    ddx = (-sth * F - mpLp * cph * ddphi + mpLp * sph * (phd ** 2)) / M      # :1098
    ddz = (cth * F - M * R(g) - mpLp * sph * ddphi - mpLp * cph * (phd ** 2)) / M  # :1101
    vx_n = vx + ddx * dt                                      # :1105
    vz_n = vz + ddz * dt
    thd_n = thd + (ddtheta32 * F32(timestep)).astype(R)       # :1107 (float32 product)
    phd_n = phd + ddphi * dt
    x_n = x + vx_n * dt                                       # :1111-1112
    z_n = z + vz_n * dt
    theta = np.arctan2(sth, cth)                              # :1116 (old rate!)
    sth_n, cth_n = np.sin(theta + thd * dt), np.cos(theta + thd * dt)
    phi = np.arctan2(sph, cph)                                # :1122
    sph_n, cph_n = np.sin(phi + phd * dt), np.cos(phi + phd * dt)
    nxt = np.stack([x_n, z_n, vx_n, vz_n, sth_n, cth_n, thd_n, sph_n, cph_n, phd_n], axis=1)

    # costs on the new state, :1186-1191
    pos_cost = (np.abs(x_n) + np.abs(z_n)) + (x_n ** 2 + z_n ** 2)
    vel_cost = vx_n ** 2 + vz_n ** 2
    theta_cost = R(1) - np.abs(cth_n)
    omega_cost = thd_n ** 2
    phi_cost = cph_n ** 3
    phid_cost = phd_n ** 2
    terms = [-R(15.0) * pos_cost, -R(0.5) * vel_cost, -R(5.0) * theta_cost, -R(5) * omega_cost,
             -(R(25.0) * phi_cost - R(25.0)) * (R(1) / (R(1) + R(5) * phid_cost))]   # :1195-1201
    acc = terms[0]
    for t_ in terms[1:]:
        acc = acc + t_
    reward = dt * acc
    balanced = ((x_n ** 2 + z_n ** 2) ** R(0.5) < R(QP2D["balance_radius"])) & \
               (cph_n < R(-0.95)) & (np.abs(phd_n) < R(0.1))                # :1204
    reward = np.where(balanced, reward + R(100) * dt, reward)
    tb = np.where(balanced, np.asarray(time_balanced, dtype=R) + dt, R(0))
    steps_n = np.asarray(steps).astype(np.int64) + 1                         # :1211
    b = R(QP2D["bound"])
    oob = (x_n < -b) | (x_n > b) | (z_n < -b) | (z_n > b)                    # :1020-1022
    reward = np.where(oob, reward - R(1000) * dt, reward)                    # :1215-1217
    truncated = (steps_n >= max_steps) | oob                                 # :1220
    return nxt, reward, truncated, steps_n, tb


def quadpole2d_reset(phi0, dtype=np.float64):
    """quadrotor_env.py:930-961: quad [0,0,0,0,0,1,0], pend [sin p0, cos p0, 0]."""
    phi0 = np.asarray(phi0, dtype=dtype)
    z = np.zeros_like(phi0)
    o = np.ones_like(phi0)
    return np.stack([z, z, z, z, z, o, z, np.sin(phi0), np.cos(phi0), z], axis=1)


# ---------------------------------------------------------------------------
# QuadPole  (environments/quadrotor_env.py:353-713) + quaternion helpers :190-228
# ---------------------------------------------------------------------------
QP3D = dict(mass=1.5, load_mass=0.5, gravity=9.80665, tether=0.5, Ixx=4e-1, Iyy=4e-1,
            Izz=2.5e-1, torque_constant=0.1, arm=0.5, bound=1.5)   # :362-382


def quat_mult(q, r):
    """Hamilton product, scalar-first, batched on axis 0.  quadrotor_env.py:190-202."""
    q0, q1, q2, q3 = (q[:, i] for i in range(4))
    r0, r1, r2, r3 = (r[:, i] for i in range(4))
    return np.stack([
        q0 * r0 - q1 * r1 - q2 * r2 - q3 * r3,
        q0 * r1 + q1 * r0 + q2 * r3 - q3 * r2,
        q0 * r2 - q1 * r3 + q2 * r0 + q3 * r1,
        q0 * r3 + q1 * r2 - q2 * r1 + q3 * r0], axis=1)


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                     a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def quadpole_step(state, action, steps, time_balanced, *, max_steps=500,
                  timestep=0.02, dtype=np.float64):
    """One QuadPole.step for N envs.  quadrotor_env.py:409-413 (_wrap_action),
    :417-528 (_dynamics), :625-713 (step).

    State (N,20) = [pos3, vel3, q4, omega3, q_p4, omega_p3].
    Returns (next_state, reward, truncated, steps_next, time_balanced_next)."""
    R = dtype
    P = QP3D
    state = np.asarray(state, dtype=R)
    n = len(state)
    a32 = np.asarray(action, dtype=F32).reshape(n, 4)
    m0, m_p, g, L = P["mass"], P["load_mass"], P["gravity"], P["tether"]
    Ixx, Iyy, Izz = P["Ixx"], P["Iyy"], P["Izz"]
    hover32 = F32((m0 + m_p) * g / 4)                          # :382
    u32 = hover32 + hover32 * np.clip(a32, F32(-1), F32(1))    # :413 float32
    u1, u2, u3, u4 = (u32[:, i] for i in range(4))
    u_tot = (((u1 + u2) + u3) + u4).astype(R)                  # :445 float32 sum
    dt = R(timestep)

    pos, vel, q, om, qp, omp = (state[:, 0:3], state[:, 3:6], state[:, 6:10],
                                state[:, 10:13], state[:, 13:17], state[:, 17:20])
    q0, q1, q2, q3 = (q[:, i] for i in range(4))
    # third column of R(q) times u_tot, :210-219,466
    Fx = (R(2) * (q1 * q3 + q0 * q2)) * u_tot
    Fy = (R(2) * (q2 * q3 - q0 * q1)) * u_tot
    Fz = (R(1) - R(2) * (q1 ** 2 + q2 ** 2)) * u_tot
    Fth = np.stack([Fx, Fy, Fz], axis=1)
    # tether direction: rotate [0,0,-1] by q_p, :221-228,470
    zeros = np.zeros(n, dtype=R)
    qv = np.stack([zeros, zeros, zeros, -np.ones(n, dtype=R)], axis=1)
    qpc = qp * np.array([1, -1, -1, -1], dtype=R)
    ut = quat_mult(quat_mult(qp, qv), qpc)[:, 1:]
    ud = _cross(omp, ut)                                       # :473
    ud_norm = np.sqrt(ud[:, 0] * ud[:, 0] + ud[:, 1] * ud[:, 1] + ud[:, 2] * ud[:, 2])
    Fdot = Fth[:, 0] * ut[:, 0] + Fth[:, 1] * ut[:, 1] + Fth[:, 2] * ut[:, 2]
    T = R(m_p / (m0 + m_p)) * (Fdot - R(m0 * L) * ud_norm ** 2)   # :476
    gvec = np.array([0.0, 0.0, -g], dtype=R)
    mg = R(m0) * gvec
    acc = R(1 / m0) * ((mg[None, :] + Fth) - T[:, None] * ut)  # :480
    vel_n = vel + acc * dt                                     # :483
    pos_n = pos + vel_n * dt                                   # :484

    s22 = np.sqrt(2) / 2
    tau_x = (R(s22) * (((u1 + u3) - u2) - u4).astype(R) * R(P["arm"])
             - R(Izz - Iyy) * om[:, 1] * om[:, 2])             # :487
    tau_y = (R(s22) * (((u3 + u4) - u1) - u2).astype(R) * R(P["arm"])
             - R(Izz - Ixx) * om[:, 0] * om[:, 2])             # :488
    tau_z = (F32(P["torque_constant"]) * (((u1 + u4) - u2) - u3)).astype(R)  # :489 float32 product
    Jom = np.stack([R(Ixx) * om[:, 0], R(Iyy) * om[:, 1], R(Izz) * om[:, 2]], axis=1)  # :493
    cr = _cross(om, Jom)                                       # :494
    omd = np.stack([(tau_x - cr[:, 0]) / R(Ixx), (tau_y - cr[:, 1]) / R(Iyy),
                    (tau_z - cr[:, 2]) / R(Izz)], axis=1)      # :495-499
    om_n = om + omd * dt                                       # :500
    om4 = np.concatenate([zeros[:, None], om_n], axis=1)
    q_n = q + (R(0.5) * quat_mult(q, om4)) * dt                # :504-505
    q_n = q_n / np.sqrt((q_n * q_n).sum(axis=1))[:, None]      # :506

    arm_v = R(L) * ut
    frc = T[:, None] * ut + (gvec * R(m_p))[None, :]
    ompd = _cross(arm_v, frc) / R(m_p * L ** 2)                # :511
    omp_n = omp + ompd * dt                                    # :512
    omp4 = np.concatenate([zeros[:, None], omp_n], axis=1)
    qp_n = qp + (R(0.5) * quat_mult(omp4, qp)) * dt            # :515-516 (left multiplication)
    qp_n = qp_n / np.sqrt((qp_n * qp_n).sum(axis=1))[:, None]  # :517
    nxt = np.concatenate([pos_n, vel_n, q_n, om_n, qp_n, omp_n], axis=1)   # :526

    # reward on the new state, :669-699
    th_q = R(1) - np.abs(q_n[:, 0])
    th_p = R(1) - np.abs(qp_n[:, 0])
    c_pos = (pos_n[:, 0] ** 2 + pos_n[:, 1] ** 2) + pos_n[:, 2] ** 2
    c_vel = (vel_n[:, 0] ** 2 + vel_n[:, 1] ** 2) + vel_n[:, 2] ** 2
    c_rate = (om_n[:, 0] ** 2 + om_n[:, 1] ** 2) + om_n[:, 2] ** 2
    c_prate = (omp_n[:, 0] ** 2 + omp_n[:, 1] ** 2) + omp_n[:, 2] ** 2
    terms = [np.ones(n, dtype=R),
             R(5) / (R(1) + R(10) * c_pos),
             R(10) / (R(1) + R(10) * c_vel),
             R(0.1) / (R(1) + th_q ** 2),
             R(5.0) / (R(1) + c_rate),
             R(10) / (R(1) + R(10) * th_p ** 2),
             R(1) / (R(1) + R(10) * c_prate)]
    acc_r = terms[0]
    for t_ in terms[1:]:
        acc_r = acc_r + t_
    reward = dt * acc_r
    b = R(P["bound"])
    oob = ((pos_n < -b) | (pos_n > b)).any(axis=1)             # :614-622
    reward = np.where(oob, reward - R(10_000) * dt, reward)    # :705-706
    steps_n = np.asarray(steps).astype(np.int64) + 1           # :648
    truncated = (steps_n >= max_steps) | oob                   # :710
    tb = np.asarray(time_balanced, dtype=R) * 0                # never incremented in QuadPole.step
    return nxt, reward, truncated, steps_n, tb


def quadpole_reset(alpha, beta, dtype=np.float64):
    """quadrotor_env.py:530-576.  alpha, beta ~ U(-1,1) -> q_p = norm(q_y (x) q_x)."""
    alpha = np.asarray(alpha, dtype=dtype)
    beta = np.asarray(beta, dtype=dtype)
    n = len(alpha)
    z = np.zeros(n, dtype=dtype)
    qx = np.stack([np.cos(alpha / 2), np.sin(alpha / 2), z, z], axis=1)
    qy = np.stack([np.cos(beta / 2), z, np.sin(beta / 2), z], axis=1)
    qp = quat_mult(qy, qx)
    qp = qp / np.sqrt((qp * qp).sum(axis=1))[:, None]
    st = np.zeros((n, 20), dtype=dtype)
    st[:, 6] = 1
    st[:, 13:17] = qp
    return st


# ---------------------------------------------------------------------------
# Quadrotor._dynamics (12-state Euler-angle body; the class itself is a stub)
# environments/quadrotor_env.py:113-169
# ---------------------------------------------------------------------------
QUADROTOR = dict(mass=1.0, arm_length=0.2, Ixx=0.005, Iyy=0.005, Izz=0.006,
                 torque_constant=0.017, gravity=9.80665, timestep=0.05)   # :9-16


def quadrotor_dynamics(state, control, dtype=np.float64):
    """Explicit-Euler step of the 12-state quadrotor; control is used raw (no wrap).
    quadrotor_env.py:128-169.  The reference's R[2][1] uses -sin(phi)*sin(psi)
    (typo at :144); only column 3 of R is used so it is inert."""
    R_ = dtype
    P = QUADROTOR
    s = np.asarray(state, dtype=R_)
    c = np.asarray(control, dtype=R_).reshape(len(s), 4)
    x, y, z, xd, yd, zd, phi, th, psi, p, q, r = (s[:, i] for i in range(12))
    u1, u2, u3, u4 = (c[:, i] for i in range(4))
    ut = ((u1 + u2) + u3) + u4
    m = R_(P["mass"])
    ax = R_(1) / m * ((-np.sin(th)) * ut)
    ay = R_(1) / m * ((np.sin(phi) * np.cos(th)) * ut)
    az = R_(1) / m * ((np.cos(phi) * np.cos(th)) * ut + (-m * R_(P["gravity"])))
    phid = (p + np.sin(phi) * np.tan(th) * q) + np.cos(phi) * np.tan(th) * r
    thd = np.cos(phi) * q + (-np.sin(phi)) * r
    psid = np.sin(phi) / np.cos(th) * q + np.cos(phi) / np.cos(th) * r
    s22 = R_(np.sqrt(2) / 2)
    al = R_(P["arm_length"])
    pd = (s22 * (u1 + u3 - u2 - u4) * al - R_(P["Izz"] - P["Iyy"]) * q * r) / R_(P["Ixx"])
    qd = (s22 * (u3 + u4 - u1 - u2) * al - R_(P["Izz"] - P["Ixx"]) * p * r) / R_(P["Iyy"])
    rd = (R_(P["torque_constant"]) * (u1 + u4 - u2 - u3)) / R_(P["Izz"])
    rates = np.stack([xd, yd, zd, ax, ay, az, phid, thd, psid, pd, qd, rd], axis=1)
    return s + rates * R_(P["timestep"])


# ---------------------------------------------------------------------------
# Pendulum  (environments/pendulum_env.py) -- the one env whose episode can TERMINATE:
# `terminated = _time_balanced > 5` (:151).  It returns (obs, reward, truncated, terminated,
# info) in that order (:158), which the worker unpacks as (.., terminated, truncated, ..)
# (rollout_worker.py:56) -- harmless, `done` is their disjunction.
# ---------------------------------------------------------------------------
PENDULUM_DEFAULTS = dict(mass=1.0, length=0.5, gravity=9.80665, timestep=0.05)   # pendulum_env.py:12-15
PENDULUM_BALANCE_TIME = 5.0                                                         # :151


def pendulum_balance_term_steps(timestep: float = 0.05, limit: float = PENDULUM_BALANCE_TIME) -> int:
    """Smallest number k of CONSECUTIVE balanced steps after which `_time_balanced > 5` holds
    (`_time_balanced = _time_balanced + timestep if cos_theta <= -0.99 else 0`, :135, fp64-accumulated)."""
    tb, k = 0, 0
    while True:
        k += 1
        tb = tb + timestep
        if tb > limit:
            return k


def pendulum_step(state, action, steps, time_balanced, *, max_steps=200, mass=1.0, length=0.5,
                  gravity=9.80665, timestep=0.05, dtype=np.float64):
    """One Pendulum.step for N envs.  pendulum_env.py:45-46 (_wrap_action), :48-74 (_dynamics), :124-158 (step).

    Returns (next_state (N,3), reward, truncated (time only, :150), steps_next, time_balanced_next);
    the episode also ends when time_balanced_next > 5 (:151, ENV_SPECS['Pendulum']['balance_terminates'])."""
    R = dtype
    state = np.asarray(state, dtype=R)
    a32 = np.clip(np.asarray(action, dtype=F32).reshape(len(state), 1), F32(-1), F32(1))   # :46 float32
    u = a32[:, 0].astype(R)               # float32 array - np.float64 scalar promotes to float64 (:63)
    s, c, thd = (state[:, i] for i in range(3))                                  # :56
    thd = np.clip(thd, R(-10), R(10))                                            # :57
    theta = np.arctan2(s, c)                                                     # :59
    g_term = R(mass * gravity * length) * np.sin(theta)                          # python-float product first (:61)
    alpha = R(1 / (mass * length ** 2)) * (u - g_term)                           # :61
    dt = R(timestep)
    thd_n = thd + alpha * dt                                                     # :63
    theta_n = theta + thd_n * dt                                                 # :64
    s_n, c_n = np.sin(theta_n), np.cos(theta_n)
    nxt = np.stack([s_n, c_n, thd_n], axis=1)                                    # :68-72

    steps_n = np.asarray(steps).astype(np.int64) + 1                             # :133
    balanced = c_n <= R(-0.99)                                                   # :135
    tb = np.where(balanced, np.asarray(time_balanced, dtype=R) + dt, R(0))
    energy32 = F32(-0.001) * (a32 ** 2).sum(axis=1, dtype=F32)                   # float32 (weak python scalar), :145
    e1 = R(-10) * np.abs(R(-1) - c_n) ** R(0.5)                                  # :143
    e2 = R(-0.1) * thd_n ** 2                                                    # :144
    reward = dt * ((e1 + e2) + energy32.astype(R))                               # np.sum of a 3-list, :142
    reward = np.where(tb > 0, reward + R(1), reward)                             # :148-149
    truncated = steps_n >= cartpole_time_trunc_step(max_steps, timestep)         # :150, float-accumulated time
    return nxt, reward, truncated, steps_n, tb


def pendulum_reset(theta0, dtype=np.float64):
    """pendulum_env.py:86-106: [sin t0, cos t0, 0]; t0 ~ U(pi-0.05, pi+0.05), or U(-pi, pi) with swingup=True."""
    theta0 = np.asarray(theta0, dtype=dtype)
    return np.stack([np.sin(theta0), np.cos(theta0), np.zeros_like(theta0)], axis=1)


# ---------------------------------------------------------------------------
# registry used by tests / the port worker
# ---------------------------------------------------------------------------
ENV_SPECS = {
    "CartPole": dict(obs_dim=5, act_dim=1, step=cartpole_step, timestep=0.02),
    "QuadPole2D": dict(obs_dim=10, act_dim=2, step=quadpole2d_step, timestep=0.02),
    "QuadPole": dict(obs_dim=20, act_dim=4, step=quadpole_step, timestep=0.02),
    "Pendulum": dict(obs_dim=3, act_dim=1, step=pendulum_step, timestep=0.05, balance_terminates=PENDULUM_BALANCE_TIME),
}


def sample_initial_states(env_name, n, rng, dtype=np.float64):
    """Initial-state distributions of the reference `reset()`s (one draw order
    per env as the reference: cartpole_env.py:103, quadrotor_env.py:543-544,:951)."""
    if env_name == "CartPole":
        return cartpole_reset(rng.uniform(-np.pi, np.pi, size=n), dtype)
    if env_name == "QuadPole2D":
        return quadpole2d_reset(rng.uniform(-np.pi, np.pi, size=n), dtype)
    if env_name == "QuadPole":
        ab = rng.uniform(-1.0, 1.0, size=(n, 2))
        return quadpole_reset(ab[:, 0], ab[:, 1], dtype)
    if env_name == "Pendulum":
        return pendulum_reset(rng.uniform(np.pi - 0.05, np.pi + 0.05, size=n), dtype)
    raise KeyError(env_name)
