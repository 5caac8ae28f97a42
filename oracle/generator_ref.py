"""numpy restatement of generator_xy.__getitem__ (rnn/utils.py:2238-2371) -- TEST INFRASTRUCTURE ONLY.

Follows the source line by line with the non-numba branches (the numba kernels :1803-1868 restate the same arithmetic).  PINNED
(round 3) by outputs of the reference class itself: tests/golden/make_golden_generator.py imports rnn/utils.py in the build
container (inert placeholders for numba / h5py, in-memory datasets) and stores what generator_xy.__getitem__, eliq, eice and
relative_to_specific_humidity_climsim return; tests/test_generator.py compares this restatement with those fixtures for every
variant (mp_mode 0 / 1 / -1 / -2 targets, float64 RH -> q, v4_to_v5_inputs, re-normalisation, previous-step inputs)."""
import numpy as np


def eliq(T):
    a = np.array([-0.976195544e-15, -0.952447341e-13, 0.640689451e-10, 0.206739458e-7, 0.302950461e-5, 0.264847430e-3,
                  0.142986287e-1, 0.443987641, 6.11239921])
    return 100 * np.polyval(a, np.maximum(-80, T - 273.16))


def eice(T):
    a = np.array([0.252751365e-14, 0.146898966e-11, 0.385852041e-9, 0.602588177e-7, 0.615021634e-5, 0.420895665e-3,
                  0.188439774e-1, 0.503160820, 6.11147274])
    c = np.array([273.15, 185, -100, 0.00763685, 0.000151069, 7.48215e-07])
    T0 = 273.16
    return (T > c[0]) * eliq(T) + (T <= c[0]) * (T > c[1]) * 100 * np.polyval(a, T - T0) + \
        (T <= c[1]) * 100 * (c[3] + np.maximum(c[2], T - T0) * (c[4] + np.maximum(c[2], T - T0) * c[5]))


def rh_to_q(rh, temp, pressure):
    T0, T00 = 273.16, 253.16
    omega = (temp - T00) / (T0 - T00)
    omega = np.maximum(0, np.minimum(1, omega))
    esat = omega * eliq(temp) + (1 - omega) * eice(temp)
    return rh * ((287 * esat) / (461 * pressure))


def getitem(x_lev_b, x_sfc_b, y_lev_b, y_sfc_b, *, xcoeffs=None, ycoeffs=None, xcoeffs_ref=None, ycoeffs_ref=None,
            lbd_qc=None, lbd_qi=None, lbd_qn=None, v4_to_v5_inputs=False, cld_inp_transformation="exp",
            remove_past_sfc_inputs=False, qinput_prune=False, rh_input_to_q=False, include_q_input=False, rh_prune=False,
            output_prune=False, mp_mode=0, hybm=None, hyam=None, snowhice_fix=True):
    x_lev_b, x_sfc_b = np.array(x_lev_b, np.float32), np.array(x_sfc_b, np.float32)
    y_lev_b, y_sfc_b = np.array(y_lev_b, np.float32), np.array(y_sfc_b, np.float32)
    if xcoeffs_ref is not None:
        x_lev_b = x_lev_b * xcoeffs_ref[0][1] + xcoeffs_ref[0][0]
        x_sfc_b = x_sfc_b * xcoeffs_ref[1][1] + xcoeffs_ref[1][0]
    if remove_past_sfc_inputs:
        x_sfc_b = np.delete(x_sfc_b, (17, 18, 19, 20, 21), axis=1)
    if snowhice_fix:
        x_sfc_b[x_sfc_b > 1.0e10] = -1.0
    if rh_prune:
        x_lev_b[:, :, 1] = np.clip(x_lev_b[:, :, 1], 0.0, 1.2)
    if rh_input_to_q:
        rh, temp, sp = x_lev_b[:, :, 1], x_lev_b[:, :, 0], x_sfc_b[:, 0:1]
        pressure = sp * hybm.reshape(1, -1) + 100000.0 * hyam.reshape(1, -1)
        qwv = np.float32(rh_to_q(rh, temp, pressure))
        if include_q_input:
            x_lev_b = np.concatenate((x_lev_b, qwv.reshape(-1, x_lev_b.shape[1], 1)), axis=2)
        else:
            x_lev_b[:, :, 1] = qwv
    x_lev_b_denorm = np.copy(x_lev_b)
    if v4_to_v5_inputs:
        lf = np.clip((x_lev_b[:, :, 0] - 253.16) * 0.05, 0.0, 1.0)
        qn = x_lev_b[:, :, 2] + x_lev_b[:, :, 3]
        if qinput_prune:
            qn[:, 0:15] = 0.0
        x_lev_b[:, :, 2] = qn
        x_lev_b[:, :, 3] = lf
        if cld_inp_transformation == "exp":
            x_lev_b[:, :, 2] = 1 - np.exp(-x_lev_b[:, :, 2] * lbd_qn)
        elif cld_inp_transformation == "sqrt":
            x_lev_b[:, :, 2] = np.sqrt(np.sqrt(x_lev_b[:, :, 2]))
    else:
        if cld_inp_transformation == "exp":
            x_lev_b[:, :, 2] = 1 - np.exp(-x_lev_b[:, :, 2] * lbd_qc)
            x_lev_b[:, :, 3] = 1 - np.exp(-x_lev_b[:, :, 3] * lbd_qi)
        elif cld_inp_transformation == "sqrt":
            x_lev_b[:, :, 2] = np.sqrt(np.sqrt(x_lev_b[:, :, 2]))
            x_lev_b[:, :, 3] = np.sqrt(np.sqrt(x_lev_b[:, :, 3]))
        if qinput_prune:
            x_lev_b[:, 0:15, 2:3] = 0.0
    if xcoeffs is not None:
        x_lev_b = (x_lev_b - xcoeffs[0][0]) / xcoeffs[0][1]
        x_sfc_b = (x_sfc_b - xcoeffs[1][0]) / xcoeffs[1][1]
        if rh_input_to_q:
            col = -1 if include_q_input else 1
            q = x_lev_b[:, :, col]
            q[q < 0.0] = 0.0
            x_lev_b[:, :, col] = q
    x_lev_b[np.isnan(x_lev_b)] = 0
    if ycoeffs_ref is not None:
        y_lev_b = y_lev_b / ycoeffs_ref[0]
        y_sfc_b = y_sfc_b / ycoeffs_ref[1]
    y_lev_b_denorm, y_sfc_b_denorm = np.copy(y_lev_b), np.copy(y_sfc_b)
    if mp_mode > 0:
        y_lev_b[:, :, 2] = y_lev_b[:, :, 2] + y_lev_b[:, :, 3]
        y_lev_b = np.delete(y_lev_b, 3, axis=2)
    elif mp_mode < 0:
        T_before, qliq_before, qice_before = x_lev_b_denorm[:, :, 0], x_lev_b_denorm[:, :, 2], x_lev_b_denorm[:, :, 3]
        qn_before = qliq_before + qice_before
        dqliq, dqice = y_lev_b[:, :, 2], y_lev_b[:, :, 3]
        dqn = dqliq + dqice
        qn_new = qn_before + dqn * 1200
        qn_new[qn_new < 0.0] = 0.0
        qliq_new = qliq_before + dqliq * 1200
        T_new = T_before + y_lev_b[:, :, 0] * 1200
        inds = (qn_new > 1e-20) & (dqn > 1e-20)
        liq_frac = (T_new - 253.16) / 20.0
        liq_frac[liq_frac < 0.0] = 0.0
        liq_frac[liq_frac > 1.0] = 1.0
        liq_frac[inds] = qliq_new[inds] / qn_new[inds]
        liq_frac[liq_frac < 0.0] = 0.0
        liq_frac[liq_frac > 1.0] = 1.0
        if mp_mode == -2:
            qv_before = x_lev_b_denorm[:, :, -1]
            dqv = y_lev_b[:, :, 1]
            dqtot = dqv + dqn
            qv_new = qv_before + dqv * 1200
            qv_new[qv_new < 0.0] = 0.0
            qtot_new = qv_new + qn_new
            inds_pos = qtot_new > 0
            tot_cld_frac = np.zeros(dqn.shape, dtype=np.float32)
            tot_cld_frac[inds_pos] = qn_new[inds_pos] / qtot_new[inds_pos]
            tot_cld_frac = np.sqrt(np.sqrt(tot_cld_frac))
            y_lev_b[:, :, 1] = dqtot
            y_lev_b[:, :, 2] = tot_cld_frac
        else:
            y_lev_b[:, :, 2] = dqn
        y_lev_b[:, :, 3] = liq_frac
    y_lev_b = y_lev_b * ycoeffs[0]
    if output_prune:
        y_lev_b[:, 0:12, 1:] = 0.0
    y_sfc_b = y_sfc_b * ycoeffs[1]
    return x_lev_b, x_sfc_b, y_lev_b, y_sfc_b, x_lev_b_denorm, y_lev_b_denorm, y_sfc_b_denorm
