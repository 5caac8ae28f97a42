"""TEST INFRASTRUCTURE ONLY (never imported by climsim_amd/): numpy restatement of the evaluation scores of
climsim_utils/data_utils.py (calc_MAE :1843-1857, calc_RMSE :1859-1874, calc_R2 :1876-1892, calc_bias :1894-1908,
calc_CRPS :1910-1935).  data_utils.py itself cannot be imported here (xarray / netCDF4 / h5py at its top), so these
follow its text statement by statement, evaluated in float64; PARITY UNPINNED by reference outputs."""
import numpy as np


def calc_MAE(pred, target, avg_grid=True):
    mae = np.abs(pred - target).mean(axis=0)
    return mae.mean(axis=0) if avg_grid else mae


def calc_RMSE(pred, target, avg_grid=True):
    rmse = np.sqrt(((pred - target) ** 2).mean(axis=0))
    return rmse.mean(axis=0) if avg_grid else rmse


def calc_R2(pred, target, avg_grid=True):
    sq_diff = (pred - target) ** 2
    tss_time = (target - target.mean(axis=0)[np.newaxis, ...]) ** 2
    r2 = 1 - sq_diff.sum(axis=0) / tss_time.sum(axis=0)
    return r2.mean(axis=0) if avg_grid else r2


def calc_bias(pred, target, avg_grid=True):
    bias = pred.mean(axis=0) - target.mean(axis=0)
    return bias.mean(axis=0) if avg_grid else bias


def calc_CRPS(samplepreds, target, avg_grid=True):
    num_crps = samplepreds.shape[-1]
    mae = np.mean(np.abs(samplepreds - target[..., np.newaxis]), axis=(0, -1))
    samplepreds = np.sort(samplepreds, axis=-1)
    diff = samplepreds[..., 1:] - samplepreds[..., :-1]
    count = np.arange(1, num_crps) * np.arange(num_crps - 1, 0, -1)
    spread = (diff * count).sum(axis=-1).mean(axis=0)
    crps = mae - spread / (num_crps * (num_crps - 1))
    return crps.mean(axis=0) if avg_grid else crps
