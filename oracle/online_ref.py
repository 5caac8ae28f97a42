"""TEST INFRASTRUCTURE ONLY (never imported by climsim_amd/): CPU restatement of the reference's generic online wrapper.

  mlp_forward    online_testing/baseline_models/MLP_v2rh/training/mlp.py:51-67
  new_model      online_testing/model_postprocessing/v4_nn_wrapper.ipynb cell 5 (NewModel.preprocessing :12-26,
                 postprocessing :28-35, forward :37-41)

PARITY UNPINNED: the reference MLP derives from `modulus.Module` (not installed) and no trained weights ship, so this
restatement cannot be checked against the reference's own code or outputs; tests compare the HIP path with it only."""
import torch


def mlp_forward(x, weights, biases, output_prune, strato_lev_out):
    for w, b in zip(weights[:-1], biases[:-1]):
        x = torch.relu(x @ w.T + b)
    x = x @ weights[-1].T + biases[-1]
    if output_prune:
        x = x.clone()
        for o in (60, 120, 180, 240):
            x[:, o:o + strato_lev_out] = 0
    x = x.clone()
    x[:, -8:] = torch.relu(x[:, -8:])
    return x


def new_model(x, weights, biases, input_sub, input_div, out_scale, lbd_qc, lbd_qi, output_prune=True, strato_lev_out=12):
    x = x.clone()
    x[:, 120:180] = 1 - torch.exp(-x[:, 120:180] * lbd_qc)
    x[:, 180:240] = 1 - torch.exp(-x[:, 180:240] * lbd_qi)
    x = (x - input_sub) / input_div
    x = torch.where(torch.isnan(x), torch.zeros((), dtype=x.dtype), x)
    x = torch.where(torch.isinf(x), torch.zeros((), dtype=x.dtype), x)
    x[:, 120:135] = 0
    x[:, 180:195] = 0
    x[:, 60:120] = torch.clamp(x[:, 60:120], 0, 1.2)
    y = mlp_forward(x, weights, biases, output_prune, strato_lev_out)
    for a, b in ((60, 75), (120, 148), (180, 195), (240, 255), (300, 315)):
        y[:, a:b] = 0
    return y / out_scale
