"""TEST INFRASTRUCTURE ONLY (never imported by climsim_amd/): CPU restatement of the reference's physRNN "Hidden" model
(SURVEY section 8f #1), the architecture of the `rnn/saved_models/physRNN-Hidden_*_script_cpu.pt` artefacts:

  forward              rnn/models/models_phys.py:1586-1823 (as specialised in the artefact: GRU 128/128, nx = 21 (+ pressure),
                       ilev_crm = 10, mp_ncol = 16, nh_mem = 15 + 1 stored-water channel, no radiation scheme)
  microphysics_decode  rnn/models/models_phys.py:414-748
  pressure layers      rnn/layers.py:117-168 (LayerPressure, LayerPressureThickness, level pressure)

PINNED: tests/golden/make_golden_physrnn.py runs the shipped TorchScript artefact itself (torch.jit.load, CPU) on seeded
inputs, reproduces its internal `randn` draw for rnn2's initial state and stores inputs + outputs; this restatement is
checked against those outputs in tests/test_physrnn.py (CPU), and the HIP path against both (GPU).

P: dict of float tensors under the artefact's state_dict names.  hx2: the (B, nh) draw the artefact makes internally."""
import torch
import torch.nn.functional as F

CP, G, LV, LS, ONE_OVER_G = 1004.64, 9.80665, 2510400.0, 2844000.0, 0.1019716213

_JITTER = None      # test-only (tests/test_physrnn_frozen.py): a torch.Generator -> the sub-grid tendencies are re-rounded in their last bit


def _jit(t):
    if _JITTER is None:
        return t
    return t * (1.0 + 1.2e-7 * (2.0 * torch.rand(t.shape, generator=_JITTER, dtype=t.dtype) - 1.0))


def _gru(x, h0, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU, batch_first, one layer: gates r, z, n."""
    B, T, _ = x.shape
    nh = h0.shape[1]
    h, out = h0, []
    for t in range(T):
        gi = x[:, t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        r = torch.sigmoid(gi[:, :nh] + gh[:, :nh])
        z = torch.sigmoid(gi[:, nh:2 * nh] + gh[:, nh:2 * nh])
        n = torch.tanh(gi[:, 2 * nh:] + r * gh[:, 2 * nh:])
        h = (1 - z) * n + z * h
        out.append(h)
    return torch.stack(out, 1), h


def _lin(P, name, x):
    return x @ P[name + ".weight"].T + P[name + ".bias"]


def forward(P, inputs_main, inputs_aux, rnn_mem, inputs_denorm, hx2, ilev_crm=10, mp_ncol=16, nh_mem0=15, taps=None):
    """-> out_new (B, 60, 5), out_sfc (B, 8), rnn_mem (B, 50, 16)"""
    B, nlev, _ = inputs_main.shape
    hyam, hybm, hyai, hybi = (P[k].reshape(1, -1, 1) for k in ("hyam", "hybm", "hyai", "hybi"))
    P_old = rnn_mem[:, -1, -1]
    sp = inputs_aux[:, 0:1].unsqueeze(1) * P["xdiv_sca"][0:1] + P["xmean_sca"][0:1]          # (B,1,1)
    play = hyam * 100000.0 + sp * hybm                                                        # (B,60,1)
    pres = torch.sqrt(play) / 314.0
    delta_plev = sp * (hybi[:, 1:] - hybi[:, :-1]) + (hyai[:, 1:] - hyai[:, :-1]) * 100000.0  # (B,60,1)
    x = torch.tanh(_lin(P, "mlp_initial", torch.cat([inputs_main, pres], 2)))
    mem0 = torch.cat([x.new_zeros(B, ilev_crm, nh_mem0), rnn_mem[:, :, :nh_mem0]], 1)
    rnn1_in = torch.flip(torch.cat([x, mem0], 2), [1])
    hx = torch.tanh(_lin(P, "mlp_surface1", inputs_aux))
    rnn1out, _ = _gru(rnn1_in, hx, P["rnn1.weight_ih_l0"], P["rnn1.weight_hh_l0"], P["rnn1.bias_ih_l0"], P["rnn1.bias_hh_l0"])
    rnn1out = torch.flip(rnn1out, [1])
    rnn2out, last_h = _gru(rnn1out, hx2, P["rnn2.weight_ih_l0"], P["rnn2.weight_hh_l0"], P["rnn2.bias_ih_l0"], P["rnn2.bias_hh_l0"])
    if taps is not None:
        taps["rnn1out"], taps["rnn2out"] = rnn1out, rnn2out
    mem_new = _lin(P, "mlp_latent", rnn2out)[:, ilev_crm:]                                    # (B,50,15)
    r2 = rnn2out[:, ilev_crm:]                                                                # (B,50,nh)
    out_sfc_rad = torch.relu(_lin(P, "mlp_surface_output_rad", last_h))                       # (B,6)
    dT_rad = _lin(P, "mlp_output_rad", rnn2out)                                               # (B,60,1)
    out = _lin(P, "mlp_output", mem_new)                                                      # (B,50,5)

    dec = microphysics_decode(P, out, mem_new, r2, last_h, inputs_denorm, delta_plev, play, P_old, ilev_crm, mp_ncol)
    out_new, precc, precsc, mem_out = dec["out_new"], dec["precc"], dec["precsc"], dec["mem_out"]
    out_new[:, :, 0:1] = out_new[:, :, 0:1] + dT_rad
    out_sfc = torch.cat([out_sfc_rad[:, 0:2], precsc, precc, out_sfc_rad[:, 2:]], 1)
    return out_new, out_sfc, mem_out


def microphysics_decode(P, out, mem_new, r2, last_h, inputs_denorm, delta_plev, play, P_old, ilev_crm, mp_ncol, copy_dT=True,
                        clear_sky=False, nx21=False, grid_T=None):
    """rnn/models/models_phys.py:414-748.  copy_dT: the nx = 21 graphs also pass the decoder's raw column-0 output of the
    levels below ilev_crm + 2 through to out_new; the radiation graphs (oracle/physrnn_rad_ref.py) do not.
    clear_sky (the physRNN_physRad-* graphs, `use_clear_sky_region`, no sub-grid temperature): region 0 holds no condensate
    (mlp_qn_crm / mlp_evap_cond_vapor_crm have mp_ncol - 1 outputs, a zero is prepended), temperature and eddy heat flux are
    those of the grid column (mlp_eddy_diff has one output, no mlp_t_crm), and the latent heating is formed from the
    area-summed condensation / evaporation.
    nx21 (the generation of the frozen `*_wrapped` exports, oracle/physrnn_frozen_ref.py): the eddy heat flux is defined at layer
    tops like the moisture fluxes (zero at the surface, where the Hidden graphs take -relu of it), and the cloud liquid fraction
    is PER REGION -- the ramp on the region's own temperature after the flux divergence, or the learned `mlp_liq_frac_crm` head
    -- both in the latent heating and, returned as `liq_frac`, in the radiation scheme's cloud water paths.
    -> dict with out_new, precc, precsc, mem_out and the updated sub-column state (T_crm, qv_crm, qn_crm, area_frac)."""
    B, nlev = inputs_denorm.shape[0], inputs_denorm.shape[1]
    # grid_T: no sub-grid temperature (grid temperature in the eddy flux and the liquid ramp, latent heating from the area-summed
    # rates); it comes with the clear-sky region in the physRad graphs, and WITHOUT it in the frozen exports num45826 / num74834
    grid_T = clear_sky if grid_T is None else grid_T
    x = out
    ys = P["yscale_lev"][ilev_crm:]                                                           # (50,5)
    out_new = x.new_zeros(B, nlev, 5)
    pres_diff = delta_plev[:, ilev_crm:]                                                      # (B,50,1)
    out_new[:, ilev_crm + 2:, -2:] = out[:, 2:, -2:]
    if copy_dT:
        out_new[:, ilev_crm + 2:, 0] = out[:, 2:, 0]
    xd = inputs_denorm[:, ilev_crm:]
    qv_gcm, T_gcm, qliq_gcm, qice_gcm = xd[:, :, -1:], xd[:, :, 0:1], xd[:, :, 2:3], xd[:, :, 3:4]
    qn_gcm = qliq_gcm + qice_gcm
    qv_crm = F.softplus(_lin(P, "mlp_qv_crm", r2))
    qn_crm = F.softplus(_lin(P, "mlp_qn_crm", r2))
    if clear_sky:
        qn_crm = torch.cat([qn_crm.new_zeros(B, nlev - ilev_crm, 1), qn_crm], 2)
    area_frac = torch.softmax(_lin(P, "mlp_subgrid_area_frac", r2), 2)

    def rescale(q, gcm):
        mean = (q * area_frac).sum(-1, keepdim=True)
        return q * torch.where(mean == 0, torch.ones_like(mean), gcm / mean)
    qv_crm, qn_crm = rescale(qv_crm, qv_gcm), rescale(qn_crm, qn_gcm)
    if grid_T:
        T_crm = T_gcm                                                                         # (B,50,1)
    else:
        deltaT = _lin(P, "mlp_t_crm", r2)
        T_crm = T_gcm + (deltaT - (deltaT * area_frac).sum(-1, keepdim=True))
    flux1 = _lin(P, "mlp_massflux", r2)
    eddy = _lin(P, "mlp_eddy_diff", r2)
    zer = x.new_zeros(B, 1, mp_ncol)
    play_diff = play[:, ilev_crm:] - play[:, ilev_crm - 1:-1]
    fH = (eddy * (CP / G)) * T_crm * play_diff
    if nx21:
        fH = torch.cat([zer[:, :, :fH.shape[2]], fH[:, :-1], zer[:, :, :fH.shape[2]]], 1)
    else:
        fH = torch.cat([fH[:, :-1], -torch.relu(fH[:, -1:])], 1)
        fH = torch.cat([zer[:, :, :fH.shape[2]], fH], 1)
    flux_t_dp = (fH[:, 1:] - fH[:, :-1]) / pres_diff * (-G / CP)
    f_qv = flux1 * 300000.0 * qv_crm
    f_qn = flux1 * 300000.0 * qn_crm
    qice_crm = rescale(F.softplus(_lin(P, "mlp_qice_crm", r2)), qice_gcm)
    sed = torch.relu(_lin(P, "mlp_sed_qn_crm", r2)) * G * qice_crm * ys[:, 2].reshape(1, -1, 1)
    sedimentation = (area_frac[:, -1] * sed[:, -1]).sum(1)
    sed = torch.cat([zer, sed], 1)
    sed_qn_dp = (sed[:, 1:] - sed[:, :-1]) / pres_diff * (-G)

    def div_flux(f):
        f = torch.cat([zer, f[:, :-1], zer], 1)
        return (f[:, 1:] - f[:, :-1]) / pres_diff * (-G)
    flux_qv_dp, flux_qn_dp = div_flux(f_qv), div_flux(f_qn)
    evap_prec = torch.relu(_lin(P, "mlp_evap_prec_crm", r2)) + 1e-6
    cond = _lin(P, "mlp_evap_cond_vapor_crm", r2)
    if clear_sky:
        cond = torch.cat([cond.new_zeros(B, nlev - ilev_crm, 1), cond], 2)
    P_vert = torch.softmax(out[:, :, 2], 1) * P_old.unsqueeze(1)                              # (B,50)
    evap_prec = evap_prec * P_vert.unsqueeze(2)
    alpha = torch.relu(_lin(P, "mlp_mp_aa_crm", r2))
    ys1, ys2, ys0 = ys[:, 1:2], ys[:, 2:3], ys[:, 0:1]
    dqn_aa = alpha * qn_crm * ys[:, 2].reshape(1, -1, 1)
    cond = torch.maximum(cond, -(ys2 * qn_crm / 1200) - flux_qn_dp + dqn_aa - sed_qn_dp)
    evap_prec = torch.maximum(evap_prec, -(ys1 * qv_crm / 1200) - flux_qv_dp + cond)
    dqn_aa = torch.maximum(dqn_aa, flux_qn_dp + cond + sed_qn_dp - ys2 * (-qn_crm + 0.0006) / 1200)
    # (the clamps above make q + dq * 1200 / yscale EXACTLY zero where they bind; in float32 a residue of either sign is left, relu keeps
    #  the positive ones, and the radiation scheme takes the FOURTH ROOT of the vapour residue: _jit lets a test realise that rounding)
    dqv_crm = _jit(flux_qv_dp - cond + evap_prec)
    dqn_crm = _jit(flux_qn_dp + cond - dqn_aa + sed_qn_dp)
    if grid_T:
        temp = T_gcm.squeeze(2) + (flux_t_dp.squeeze(2) / ys[:, 0]) * 1200
        liq = F.hardtanh((temp - 253.16) * 0.05, 0.0, 1.0).unsqueeze(2)
        cond_s, evap_s = (area_frac * cond).sum(2, keepdim=True), (area_frac * evap_prec).sum(2, keepdim=True)
        net_cond = ((liq * LV + (1 - liq) * LS) * cond_s - evap_s * LV) * (1 / CP)
    elif nx21:
        if "mlp_liq_frac_crm.weight" in P:
            liq = torch.sigmoid(_lin(P, "mlp_liq_frac_crm", r2))
        else:
            liq = F.hardtanh((T_crm + (flux_t_dp / ys0) * 1200 - 253.16) * 0.05, 0.0, 1.0)                  # (B,50,nreg)
        net_cond = ((liq * LV + (1 - liq) * LS) * cond - evap_prec * LV) * 0.00099538143016403898
    else:
        temp = T_gcm.squeeze(2) + ((area_frac * flux_t_dp).sum(2) / ys[:, 0]) * 1200
        liq = F.hardtanh((temp - 253.16) * 0.05, 0.0, 1.0).unsqueeze(2)
        net_cond = ((liq * LV + (1 - liq) * LS) * cond - evap_prec * LV) * (1 / CP)
    dT_crm = flux_t_dp + net_cond / ys1 * ys0
    out_new[:, ilev_crm:, 0] = out_new[:, ilev_crm:, 0] + (area_frac * dT_crm).sum(2)
    out_new[:, ilev_crm:, 1] = (area_frac * dqv_crm).sum(2)
    out_new[:, ilev_crm:, 2] = (area_frac * dqn_crm).sum(2)
    d_prec = (area_frac * (dqn_aa - evap_prec)).sum(2)
    water_new = P_old + (pres_diff.squeeze(2) * ONE_OVER_G * d_prec).sum(1)
    water_new = torch.relu(water_new)
    rel = torch.sigmoid(_lin(P, "mlp_precip_release", last_h)).squeeze(1)
    released = rel * water_new
    stored = water_new * (1 - rel)
    Tsfc = inputs_denorm[:, -1, 0]
    Pmax = P["yscale_sca"][3] * 1000 * 5.58e-18 * torch.exp(Tsfc * 0.077)
    excess = torch.relu(stored - Pmax)
    stored = stored - excess
    mem_out = torch.cat([mem_new, stored.reshape(B, 1, 1).expand(B, nlev - ilev_crm, 1)], 2)
    precc = ((sedimentation + released + excess) / 1000).unsqueeze(1)
    snowfrac = F.hardtanh((-inputs_denorm[:, -1, 0:1] + 283.3) / 14.6, 0.0, 1.0)
    precsc = snowfrac * precc
    ys_ = lambda k: ys[:, k:k + 1]
    return dict(out_new=out_new, precc=precc, precsc=precsc, mem_out=mem_out, area_frac=area_frac, liq_frac=liq,
                T_crm=torch.relu(T_crm + dT_crm * 1200 / ys_(0)), qv_crm=torch.relu(qv_crm + dqv_crm * 1200 / ys_(1)),
                qn_crm=torch.relu(qn_crm + dqn_crm * 1200 / ys_(2)), qn_crm_old=qn_crm)
