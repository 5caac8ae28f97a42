#!/usr/bin/env python3
"""Constant extractor for the FROZEN physRNN exports (rnn/saved_models/*_wrapped.pt) -- the loader of the reference's deployed artefact
format: `load_export(path)` -> (state_dict, cfg) for `climsim_amd.physrnn.physical_RNN_wrapped` (also `physical_RNN_wrapped.from_export`).

`torch.jit.freeze` inlined rnn/utils.py::model_wrapper and rnn/models/models_phys.py::physical_RNN_autoreg into one `forward` and
turned every parameter and buffer into an anonymous graph constant `CONSTANTS.cN` (numbered in order of first use).  The serialised
code keeps the reference's own variable names (qv_crm, area_frac, alpha, sed, tau_lw, ...), so a constant's role follows from the
NAMED expression it feeds.  This module
  1. parses `module.code` into single assignments and expands the `_N` temporaries into the named variable that consumes them,
  2. matches the expanded expressions against the reference's statements (models_phys.py:414-748 decoder heads, :816-1270 optics,
     :1272-1490 solver, :1586-1823 forward; rnn/utils.py:134-295 wrapper) and names every constant with the state_dict key it had
     before freezing (weights transposed back to (out, in)),
  3. reports the switches in which the twenty code variants differ (read off the code text), and any constant it could not name.
Nothing of the reference's source is stored: the output is a dict of tensors + a dict of flags."""
import re

import torch

C = r"CONSTANTS\.c(\d+)"
LIN0 = r"torch\.add\(torch\.matmul\((\w+), " + C + r"\), " + C + r"\)"
# the `_gpu` exports were frozen with horizontally fused Linear layers: ONE matmul against the concatenated weights of all heads that
# share an input, sliced afterwards -- `torch.slice(torch.add(torch.matmul(x, cW), cB), -1, a, b)` is head columns a:b of (cW, cB)
LIN = r"(?:torch\.slice\()?" + LIN0 + r"(?:, -1, (\d+), (\d+)\))?"
NG_LIN = 5          # groups per LIN: input, weight, bias, slice begin, slice end


def parse(code):
    """-> (ordered list of (lhs, rhs), dict temp -> rhs).  Only top-level single-line assignments matter here."""
    stmts, temps = [], {}
    for line in code.splitlines():
        m = re.match(r"^\s+([\w, ]+?),? = (.*)$", line)
        if not m:
            continue
        lhs, rhs = m.group(1).strip(), m.group(2)
        stmts.append((lhs, rhs))
        if re.fullmatch(r"_\d+", lhs):
            temps[lhs] = rhs
    return stmts, temps


def expand(rhs, temps, depth=0):
    if depth > 12:
        return rhs
    return re.sub(r"\b_\d+\b", lambda m: expand(temps[m.group(0)], temps, depth + 1) if m.group(0) in temps else m.group(0), rhs)


class Extractor:
    def __init__(self, module):
        self.code, consts = module.code_with_constants
        self.cm = consts.const_mapping
        self.stmts, self.temps = parse(self.code)
        self.named = [(l, expand(r, self.temps)) for l, r in self.stmts if not re.fullmatch(r"_\d+", l)]
        # copy_ / index_put_ results are temporaries nobody reads: keep their expanded text too
        self.sinks = [expand(r, self.temps) for l, r in self.stmts if re.fullmatch(r"_\d+", l) and ("copy_(" in r or "index_put_" in r)]
        self.P, self.used = {}, set()

    def c(self, n):
        self.used.add(int(n))
        v = self.cm[f"c{int(n)}"]
        return v.detach().clone() if isinstance(v, torch.Tensor) else torch.tensor(v)

    def find(self, lhs_re, rhs_re, which=0, required=True):
        hits = []
        for l, r in self.named:
            if re.fullmatch(lhs_re, l):
                m = re.search(rhs_re, r)
                if m:
                    hits.append(m)
        if len(hits) <= which:
            if required:
                raise KeyError(f"no statement `{lhs_re} = ... {rhs_re}`")
            return None
        return hits[which]

    def linear(self, name, lhs_re, wrap=r"{LIN}", which=0, required=True, lin_index=0):
        """name.weight / name.bias from the `lin_index`-th matmul+add of the statement."""
        pat = wrap.replace("{LIN}", LIN)
        m = self.find(lhs_re, pat, which, required)
        if m is None:
            return False
        g = m.groups()
        self.P[name + ".weight"], self.P[name + ".bias"] = self.lin_wb(g[NG_LIN * lin_index + 1:NG_LIN * lin_index + 5])
        return True

    def lin_wb(self, g):
        """(weight constant, bias constant, slice begin | None, slice end | None) -> weight (out, in), bias (out)"""
        w, b = self.c(g[0]), self.c(g[1]).reshape(-1)
        if g[2] is not None:
            w, b = w[:, int(g[2]):int(g[3])], b[int(g[2]):int(g[3])]
        return w.t().contiguous(), b.contiguous()

    def const(self, name, lhs_re, rhs_re, group=1, required=True, which=0):
        m = self.find(lhs_re, rhs_re, which, required)
        if m is None:
            return False
        self.P[name] = self.c(m.group(group))
        return True

    # ------------------------------------------------------------------------------------------------------------------
    def run(self):
        P, code = self.P, self.code
        F = {}
        # ---- recurrent layers ------------------------------------------------------------------------------------------
        for r in ("rnn1", "rnn2"):
            m = re.search(rf"self_original_model_{r}__flat_weights = \[{C}, {C}, {C}, {C}\]", code)
            for k, n in zip(("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"), m.groups()):
                P[f"{r}.{k}"] = self.c(n)
        F["nh"] = P["rnn2.weight_hh_l0"].shape[1]
        F["rnn3"] = "epss = torch.randn" in code
        if F["rnn3"]:      # MyStochasticGRULayer5 without bias (models_torch_kernels.py:834-891)
            self.const("rnn3.weight_encoder", r"predicted_distribution", rf"torch\.mm\(\w+, {C}\)")
            self.const("rnn3.weight_ih", r"x_results", rf"torch\.mm\(\w+, {C}\)")
            self.const("rnn3.weight_zh", r"z_results", rf"torch\.mm\(\w+, {C}\)")
        # ---- wrapper (rnn/utils.py:134-295) ---------------------------------------------------------------------------------
        m = self.find(r"pres", rf"torch\.add\({C}, torch\.mul\(sp, {C}\)\)")
        P["hyam"], P["hybm"] = self.c(m.group(1)) / 100000.0, self.c(m.group(2))
        self.const("snow_fill", r"x_sfc\d*", rf"torch\.where\(torch\.ge\(x_sfc\d*, 10000000000\), {C}, x_sfc\d*\)")
        lam = re.findall(rf"torch\.mul\(torch\.neg\(_\d+\), {C}\)", code)
        P["lbd_qc"], P["lbd_qi"] = self.c(lam[0]), self.c(lam[1])
        m = self.find(r"x_main\d+", rf"torch\.div\(torch\.sub\(x_main\d*, {C}\), {C}\)")
        P["xmean_lev"], P["xdiv_lev"] = self.c(m.group(1)), self.c(m.group(2))
        m = self.find(r"x_sfc\d+", rf"torch\.div\(torch\.sub\(x_sfc\d*, {C}\), {C}\)")
        P["xmean_sca"], P["xdiv_sca"] = self.c(m.group(1)), self.c(m.group(2))
        m = self.find(r"sp\d+", rf"torch\.add\(torch\.mul\(sp\d*, {C}\), {C}\)")
        P["sp_div"], P["sp_mean"] = self.c(m.group(1)), self.c(m.group(2))
        m = self.find(r"pres\d+", rf"torch\.add\({C}, torch\.mul\(sp\d+, {C}\)\)")
        P["hyam_m"], P["hybm_m"] = self.c(m.group(1)).reshape(-1) / 100000.0, self.c(m.group(2)).reshape(-1)
        m = self.find(r"delta_plev", rf"torch\.add\(torch\.mul\(sp\d+, {C}\), {C}\)")
        P["dhybi"], P["dhyai"] = self.c(m.group(1)).reshape(-1), self.c(m.group(2)).reshape(-1) / 100000.0
        m = self.find(r"plev", rf"torch\.add\(torch\.mul\(sp\d+, {C}\), {C}\)")
        P["hybi"], P["hyai"] = self.c(m.group(1)).reshape(-1), self.c(m.group(2)).reshape(-1) / 100000.0
        # ---- input MLPs, latent / output ----------------------------------------------------------------------------------
        self.linear("mlp_initial", r"inputs_main_crm\d*", r"torch\.tanh\({LIN}\)")
        self.linear("mlp_surface1", r"hx\d*", r"torch\.tanh\({LIN}\)")
        m = re.search(r"torch\.add\(torch\.matmul\(" + LIN + rf", {C}\), {C}\)", "\n".join(r for _, r in self.named) + "\n".join(self.sinks))
        g = m.groups()
        P["mlp_latent.weight"], P["mlp_latent.bias"] = self.lin_wb(g[1:5])
        P["mlp_output.weight"], P["mlp_output.bias"] = self.c(g[5]).t().contiguous(), self.c(g[6])
        # ---- decoder heads (models_phys.py:414-748) ------------------------------------------------------------------------
        self.linear("mlp_qv_crm", r"qv_crm", r"torch\.softplus\({LIN}, 1\., 20\.\)")
        if not self.linear("mlp_qn_crm", r"qn_crm", r"torch\.softplus\({LIN}, 1\., 20\.\)", required=False):
            self.linear("mlp_qn_crm", r"qn_crm\d*", r"torch\.softplus\({LIN}, 1\., 20\.\)")
        self.linear("mlp_subgrid_area_frac", r"area_frac", r"torch\.softmax\({LIN}, 2\)")
        F["pred_subgrid_temp"] = self.linear("mlp_t_crm", r"deltaT", r"torch\.sub\({LIN}", required=False)
        self.linear("mlp_massflux", r"flux_net_qv", r"torch\.mul\(torch\.mul\({LIN}, 300000\.\)")
        self.linear("mlp_eddy_diff", r"flux_net_H", r"{LIN}")
        F["ice_sedimentation"] = self.linear("mlp_qice_crm", r"qice_crm", r"torch\.softplus\({LIN}, 1\., 20\.\)", required=False)
        if F["ice_sedimentation"]:
            self.linear("mlp_sed_qn_crm", r"sed", r"torch\.relu\({LIN}\)")
        self.linear("mlp_evap_prec_crm", r"result\d*", r"torch\.relu\({LIN}\)")
        self.linear("mlp_evap_cond_vapor_crm", r"dq_cond_evap_vapor\d*", r"{LIN}")
        self.linear("mlp_mp_aa_crm", r"alpha", r"torch\.relu\({LIN}\)")
        self.linear("mlp_precip_release", r"precc_release_fraction", r"torch\.sigmoid\({LIN}\)")
        F["pred_subgrid_liq_frac"] = self.linear("mlp_liq_frac_crm", r"liq_frac_crm\d*", r"torch\.sigmoid\({LIN}\)", required=False)
        F["nreg"] = P["mlp_qv_crm.weight"].shape[0]
        F["clear_sky"] = P["mlp_qn_crm.weight"].shape[0] == F["nreg"] - 1
        if F["ice_sedimentation"]:
            self.const("ys_qn50", r"sed\d+", rf"torch\.mul\(torch\.mul\(torch\.mul\(sed, 9\.806\d*\), qice_crm\d*\), {C}\)")
        else:
            self.const("ys_qn50", r"dqn_aa", rf"torch\.mul\(torch\.mul\(alpha, qn_crm\d*\), {C}\)")
        self.const("ys_T50", r"temp\d*", rf"torch\.div\(flux_t_dp\w*, {C}\)", required=False) or \
            self.const("ys_T50", r"net_condensation_crm\d*", rf"\), {C}\)$")
        self.const("ys_T50_sq", r"temp\d*", rf"torch\.div\(torch\.squeeze\(flux_t_dp\w*\), {C}\)", required=False)     # (the grid-temperature decoder's copy)
        self.const("pmax_coef", r"Pmax", rf"torch\.mul\({C}, ")
        self.const("yscale_lev_3d", r"out_denorm", rf"torch\.div\(out_new, {C}\)", required=False)     # (absent where radiation reads the state before the step)
        # ---- LW gas optics + reductions (models_phys.py:816-1270, rnn/layers.py gasopt_mlp) -------------------------------
        m = self.find(r"x_gas\d+", rf"torch\.div\(torch\.sub\(x_gas, {C}\), {C}\)")
        P["gas_optics_model_lw.xmin"], P["gas_optics_model_lw.xdiv"] = self.c(m.group(1)), self.c(m.group(2))
        soft = r"torch\.div\({LIN}, torch\.add\(torch\.abs\("
        self.linear("gas_optics_model_lw.mlp1", r"x\d+", soft, which=0)
        self.linear("gas_optics_model_lw.mlp2", r"x\d+", soft, which=1)
        m = re.search(r"tau\d*, pfrac\d*,? = torch\.chunk\((_\d+), 2, -1\)", code)
        mm = re.search(LIN, expand(self.temps[m.group(1)], self.temps))
        P["gas_optics_model_lw.mlp3.weight"], P["gas_optics_model_lw.mlp3.bias"] = self.lin_wb(mm.groups()[1:5])
        m = self.find(r"tau\d+", rf"torch\.mul\(col_dry\d*, torch\.pow\(torch\.add\(torch\.mul\({C}, tau\d*\), {C}\), 8\)\)")
        P["gas_optics_model_lw.ystd"], P["gas_optics_model_lw.ymean"] = self.c(m.group(1)), self.c(m.group(2))
        self.linear("gas_optics_lw_reduce2", r"pfrac\d+", r"torch\.softmax\({LIN}, 2\)")
        self.linear("gas_optics_lw_reduce1", r"tau_lw\d*", r"torch\.mul\(torch\.softplus\({LIN}, 1\., 20\.\), 0\.01\)")
        F["cloud_optics_lw"] = self.linear("cloud_optics_lw", r"tau_lw_cld\d*", r"torch\.relu\({LIN}\)", required=False)
        # ---- SW gas optics --------------------------------------------------------------------------------------------------
        m = self.find(r"x_gas_1\d*", rf"torch\.div\(torch\.sub\(x_gas_1\d*, {C}\), {C}\)", required=False)
        F["sw_mlp"] = m is None
        if m is None:
            # earlier sub-generation: SW optical properties of every g-point from ONE two-layer MLP on (pressure, T, q_v, cloud water,
            # three gases, effective radii, the 15 new latent channels) -- models_phys.py's mlp_sw_optprops, as in the unfrozen num4050 family
            self.linear("mlp_sw_optprops1", r"sw_optprops", soft)
            self.linear("mlp_sw_optprops2", r"sw_optprops\d+", r"torch\.reshape\({LIN}, ")
            self.const("lbd_qn", r"qn_new", rf"torch\.add\(torch\.neg\(torch\.exp\(torch\.mul\(torch\.neg\(qn\d*\), {C}\)\)\), 1\)")
            F["sw_gas_reduce"], F["sw_ng_gas"], F["sw_random_mask"] = False, 0, False
            F["rad_updated_qn"] = bool(re.search(r"qn0 = torch\.relu\(torch\.add\(qn, dqn0\)\)", code))
            return self._tail(P, F, code)
        P["gas_optics_model_sw1.xmin"], P["gas_optics_model_sw1.xdiv"] = self.c(m.group(1)), self.c(m.group(2))
        m = re.search(rf"torch\.div\(torch\.sub\(vmr_h2o_2\d*, {C}\), {C}\)", code)
        if m:
            P["gas_optics_model_sw1.xmin_h2o"], P["gas_optics_model_sw1.xdiv_h2o"] = self.c(m.group(1)), self.c(m.group(2))
        for i, w in ((1, 2), (2, 4)):      # softsign layers 3, 4 (model 1) and 5, 6 (model 2) of the first humidity variant
            self.linear(f"gas_optics_model_sw{i}.mlp1", r"x\d+", soft, which=w)
            self.linear(f"gas_optics_model_sw{i}.mlp2", r"x\d+", soft, which=w + 1)
        taus = [(l, r) for l, r in self.named if re.fullmatch(r"tau\d+", l) and re.search(r"torch\.mul\(col_dry_crm_1\d*, torch\.pow\(" + LIN + r", 8\)\)", r)]
        for i, (l, r) in enumerate(taus[:2]):
            mm = re.search(LIN, r)
            P[f"gas_optics_model_sw{i + 1}.mlp3.weight"], P[f"gas_optics_model_sw{i + 1}.mlp3.bias"] = self.lin_wb(mm.groups()[1:5])
        F["sw_gas_ystd"] = not taus
        if not taus:      # another sub-generation (num27378, num45826, num74834): tau = N (ystd y + ymean)^8 as in the LW model, no 1e-17
            pat = rf"torch\.mul\(col_dry_crm_1\d*, torch\.pow\(torch\.add\(torch\.mul\({C}, " + LIN + rf"\), {C}\), 8\)\)"
            taus = [re.search(pat, r) for l, r in self.named if re.fullmatch(r"tau_sw(_scat)?\d*", l) and re.search(pat, r)]
            for i, mm in enumerate(taus[:2]):
                g = mm.groups()
                P[f"gas_optics_model_sw{i + 1}.ystd"], P[f"gas_optics_model_sw{i + 1}.ymean"] = self.c(g[0]), self.c(g[-1])
                P[f"gas_optics_model_sw{i + 1}.mlp3.weight"], P[f"gas_optics_model_sw{i + 1}.mlp3.bias"] = self.lin_wb(g[2:6])
        F["sw_gas_reduce"] = self.linear("gas_optics_sw_reduce1", r"tau_sw\w*", r"torch\.softplus\({LIN}, 1\., 20\.\)", required=False)
        if F["sw_gas_reduce"]:
            self.linear("gas_optics_sw_reduce2", r"tau_sw_scat\w*", r"torch\.softplus\({LIN}, 1\., 20\.\)")
        F["sw_ng_gas"] = P["gas_optics_model_sw1.mlp3.weight"].shape[0]
        F["sw_random_mask"] = "torch.rand_like(tau_sw1)" in code
        return self._tail(P, F, code)

    def _tail(self, P, F, code):
        F["rad_updated_T"] = bool(re.search(r"T\d* = torch\.relu\(torch\.add\(T, dT\d*\)\)", code))    # radiation on the updated temperature (all but num36398)
        F["rnn3_last_mul"] = "last_h = torch.mul(hidden" in code      # release / surface heads read rnn2's last state TIMES the third RNN's (num36398)
        F["sw_scat_clamp"] = bool(re.search(r"tau_sw_scat_tot\d* = torch\.clamp\(tau_sw_scat_tot\d*, 1", code))
        F["cld_qn_updated"] = bool(re.search(r"qn_crm\d* = torch\.relu\(torch\.add\(qn_crm\d*, ", code))     # cloud water paths of the radiation scheme
        F["sfc_sw_down"] = "flux_sw_dn_sfc" in code              # first surface output: downward (num82174) instead of net shortwave
        F["cld_liq_from_updated_T"] = bool(re.search(r"torch\.sub\(torch\.squeeze\(T_crm\d*\), 253\.16", code))   # cloud LW optics: ramp on the UPDATED T_crm
        F["rad_updated_qv"] = bool(re.search(r"qv0 = torch\.relu\(torch\.add\(qv, dqv0\)\)", code))
        self.const("solar_weights", r"incoming_toa\d+", rf"torch\.mul\(incoming_toa\d*, {C}\)")
        a = self.find(r"SOLL", rf"torch\.mul\({C}, sw_dir_dn_\w+\)", required=False)
        b = self.find(r"SOLS", rf"torch\.mul\({C}, sw_dir_dn_\w+\)", required=False)
        F["albedo_mix_learned"] = a is not None
        if a is not None:
            P["mix_near"], P["mix_vis"] = self.c(a.group(1)).reshape(-1), self.c(b.group(1)).reshape(-1)
        self.const("yscale_T60", r"dT_rad\d+", rf"torch\.mul\(dT_rad\d*, {C}\)")
        self.const("yscale_sca_rad", r"out_sfc_rad\d+", rf"torch\.mul\(out_sfc_rad\d*, {C}\)")
        self.const("yscale_lev", r"out_denorm\d+", rf"torch\.div\(out_new\d+, {C}\)", required=False) or \
            self.const("yscale_lev", r"out_denorm", rf"torch\.div\(out_new\d+, {C}\)")
        self.const("yscale_sca", r"out_sfc_denorm", rf"torch\.div\(out_sfc\d*, {C}\)")
        m = re.search(r"repeats = torch\.tensor\(\[([\d, ]+)\]", code)
        F["band_repeats"] = [int(v) for v in m.group(1).split(",")] if m else None
        m2 = re.search(r"nu_low = torch\.tensor\(\[([\d, ]+)\]", code)
        if m and m2:      # Slingo / Ebert-Curry band of every g-point, as the serialised bucketize + repeat_interleave evaluate
            import bisect
            wav = [1e4 / float(v) for v in m2.group(1).split(",")]
            sidx = [bisect.bisect_right([0.7, 1.25, 2.38], w) for w in wav]
            F["band_idx"] = [b for b, r in zip(sidx, F["band_repeats"]) for _ in range(r)]
        else:
            # a later form of the same map: columns of the 4-band coefficient stack copied into slices of an empty (6, ng) tensor
            cp = re.findall(r"torch\.copy_\(torch\.slice\(torch\.slice\(y\), 1, (\d+)(?:, (\d+))?\), torch\.slice\(torch\.slice\(x\d+\), 1, (\d+), \d+\)\)",
                            "\n".join(self.sinks))
            ng = P["gas_optics_lw_reduce1.weight"].shape[0]
            if cp:
                idx = [None] * ng
                for a, b, band in cp:
                    for g in range(int(a), int(b) if b else ng):
                        idx[g] = int(band)
                F["band_idx"] = idx if None not in idx else None
        # yet another: the band quantities multiplied by a learned (4, ng) band -> g-point matrix (ONE constant for liquid and ice)
        mm = re.findall(rf"torch\.matmul\((?:k\d*|kscag?_sw_cld_\w+), {C}\)", code)
        F["cld_band_matrix"] = bool(mm)
        if mm:
            if len(set(mm)) != 1 or len(mm) != 6:
                raise KeyError("cloud band matrix: expected one constant used six times")
            P["cloud_band_to_gpt"] = self.c(mm[0])
        # the ice SW optics read the ICE effective radius (later exports) or, as first serialised, the liquid one clamped to 13..130
        F["ice_optics_on_ice_radius"] = not bool(re.search(r"re_um\d* = torch\.clamp\(liq_eff_rad\d*, 13\., 130\.\)", code))
        # g-points [0, n_ir) take the near-infrared surface albedo, [n_ir, n_mix_end) the mixed one, the rest the visible one
        a1 = re.search(r"torch\.slice\(albedo_surf_dir_sw, 0, 0, (\d+)\)", code)
        a2 = re.search(r"torch\.slice\(albedo_surf_dir_sw, 0, %s, (\d+)\)" % (a1.group(1) if a1 else "x"), code)
        if a1 and a2:
            F["n_ir"], F["n_mix_end"] = int(a1.group(1)), int(a2.group(1))
        # the 4-band Slingo / Ebert-Curry coefficient lists as serialised (compared with the tables of the restatement by the golden script)
        F["cloud_tables"] = [[float(v) for v in m_.group(1).split(",")] for m_ in re.finditer(r"^  _\d+ = \[([-\d.e, ]+)\]$", code, re.M)
                             if len(m_.group(1).split(",")) == 4]
        F["mem_channels"] = int(re.search(r"torch\.slice\(torch\.slice\(torch\.slice\(rnn\w*_mem\w*\), 1\), 2, 0, (\d+)\)", code).group(1)) + 1
        F["unnamed"] = sorted(int(k[1:]) for k in self.cm if int(k[1:]) not in self.used)
        return P, F


def load_export(path):
    """A `*_wrapped.pt` export -> (state_dict under the names the constants had before freezing, cfg: the switches of its code variant).
    torch.jit.load executes nothing from the file; the `_gpu` exports load with their constants mapped to the CPU."""
    m = torch.jit.load(path, map_location="cpu")
    P, F = Extractor(m).run()
    if F["unnamed"]:
        raise RuntimeError(f"frozen export {path}: constants {F['unnamed']} of its serialised code are not named -- an unknown code variant")
    if F.get("band_idx") is None:
        F["band_idx"] = [0] * F["nreg"]
    return {k: v.cpu() for k, v in P.items()}, F


if __name__ == "__main__":
    import glob
    import sys
    torch.set_num_threads(4)
    for f in sys.argv[1:] or sorted(glob.glob("/root/reference/rnn/saved_models/*_cpu_wrapped.pt")):
        ex = Extractor(torch.jit.load(f, map_location="cpu"))
        try:
            P, F = ex.run()
            print(f.split("_num")[1].split("_script")[0], {k: v for k, v in F.items()}, len(P))
        except Exception as e:
            print(f.split("_num")[1].split("_script")[0], "FAILED", repr(e), "unnamed so far:", sorted(int(k[1:]) for k in ex.cm if int(k[1:]) not in ex.used)[:20])
