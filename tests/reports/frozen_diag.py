"""GPU diagnostic: where (column, level, g-point) the HIP frozen-export path departs most from the float64 restatement, with the work-array
taps of the radiation scheme next to the restatement's (python tests/reports/frozen_diag.py <fixture> <case>).  Test infrastructure."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from test_physrnn_frozen import _load, _case, _f64, R
from climsim_amd.physrnn import physical_RNN_wrapped
fx, ci = sys.argv[1], int(sys.argv[2])
g, P, FL = _load(fx)
m = physical_RNN_wrapped(P, FL, max_batch=64)
x, s, mem, dr, ref = _case(g, P, ci)
d = lambda t: None if t is None else t.cuda()
got = [t.cpu() for t in m(d(x), d(s), d(mem), hx2=d(dr["hx2"]), mask_u=d(dr["mask_u"]))]
taps = {}
o64 = _f64(P, FL, x, s, mem, dr, taps)
o32 = R.forward(P, FL, x, s, mem, dr["hx2"], dr["mask_u"])
e = (got[0][..., 0].double() - o64[0][..., 0]).abs(); e[:, :4] = 0
b, l = np.unravel_index(int(e.argmax()), e.shape)
print("max err", e.max().item(), "at column", b, "level", l)
for L in range(max(0, l - 3), min(60, l + 4)):
    print(L, "hip %.6e ref %.6e f32 %.6e f64 %.6e" % (got[0][b, L, 0], ref[0][b, L, 0], o32[0][b, L, 0], o64[0][b, L, 0]))
print("col errs other channels", [(got[0][b, :, c].double() - o64[0][b, :, c]).abs().max().item() for c in range(6)])
print("sfc hip", got[1][b].tolist()); print("sfc f64", o64[1][b].tolist())
for k in ("tau_sw","ssa","asy","tau_lw"):
    t=taps[k][b]; print(k, "min %.3e max %.3e"%(t.min(), t.max()), "at level", t[l].tolist()[:6])
print("mu0", (s[b, 6]).item(), "per-column max err", e.amax(1).topk(5))
import ctypes
from climsim_amd import _lib
from climsim_amd.emulator import _ptr
B = x.shape[0]
def tap(which, n):
    t = torch.empty(n, device="cuda")
    rc = _lib.lib().csa_phys_tap(m._h, which, B, _ptr(t), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return t.cpu()
S2 = tap(6, 60 * B * 48).view(60, B, 48)
CS = tap(3, 50 * B * 48).view(50, B, 48)
for name, k0 in (("tau_sw", 0), ("ssa", 16), ("asy", 32)):
    h_ = S2[:, b, k0:k0 + 16].double(); r_ = taps[name][b]
    er = ((h_ - r_).abs() / (r_.abs() + 1e-30))
    er[r_.abs() < 1e-12] = 0
    L_, g_ = np.unravel_index(int(er.argmax()), er.shape)
    print(name, "max rel err %.3e at level %d g %d: hip %.6e ref %.6e" % (er.max(), L_, g_, h_[L_, g_], r_[L_, g_]))
ct = taps["c_tau"][b][10:]
h_ = CS[:, b, 0:16].double(); er = (h_ - ct).abs() / (ct.abs() + 1e-30); er[ct.abs() < 1e-12] = 0
L_, g_ = np.unravel_index(int(er.argmax()), er.shape)
print("c_tau max rel err %.3e at crm level %d g %d: hip %.6e ref %.6e" % (er.max(), L_, g_, h_[L_, g_], ct[L_, g_]))
print("area_frac at that level", taps["area_frac"][b, L_].tolist())
print("---- absolute")
th, tr_ = S2[:, b, 0:16].double(), taps["tau_sw"][b]
sh, sr = S2[:, b, 16:32].double() * th, taps["ssa"][b] * tr_
for name, h_, r_ in (("tau", th, tr_), ("tau*ssa", sh, sr)):
    er = (h_ - r_).abs()
    for _ in range(4):
        L_, g_ = np.unravel_index(int(er.argmax()), er.shape)
        print(name, "abs err %.3e at level %d g %d: hip %.6e ref %.6e" % (er[L_, g_], L_, g_, h_[L_, g_], r_[L_, g_]))
        er[L_, g_] = 0
ta, ts = taps["tau_abs"][b], taps["tau_sca"][b]
print("gas tau_abs / tau_sca ref at level 52:", ta[52].tolist()[:4], ts[52].tolist()[:4])
print("---- XR inputs at level 58")
cap = []
orig = R._lin
def lin(P_, name, x_):
    if name == "gas_optics_model_sw1.mlp1": cap.append(x_.detach().clone())
    return orig(P_, name, x_)
R._lin = lin
R.forward(P, FL, x, s, mem, dr["hx2"], dr["mask_u"])
R._lin = orig
XR = tap(5, 60 * B * 24).view(60, B, 24)
for L_ in (58, 14):
    print("level", L_, "hip XR", [round(v, 5) for v in XR[L_, b, :10].tolist()])
    print("   ref variant1", [round(v, 5) for v in cap[0][b, L_].tolist()], "variant2 x2", round(cap[1][b, L_, 2].item(), 5))
print("area_frac level 58:", [round(v, 4) for v in taps["area_frac"][b, 48].tolist()])
