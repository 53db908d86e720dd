"""CPU emulation (round 3): error of an fp16 Winograd F(2x2,3x3) residual tower (fp32 transforms, transformed
operands rounded to fp16, fp32 accumulation) against the direct fp16 tower and fp64, for the weight sets of tests/nethelpers.py.
usage: python tools/winograd_error_emulation.py"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
import nethelpers as H
from cchess_zero_amd.net import PolicyValueModule
torch.set_num_threads(8)
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float32)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float32)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float32)
def wino_conv(x, w, b, dt):
    """x [N,C,9,10] (already rounded to dt), w [O,C,3,3] fp32, F(2x2,3x3); transformed operands rounded to dt."""
    r = (lambda t: t.to(dt).float()) if dt is not None else (lambda t: t)
    N, C, Hh, Ww = x.shape
    xp = F.pad(x, (1, 1 + (Ww % 2), 1, 1 + (Hh % 2)))          # pad to even output size: 10 x 10
    Ho, Wo = Hh + (Hh % 2), Ww + (Ww % 2)
    # tiles of 4x4 with stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [N,C,th,tw,4,4]
    U = r(torch.einsum("ij,nctsjk,lk->nctsil", Bt, t, Bt))      # B^T d B
    V = r(torch.einsum("ij,ocjk,lk->ocil", G, w, G))            # G g G^T  [O,C,4,4]
    M = torch.einsum("nctsil,ocil->notsil", U, V)               # elementwise in the transform domain, summed over c
    Y = torch.einsum("ij,notsjk,lk->notsil", At, M, At)         # [N,O,th,tw,2,2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], Ho, Wo)[:, :, :Hh, :Ww]
    return y + b.view(1, -1, 1, 1)
def fwd(m, x, dt, wino):
    r = (lambda t: t.to(dt).float()) if dt is not None else (lambda t: t)
    def cb(c, h, res=None, use_w=False):
        w, b = c.folded()
        y = wino_conv(h, w, b, dt) if (use_w and w.shape[-1] == 3) else F.conv2d(h, r(w), b, padding=w.shape[-1] // 2)
        return y if res is None else y + res
    h = r(torch.relu(cb(m.conv_in, r(x))))
    for a, b in m.blocks:
        t = r(torch.relu(cb(a, h, use_w=wino in ("both", "first"))))
        h = r(torch.relu(cb(b, t, res=h, use_w=wino in ("both", "second"))))
    wp, bp = m.policy_conv.folded(); wv, bv = m.value_conv.folded()
    p = torch.relu(F.conv2d(h, wp, bp)).permute(0, 2, 3, 1).reshape(h.shape[0], 180)
    v = torch.relu(F.conv2d(h, wv, bv)).permute(0, 2, 3, 1).reshape(h.shape[0], 90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(v))))
class N: pass
x = torch.from_numpy(H.positions(48, 2)).permute(0, 3, 1, 2)
for ws in ("glorot", "trained_like"):
    n = N(); n.module = PolicyValueModule(7, seed=1); n.refresh = lambda: None
    H.WEIGHT_SETS[ws](n)
    md = PolicyValueModule(7, seed=1); md.load_state_dict(n.module.state_dict()); md = md.double()
    with torch.no_grad():
        lr, vr = md(x.double())
        # sanity: winograd in fp32 == direct
        l0, v0 = fwd(n.module, x, None, "both")
        print(ws, "fp32 winograd vs fp64 direct: dlogit %.3g" % float((l0.double() - lr).abs().max()))
        for wino in ("none", "second", "both"):
            l, v = fwd(n.module, x, torch.float16, wino)
            e = H.errors(l.numpy(), v.numpy(), lr.float().numpy(), vr.float().numpy())
            print(ws, "fp16", wino, "dlogit %.3g rel %.3g dprob %.3g dvalue %.3g" % (e["dlogit"], e["dlogit_rel"], e["dprob"], e["dvalue"]))
