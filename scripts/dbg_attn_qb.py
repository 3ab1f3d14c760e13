import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import _lib, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, H, Tq, Tk = 2, 32, 1280, 1280
D = H * 128
g = torch.Generator(device=dev).manual_seed(B * 7 + H + Tq + Tk)
q = torch.randn((B * Tq, D), generator=g, device=dev).to(BF)
k = torch.randn((B * Tk, D), generator=g, device=dev).to(BF)
vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(BF)
ss = (q.float() ** 2).reshape(B * Tq, D // 64, 64).sum(-1).contiguous()
w = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(BF)
cos, sin = torch.randn((H, Tq, 64), generator=g, device=dev), torch.randn((H, Tq, 64), generator=g, device=dev)
with _lib.use_library(_lib.AB_LIB_PATH):
    outs = {}
    for env in ("2", "0", "3"):
        os.environ["LTXK_FA_QB"] = env
        for name, kw in (("norope", dict(q_sumsq=ss, q_norm_weight=w, eps=1e-6)), ("rope", dict(q_sumsq=ss, q_norm_weight=w, cos=cos, sin=sin, eps=1e-6))):
            for rep in range(2):
                o = torch.zeros((B * Tq, D), dtype=BF, device=dev)
                ops.flash_attn(q, k, vt, o, B, H, Tq, Tk, 1.0 / math.sqrt(128), tail_split=False, **kw)
                torch.cuda.synchronize()
                outs[(env, name, rep)] = o
    for name in ("norope", "rope"):
        for env in ("0", "3"):
            a, b = outs[(env, name, 0)], outs[("2", name, 0)]
            d = (a.float() - b.float()).abs().reshape(B, Tq, H, 128)
            rows = (d.amax(dim=(0, 2, 3)) > 0).nonzero().flatten().tolist()
            print(name, "QB", env, "equal:", torch.equal(a, b), "repeatable:", torch.equal(outs[(env, name, 0)], outs[(env, name, 1)]),
                  "max abs diff", float(d.max()), "differing rows", len(rows), rows[:10], rows[-5:] if rows else [],
                  "heads", (d.amax(dim=(0, 1, 3)) > 0).nonzero().flatten().tolist()[:8], "chan", (d.amax(dim=(0, 1, 2)) > 0).nonzero().flatten().tolist()[:16])
