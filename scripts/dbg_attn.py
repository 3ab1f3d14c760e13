"""Locate attention errors: per-row relative error of ltxk_flash_attn vs an fp32 torch SDPA.  argv: B H Tq Tk"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
B, H, Tq, Tk = map(int, sys.argv[1:5])
dev = torch.device("cuda:0")
D = H * 128
g = torch.Generator().manual_seed(B * 100 + Tq)
q = torch.randn(B, Tq, D, generator=g).to(torch.bfloat16).to(dev)
k = torch.randn(B, Tk, D, generator=g).to(torch.bfloat16).to(dev)
v = torch.randn(B, Tk, D, generator=g).to(torch.bfloat16).to(dev)
Tp = (Tk + 63) // 64 * 64
vt = torch.zeros(B, D, Tp, dtype=torch.bfloat16, device=dev)
vt[:, :, :Tk] = v.transpose(1, 2)
out = torch.empty(B * Tq, D, dtype=torch.bfloat16, device=dev)
ops.flash_attn(q.reshape(B * Tq, D), k.reshape(B * Tk, D), vt, out, B, H, Tq, Tk, 1.0 / math.sqrt(128))
torch.cuda.synchronize()
qf = q.float().reshape(B, Tq, H, 128).transpose(1, 2)
kf = k.float().reshape(B, Tk, H, 128).transpose(1, 2)
vf = v.float().reshape(B, Tk, H, 128).transpose(1, 2)
s = qf @ kf.transpose(-1, -2) / math.sqrt(128)
ref = (torch.softmax(s, -1) @ vf)                   # (B,H,Tq,128)
o = out.float().reshape(B, Tq, H, 128).transpose(1, 2)
err = (o - ref).norm(dim=-1) / ref.norm(dim=-1)       # (B,H,Tq)
print("overall rel-L2", float((o - ref).norm() / ref.norm()))
bad = (err > 0.02)
print("bad rows:", int(bad.sum()), "of", bad.numel())
for b in range(B):
    for h in range(H):
        rows = bad[b, h].nonzero().flatten().tolist()
        if rows:
            print(f"b={b} h={h}: {len(rows)} bad rows, first {rows[:24]}  max err {float(err[b,h].max()):.3f}")
e2 = err.reshape(B * H, Tq)
print("mean err by row%64:", [round(float(e2[:, i::64].mean()), 4) for i in range(0, 64, 4)])
# d-pattern of a bad row
if bad.any():
    b_, h_, r_ = [int(x) for x in bad.nonzero()[0]]
    dd = (o[b_, h_, r_] - ref[b_, h_, r_]).abs()
    print("row", b_, h_, r_, "abs err by d block of 32:", [round(float(dd[i*32:(i+1)*32].mean()), 4) for i in range(4)], "ref mag", float(ref[b_,h_,r_].abs().mean()))
    print("ratio o/ref median:", float((o[b_,h_,r_]/ref[b_,h_,r_]).median()))
