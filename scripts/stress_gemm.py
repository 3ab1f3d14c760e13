"""Determinism / correctness stress of ltxk_gemm_bf16 over epilogues and shapes (vs torch fp32), repeated launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
shapes = [(2048, 512, 64), (512, 2048, 64), (512, 512, 128), (5184, 512, 512), (5184, 2048, 512), (5184, 512, 2048), (1296, 512, 512),
          (2560, 4096, 4096), (1280, 4096, 4096), (2048, 4096, 4096), (300, 512, 192), (96, 256, 64), (5184, 1536, 512)]
for (M, N, K) in shapes:
    for epi in (0, 1, 3, 4, 5):
        a = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
        b = (torch.randn(N, generator=g, device=dev) * 0.1).to(torch.bfloat16) if epi != 5 else None
        res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
        U = 3
        gate = torch.randn((U, N), generator=g, device=dev).to(torch.bfloat16)
        grow = torch.randint(0, U, (M,), generator=g, device=dev, dtype=torch.int32)
        kw = dict(epilogue=epi)
        if epi in (3, 4, 5):
            kw.update(resid=res)
        if epi == 3:
            kw.update(gate=gate, gate_row=grow, gate_stride=N)
        if epi == 5:
            kw.update(alpha=0.8)
        acc = a.float() @ w.float().t()
        y = acc + (b.float() if b is not None else 0)
        yb = y.to(torch.bfloat16).float()
        if epi == 1:
            ref = torch.nn.functional.gelu(yb, approximate="tanh")
        elif epi == 3:
            ref = res.float() + (yb * gate.float()[grow.long()]).to(torch.bfloat16).float()
        elif epi == 4:
            ref = res.float() + yb
        elif epi == 5:
            ref = res.float() + (0.8 * acc).to(torch.bfloat16).float()
        else:
            ref = yb
        outs = []
        for it in range(6):
            out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            if it % 2:      # in place on the residual, as the model does
                out = res.clone(); kw2 = dict(kw)
                if "resid" in kw2: kw2["resid"] = out
                ops.gemm(a, w, b, out=out, **kw2)
            else:
                ops.gemm(a, w, b, out=out, **kw)
            outs.append(out.float())
        torch.cuda.synchronize()
        errs = [float((o - ref).norm() / ref.norm()) for o in outs]
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        flag = "" if (max(errs) < 6e-3 and same) else "   <<<<<< BAD"
        if flag: bad += 1
        print(f"M={M} N={N} K={K} epi={epi}: rel err max {max(errs):.2e} deterministic={same}{flag}", flush=True)
print("BAD" if bad else "ALL OK", bad)
