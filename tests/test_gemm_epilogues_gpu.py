"""Every GEMM epilogue x (bias | no bias) x (short | long K) against a torch fp32 formula, launched repeatedly (also in
place on the residual): the counted-vmcnt prologue of ltxk_gemm_bf16 assumes a fixed number of epilogue-operand loads per
lane; a load the optimiser drops (dead bias under EPI_SCALE_RES, bias == NULL) once made a LoRA merge race its first
stage - wrong and different from launch to launch.  Results must be right AND bit-identical across launches."""
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SHAPES = [(2048, 512, 64), (512, 512, 128), (1296, 512, 512), (2560, 4096, 4096), (300, 512, 192), (96, 256, 64), (640, 768, 448)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_epilogues_deterministic(dev, M, N, K):
    from mlx_video_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    worst = 0.0
    for epi in (0, 1, 2, 3, 4, 5):
        for use_bias in (True, False):
            a = torch.randn((M, K), generator=g, device=dev).to(BF)
            w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(BF)
            b = (torch.randn(N, generator=g, device=dev) * 0.1).to(BF) if use_bias else None
            res = torch.randn((M, N), generator=g, device=dev).to(BF)
            gate = torch.randn((3, N), generator=g, device=dev).to(BF)
            grow = torch.randint(0, 3, (M,), generator=g, device=dev, dtype=torch.int32)
            kw = dict(epilogue=epi)
            if epi in (3, 4, 5):
                kw["resid"] = res
            if epi == 3:
                kw.update(gate=gate, gate_row=grow, gate_stride=N)
            if epi == 5:
                kw["alpha"] = 0.8
            acc = a.float() @ w.float().t()
            yb = (acc + (b.float() if b is not None else 0)).to(BF).float()
            if epi == 1:
                ref = torch.nn.functional.gelu(yb, approximate="tanh")
            elif epi == 2:
                ref = torch.nn.functional.silu(yb)
            elif epi == 3:
                ref = res.float() + (yb * gate.float()[grow.long()]).to(BF).float()
            elif epi == 4:
                ref = res.float() + yb
            elif epi == 5:
                ref = res.float() + (0.8 * acc).to(BF).float()
            else:
                ref = yb
            outs = []
            for it in range(4):
                if it % 2 and "resid" in kw:              # in place on the residual, as the model does
                    out = res.clone()
                    ops.gemm(a, w, b, out=out, **dict(kw, resid=out))
                else:
                    out = torch.empty((M, N), device=dev, dtype=BF)
                    ops.gemm(a, w, b, out=out, **kw)
                outs.append(out)
            torch.cuda.synchronize()
            for o in outs[1:]:
                assert torch.equal(outs[0], o), f"epilogue {epi} bias={use_bias}: results differ between launches"
            worst = max(worst, float((outs[0].float() - ref).norm() / ref.norm()))
    parity.auto(worst, 6e-3)
