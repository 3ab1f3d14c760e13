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


BIG_SHAPES = [(320, 256, 64), (2560, 1024, 256), (700, 512, 192), (2560, 4096, 1024), (1296, 768, 448), (2048, 2048, 512)]


@pytest.mark.parametrize("M,N,K", BIG_SHAPES)
def test_gemm_320x256_tile_equals_160x256_tile(dev, M, N, K, monkeypatch, ab_lib):
    """The 320x256-tile kernel (gemm.hip, taken for FF1-sized launches) walks K in the same order with the same MFMA as
    the 160x256 one: forced on (LTXK_GEMM_BIG=2 in the A/B build of the library) it must give the same bits, for every epilogue it supports, with and
    without bias, with a ragged last row tile (M=700, 1296), a strided A (lda = K + 64) and when its output is a strided view."""
    from mlx_video_amd import ops
    g = torch.Generator(device=dev).manual_seed(M * 3 + N + K)
    a = torch.randn((M, K + 64), generator=g, device=dev).to(BF)[:, :K]          # row stride lda = K + 64
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(BF)
    b = (torch.randn(N, generator=g, device=dev) * 0.1).to(BF)
    for epi in (0, 1, 2):
        for bias in (b, None):
            outs = {}
            for mode in ("0", "2", "3"):                     # 160-row tiles, the 320 x 256 tile, the 256 x 256 tile
                monkeypatch.setenv("LTXK_GEMM_BIG", mode)
                buf = torch.full((M + 1, N + 64), 7.0, device=dev, dtype=BF)
                ops.gemm(a, w, bias, out=buf[:M, :N], epilogue=epi)
                torch.cuda.synchronize()
                assert torch.all(buf[M:] == 7.0) and torch.all(buf[:, N:] == 7.0), "wrote outside the output view"
                outs[mode] = buf[:M, :N].clone()
            assert torch.equal(outs["0"], outs["2"]), f"320-row tile: epilogue {epi} bias={bias is not None}"
            assert torch.equal(outs["0"], outs["3"]), f"256-row tile: epilogue {epi} bias={bias is not None}"
    monkeypatch.setenv("LTXK_GEMM_BIG", "2")
    acc = a.float() @ w.float().t()
    ref = torch.nn.functional.gelu((acc + b.float()).to(BF).float(), approximate="tanh")
    out = ops.gemm(a, w, b, epilogue=1)
    parity.check(f"gemm.big_tile_gelu_{M}x{N}x{K}_vs_torch_fp32", float((out.float() - ref).norm() / ref.norm()), 6e-3)


def test_gemm_ff1_takes_the_320x256_tile(dev, monkeypatch):
    """Default dispatch at the FF1 shape (M=2560, N=16384, K=4096): same bits as with the big tile switched off."""
    from mlx_video_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn((2560, 4096), generator=g, device=dev).to(BF)
    w = (torch.randn((16384, 4096), generator=g, device=dev) * 0.02).to(BF)
    b = (torch.randn(16384, generator=g, device=dev) * 0.1).to(BF)
    from mlx_video_amd import _lib
    y1 = ops.gemm(a, w, b, epilogue=1)                       # the product library
    monkeypatch.setenv("LTXK_GEMM_BIG", "0")
    with _lib.use_library(_lib.AB_LIB_PATH):                 # the A/B build with the 320x256 tile switched off
        y0 = ops.gemm(a, w, b, epilogue=1)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("M,N,K", [(2560, 1024, 256), (700, 512, 192), (2048, 768, 320)])
def test_gemm_320x256_tile_sumsq_transposed_and_split_outputs(dev, M, N, K, monkeypatch, ab_lib):
    """Row statistics, the transposed (V^T) output - vector stores at T % 4 == 0, element stores at T = 350 - and the
    split k | V^T output of the 320x256-tile kernel against the 160x256 one: same bits."""
    from mlx_video_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N * 7 + K)
    a = torch.randn((M, K), generator=g, device=dev).to(BF)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(BF)
    b = (torch.randn(N, generator=g, device=dev) * 0.1).to(BF)
    T = M // 2
    ld = (T + 63) // 64 * 64
    res = {}
    for mode in ("0", "2", "3"):
        monkeypatch.setenv("LTXK_GEMM_BIG", mode)
        ss = torch.full((M, N // 64 + 1), -1.0, device=dev, dtype=torch.float32)
        y = ops.gemm(a, w, b, sumsq=ss)
        vt = torch.full((2, N, ld), 3.0, device=dev, dtype=BF)
        ops.gemm(a, w, b, out=vt, out_tokens_per_batch=T)
        ns = 256
        k2 = torch.empty((M, ns), device=dev, dtype=BF)
        v2 = torch.full((2, N - ns, ld), 3.0, device=dev, dtype=BF)
        ss2 = torch.full((M, ns // 64), -1.0, device=dev, dtype=torch.float32)
        ops.gemm(a, w, b, out=k2, out2=v2, n_split=ns, out_tokens_per_batch=T, sumsq=ss2)
        torch.cuda.synchronize()
        res[mode] = (y, ss, vt, k2, v2, ss2)
    for name, x0, x2, x3 in zip(("y", "sumsq", "vt", "split.k", "split.vt", "split.sumsq"), res["0"], res["2"], res["3"]):
        assert torch.equal(x0, x2), name + " (320-row tile)"
        assert torch.equal(x0, x3), name + " (256-row tile)"
    y, ss, vt, k2, v2, ss2 = res["2"]
    assert torch.all(ss[:, -1] == -1.0) and torch.all(vt[:, :, T:] == 3.0)
    assert torch.equal(vt[:, :, :T], y.reshape(2, T, N).transpose(1, 2))
    assert torch.equal(k2, y[:, :256]) and torch.equal(v2[:, :, :T], vt[:, 256:, :T])
    ref = (y.float() ** 2).reshape(M, N // 64, 64).sum(-1)
    assert float((ss[:, :-1] - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(300, 320, 128), (160, 448, 64), (2560, 576, 256), (33, 64, 64)])
def test_gemm_sumsq_when_N_is_not_a_multiple_of_256(dev, M, N, K):
    """Row statistics with a ragged last column tile (N % 256 != 0; the C ABI only asks for N % 64 == 0): the waves of that
    tile whose 64-column block lies past N must not store a partial - the slot is the NEXT row's first partial (a race
    with its real writer) or, on the last row, past the buffer.  Canary-padded buffer with sumsq_ld == N/64 exactly."""
    from mlx_video_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g, device=dev).to(BF)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(BF)
    b = (torch.randn(N, generator=g, device=dev) * 0.1).to(BF)
    P = N // 64
    for _ in range(3):
        buf = torch.full((M * P + 64,), -7.0, device=dev, dtype=torch.float32)
        ss = buf[:M * P].view(M, P)
        y = ops.gemm(a, w, b, sumsq=ss)
        torch.cuda.synchronize()
        assert torch.all(buf[M * P:] == -7.0), "sumsq written past the buffer"
        ref = (y.float() ** 2).reshape(M, P, 64).sum(-1)
        assert float((ss - ref).abs().max() / ref.abs().max()) < 1e-5


def _all_outputs(ops, a, w, b, res, gate, grow, T, ld, N, M, dev):
    """Every output form of ltxk_gemm_bf16 for one operand set: the six epilogues (with / without bias), row statistics,
    V^T, and the split k | V^T launch."""
    out = {}
    for epi in (0, 1, 2, 3, 4, 5):
        for bias in (b, None):
            kw = dict(epilogue=epi)
            if epi in (3, 4, 5):
                kw["resid"] = res
            if epi == 3:
                kw.update(gate=gate, gate_row=grow, gate_stride=N)
            if epi == 5:
                kw["alpha"] = 0.8
            buf = torch.full((M + 1, N + 64), 7.0, device=dev, dtype=BF)
            ss = torch.full((M, N // 64 + 1), -1.0, device=dev, dtype=torch.float32)
            ops.gemm(a, w, bias, out=buf[:M, :N], sumsq=ss if epi in (0, 3, 4) else None, **kw)
            torch.cuda.synchronize()
            assert torch.all(buf[M:] == 7.0) and torch.all(buf[:, N:] == 7.0), "wrote outside the output view"
            out[f"epi{epi}.bias{bias is not None}"] = buf[:M, :N].clone()
            if epi in (0, 3, 4):
                assert torch.all(ss[:, -1] == -1.0)
                out[f"epi{epi}.bias{bias is not None}.sumsq"] = ss
    vt = torch.full((2, N, ld), 3.0, device=dev, dtype=BF)
    ops.gemm(a, w, b, out=vt, out_tokens_per_batch=T)
    out["vt"] = vt
    if N > 256:
        ns = 256
        k2 = torch.empty((M, ns), device=dev, dtype=BF)
        v2 = torch.full((2, N - ns, ld), 3.0, device=dev, dtype=BF)
        ss2 = torch.full((M, ns // 64), -1.0, device=dev, dtype=torch.float32)
        ops.gemm(a, w, b, out=k2, out2=v2, n_split=ns, out_tokens_per_batch=T, sumsq=ss2)
        out.update({"split.k": k2, "split.vt": v2, "split.sumsq": ss2})
    torch.cuda.synchronize()
    return out


def _operands(M, N, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randn((M, K + 64), generator=g, device=dev).to(BF)[:, :K]          # row stride lda = K + 64
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(BF)
    b = (torch.randn(N, generator=g, device=dev) * 0.1).to(BF)
    res = torch.randn((M, N), generator=g, device=dev).to(BF)
    gate = torch.randn((3, N), generator=g, device=dev).to(BF)
    grow = torch.randint(0, 3, (M,), generator=g, device=dev, dtype=torch.int32)
    return a, w, b, res, gate, grow


@pytest.mark.parametrize("M,N,K", [(1280, 1024, 256), (700, 512, 192), (2560, 4096, 1024), (1296, 768, 448), (66, 320, 128), (160, 128, 64),
                                   (2048, 2048, 512), (350, 576, 640)])
def test_gemm_128_column_tile_equals_256_column_tile(dev, M, N, K, monkeypatch, ab_lib):
    """The 128-column tile (80 x 32 per wave; round 4, taken where 256-column tiles leave CUs idle: M=1280, N=4096 is exactly
    256 tiles of 160 x 128) walks K in the same order with the same MFMA as the 256-column one, and hands the row statistic
    of a 64-column block from its even wave to its odd wave so that the 16 additions per lane happen in the same order:
    forced (LTXK_GEMM_NT in the A/B build) it must give the same bits as the 256-column tile for every epilogue, with and
    without bias, row statistics, V^T and split outputs, ragged last row / column tiles, a strided A and a strided output view."""
    from mlx_video_amd import ops
    a, w, b, res, gate, grow = _operands(M, N, K, dev, M * 5 + N + K)
    T = M // 2
    ld = (T + 63) // 64 * 64
    monkeypatch.setenv("LTXK_GEMM_KSPLIT", "-1")
    monkeypatch.setenv("LTXK_GEMM_BIG", "0")
    outs = {}
    for nt in ("4", "2"):
        monkeypatch.setenv("LTXK_GEMM_NT", nt)
        outs[nt] = _all_outputs(ops, a, w, b, res, gate, grow, T, ld, N, M, dev)
    for name in outs["4"]:
        assert torch.equal(outs["4"][name], outs["2"][name]), f"{name}: the 128-column tile differs from the 256-column tile"
    for tt in ("1", "2", "3", "4", "5"):                                   # every row-tile height of the 128-column tile
        monkeypatch.setenv("LTXK_GEMM_TT", tt)
        ss = torch.empty((M, N // 64), device=dev, dtype=torch.float32)
        y = ops.gemm(a, w, b, epilogue=3, resid=res, gate=gate, gate_row=grow, gate_stride=N, sumsq=ss)
        torch.cuda.synchronize()
        assert torch.equal(y, outs["4"]["epi3.biasTrue"]) and torch.equal(ss, outs["4"]["epi3.biasTrue.sumsq"][:, :-1]), f"TT={tt}"


@pytest.mark.parametrize("M,N,K,S", [(64, 512, 1024, 4), (64, 4096, 4096, 8), (320, 1024, 2048, 2), (33, 320, 512, 3), (200, 768, 4096, 5), (1, 128, 1024, 4)])
def test_gemm_split_k_form(dev, M, N, K, S, monkeypatch, ab_lib):
    """The weight-streaming form for small M (split-K on the 128-column tile: fp32 slice tiles in the caller's scratch, a
    second launch that sums them in slice order and applies the epilogue): every epilogue and output form against the
    single-pass kernel.  One rounding of an fp32 sum taken in another order: <= 1 bf16 ulp on >= 99.8 % of the outputs, never
    more than 2; deterministic; the row statistic equals the sum of squares of what was stored."""
    from mlx_video_amd import ops
    a, w, b, res, gate, grow = _operands(M, N, K, dev, M * 11 + N + K)
    T = max(M // 2, 1) if M % 2 == 0 else M
    nb = M // T
    ld = (T + 63) // 64 * 64
    monkeypatch.setenv("LTXK_GEMM_BIG", "0")

    def run():
        o = _all_outputs(ops, a, w, b, res, gate, grow, T, ld, N, M, dev) if nb == 2 else {}
        if nb != 2:
            for epi in (0, 1, 3):
                kw = dict(epilogue=epi)
                if epi == 3:
                    kw.update(resid=res, gate=gate, gate_row=grow, gate_stride=N)
                ss = torch.empty((M, N // 64), device=dev, dtype=torch.float32)
                o[f"epi{epi}"] = ops.gemm(a, w, b, sumsq=ss if epi != 1 else None, **kw)
                if epi != 1:
                    o[f"epi{epi}.sumsq"] = ss
            torch.cuda.synchronize()
        return o
    monkeypatch.setenv("LTXK_GEMM_KSPLIT", "-1")
    ref = run()
    monkeypatch.setenv("LTXK_GEMM_KSPLIT", str(S))
    got, again = run(), run()
    worst = 0.0
    for name, r in ref.items():
        x = got[name]
        assert torch.equal(x, again[name]), f"{name}: split-K results differ between launches"
        if name.endswith("sumsq"):
            y = got[name[:-6]] if name != "split.sumsq" else got["split.k"]
            cols = x.shape[1] - (1 if x.shape[1] * 64 > y.shape[1] else 0)
            sq = (y.float() ** 2).reshape(M, -1, 64).sum(-1)
            assert float((x[:, :cols] - sq).abs().max() / sq.abs().max()) < 1e-5, name
            continue
        d = (x.float() - r.float()).abs()
        # ulp of the output, floored at the ulp of rms/64 (an output that cancels to ~0 still carries the fp32 error of its
        # terms); the worst element is held to 2 ulps at the scale of the row-major value it was computed from (a GELU / gated
        # output is much smaller than the rounded pre-activation whose last bit flipped)
        rms = r.float().pow(2).mean().sqrt()
        ulp = torch.maximum(r.float().abs(), rms / 64).log2().floor().exp2() * 2.0 ** -7
        ulp_big = torch.maximum(r.float().abs(), rms).log2().floor().exp2() * 2.0 ** -7
        # (an activation maps a one-ulp flip of y near a binade boundary - ulp(y) = 2^-7 just above 1.0, gelu(1.0) = 0.84 has
        # ulp 2^-8 - onto up to 3 ulps of its output: seen on < 0.005 % of GELU outputs in scripts/fuzz_gemm_forms.py)
        # and the gate multiplies it: bf16(y * gate) moves by |gate| ulps of y when y flips by one (synthetic gates reach ~4)
        worst_ulps = 3.001 if name.startswith(("epi1", "epi2")) else (2.001 * max(1.0, float(gate.abs().max())) if name.startswith("epi3") else 2.001)
        assert float((d > ulp * 1.001).float().mean()) <= 2e-3 and float((d / ulp_big).max()) <= worst_ulps, \
            f"{name}: beyond 1 ulp on {float((d > ulp * 1.001).float().mean()):.2%}, max {float((d / ulp_big).max()):.1f} ulp"
        worst = max(worst, float((x.float() - r.float()).norm() / r.float().norm()))
    parity.auto(worst, 3e-3)
