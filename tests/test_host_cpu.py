"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/ltxk.h declares (no compute calls), conditioning hooks, error behaviour."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            names |= set(re.findall(r"\b(ltxk_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from mlx_video_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"libltxk.so does not export {name}"
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    assert lib.ltxk_version() == 100


def test_product_refuses_cpu_tensors():
    from mlx_video_amd import ops
    from mlx_video_amd._lib import LtxkError
    a = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(LtxkError, match="no CPU fallback"):
        ops.gemm(a, a, None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mlx-video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"


def test_conditioning_hooks():
    from mlx_video_amd.conditioning import (LatentState, VideoConditionByKeyframeIndex, VideoConditionByLatentIndex,
                                            apply_conditioning, apply_denoise_mask, noise_blend)
    from oracle import sched as S
    g = torch.Generator().manual_seed(0)
    shape = (1, 8, 4, 2, 2)
    st = LatentState(torch.zeros(shape), torch.zeros(shape), torch.ones(1, 1, 4, 1, 1))
    c1 = torch.randn(1, 8, 1, 2, 2, generator=g)
    c2 = torch.randn(1, 8, 2, 2, 2, generator=g)
    out = apply_conditioning(st, [VideoConditionByLatentIndex(c1, 0, 1.0), VideoConditionByKeyframeIndex(c2, 3, 0.5)])
    rl, rc, rm = S.apply_conditioning(st.latent, st.clean_latent, st.denoise_mask,
                                      [("replace", c1, 0, 1.0), ("guide", c2, 3, 0.5)])
    assert torch.equal(out.latent, rl) and torch.equal(out.clean_latent, rc) and torch.equal(out.denoise_mask, rm)
    assert out.denoise_mask.flatten().tolist() == [0.0, 1.0, 1.0, 0.5]
    assert torch.equal(out.latent[:, :, 0], c1[:, :, 0]) and float(out.latent[:, :, 3].abs().max()) == 0.0   # guide keeps latent
    assert torch.equal(out.clean_latent[:, :, 3], c2[:, :, 0])                                              # clipped at F
    with pytest.raises(ValueError, match="out of bounds"):
        apply_conditioning(st, [VideoConditionByLatentIndex(c1, 4)])
    with pytest.raises(ValueError, match="does not match"):
        apply_conditioning(st, [VideoConditionByLatentIndex(torch.zeros(1, 8, 1, 3, 2), 0)])
    noise = torch.randn(shape, generator=g)
    nb = noise_blend(out, noise, 0.9)
    assert torch.allclose(nb.latent, S.noise_blend(noise, out.latent, out.denoise_mask, 0.9))
    assert torch.equal(nb.latent[:, :, 0], c1[:, :, 0])           # mask 0 => untouched
    x0 = torch.randn(shape, generator=g)
    assert torch.equal(apply_denoise_mask(x0, out.clean_latent, out.denoise_mask)[:, :, 0], c1[:, :, 0])


def test_model_strict_load_and_sanitize():
    from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
    cfg = LTXModelConfig(num_layers=1)
    with pytest.raises(ValueError, match="Missing"):
        LTXModel(cfg, {})
    raw = {"model.diffusion_model.transformer_blocks.0.attn1.to_out.0.weight": 1,
           "model.diffusion_model.transformer_blocks.0.ff.net.0.proj.bias": 2,
           "model.diffusion_model.transformer_blocks.0.ff.net.2.weight": 3,
           "model.diffusion_model.adaln_single.emb.timestep_embedder.linear_1.weight": 4,
           "model.diffusion_model.audio_embeddings_connector.x": 5, "vae.decoder.conv_in.weight": 6}
    s = LTXModel.sanitize(raw)
    assert set(s) == {"transformer_blocks.0.attn1.to_out.weight", "transformer_blocks.0.ff.proj_in.bias",
                      "transformer_blocks.0.ff.proj_out.weight", "adaln_single.emb.timestep_embedder.linear1.weight"}
    assert len(LTXModel.expected_keys(LTXModelConfig())) == 15 + 48 * 25
