"""Runs the single-GPU configurations of BASELINE.json end to end with synthetic (random-init) weights and
random text embeddings, through generate_video, and prints one JSON line per config with the phase times.
Config 4 (512x512x97, 4 seeds over 8 GPUs) runs here as ONE seed on one GPU: same per-GPU work and the temporal-tiled
VAE decode; the multi-GPU form is the driver's scaling run of bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd.generate import PipelineType, generate_video
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
from mlx_video_amd.upsampler import LatentUpsampler
from mlx_video_amd.video_vae import LTX2VideoDecoder, VideoEncoder, random_decoder_weights
from mlx_video_amd.weights import random_upsampler_weights

dev = torch.device("cuda:0")
BF = torch.bfloat16
layers = int(os.environ.get("LAYERS", "48"))
cfg = LTXModelConfig(num_layers=layers)
Wt = LTXModel.random_weights(cfg, dev)
tr = LTXModel(cfg, Wt)


def lora_file(path, rank, seed):
    """A synthetic rank-`rank` LoRA over every attention projection of every block (PyTorch LTX-2 key naming, lora.py:18-33)."""
    from safetensors.torch import save_file
    gl = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(layers):
        for raw in ("attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0"):
            sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_A.weight"] = (torch.randn(rank, 4096, generator=gl) * 0.02).to(BF)
            sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_B.weight"] = (torch.randn(4096, rank, generator=gl) * 0.02).to(BF)
    save_file(sd, path)
    return path


lora64 = lora_file("/tmp/ltxk_synth_lora64.safetensors", 64, 5)
dec = LTX2VideoDecoder(random_decoder_weights(dev))
ups = LatentUpsampler(random_upsampler_weights(dev))
g = torch.Generator(device=dev).manual_seed(1)
emb = torch.randn((1, 1024, 3840), generator=g, device=dev).to(BF)
neg = torch.randn((1, 1024, 3840), generator=g, device=dev).to(BF)


def enc_weights():
    import math
    W = {}
    def conv(name, o, i):
        W[f"{name}.weight"] = (torch.randn((o, 3, 3, 3, i), generator=g, device=dev) / math.sqrt(27 * i)).to(BF)
        W[f"{name}.bias"] = torch.zeros(o, dtype=BF, device=dev)
    from mlx_video_amd.video_vae import ENC_BLOCKS
    conv("conv_in", 128, 48)
    ch = 128
    for bi, (kind, arg) in enumerate(ENC_BLOCKS):
        if kind == "res_x":
            for li in range(arg):
                conv(f"down_blocks.{bi}.res_blocks.{li}.conv1", ch, ch); conv(f"down_blocks.{bi}.res_blocks.{li}.conv2", ch, ch)
        else:
            conv(f"down_blocks.{bi}.conv", ch * 2 // (arg[0] * arg[1] * arg[2]), ch); ch *= 2
    conv("conv_out", 129, ch)
    W["per_channel_statistics.mean"] = torch.zeros(128, dtype=BF, device=dev)
    W["per_channel_statistics.std"] = torch.ones(128, dtype=BF, device=dev)
    return W


enc = VideoEncoder(enc_weights())
runs = [
    ("config1 dev 128x128x9 1 step", dict(pipeline=PipelineType.DEV, height=128, width=128, num_frames=9, num_inference_steps=1)),
    ("config2 dev 512x512x33 40 steps CFG4", dict(pipeline=PipelineType.DEV, height=512, width=512, num_frames=33, num_inference_steps=40)),
    ("config4 (one seed) dev 512x512x97 40 steps CFG4, temporal-tiled decode", dict(pipeline=PipelineType.DEV, height=512, width=512, num_frames=97, num_inference_steps=40, tiling="temporal")),
    ("config5 ic_lora 768x768x65 video-cond, LoRA (rank 64) merged IN PLACE into a fresh model", dict(pipeline=PipelineType.IC_LORA, height=768, width=768, num_frames=65, stage1_steps=8, stage2_steps=3,
                                                 loras=[(lora64, 1.0)], lora_in_place=True, fresh_model=True,
                                                 video_conditionings=[((torch.rand((1, 3, 65, 768, 768), generator=g, device=dev) * 2 - 1).to(BF), 0, 1.0)])),
    ("config3 distilled 768x768x65 two-stage + 2x upsampler, distilled LoRA (rank 64) merged IN PLACE into the (fresh) stage-1 model for stage 2", dict(pipeline=PipelineType.DISTILLED, height=768, width=768, num_frames=65, stage1_steps=8, stage2_steps=3,
                                                 distilled_loras=[(lora64, 0.8)], lora_in_place=True, fresh_model=True)),
]
for name, kw in runs:
    if kw.pop("fresh_model", False):          # an in-place merge consumes the model: these configs get their own copy of the base weights
        del tr, Wt
        torch.cuda.empty_cache()
        Wt = LTXModel.random_weights(cfg, dev)
        tr = LTXModel(cfg, Wt)
        torch.cuda.synchronize()
    pj = f"/tmp/prof_{abs(hash(name))}.json"
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fr = generate_video(prompt="x", transformer=tr, transformer_weights=Wt, transformer_config=cfg, vae_decoder=dec, vae_encoder=enc, upsampler=ups, prompt_embeds=emb,
                        negative_prompt_embeds=neg, cfg_scale=4.0, compile_step=True, cfg_batch=True, device=dev, seed=7,
                        profile_json_path=pj, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ph = json.load(open(pj))["phases_s"]
    # achieved PFLOP/s of each denoise phase: SURVEY 8d's algorithmic FLOPs (bench.dit_forward_flops) x forwards / phase time
    from bench import dit_forward_flops
    H32, W32, Fl = kw["height"] // 32, kw["width"] // 32, 1 + (kw["num_frames"] - 1) // 8
    pf = {}
    if "dev_denoise" in ph:
        pf["dev_denoise"] = kw["num_inference_steps"] * dit_forward_flops(Fl * H32 * W32, B=2, L=layers) / ph["dev_denoise"] / 1e15
    if "stage1_denoise" in ph:
        pf["stage1_denoise"] = kw["stage1_steps"] * dit_forward_flops(Fl * (H32 // 2) * (W32 // 2), B=1, L=layers) / ph["stage1_denoise"] / 1e15
        pf["stage2_denoise"] = kw["stage2_steps"] * dit_forward_flops(Fl * H32 * W32, B=1, L=layers) / ph["stage2_denoise"] / 1e15
    print(json.dumps({"config": name, "frames": list(fr.shape), "wall_s": round(dt, 3), "phases_s": {k: round(v, 4) for k, v in ph.items()}, "achieved_PFLOPs": {k: round(v, 3) for k, v in pf.items()},
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}), flush=True)
