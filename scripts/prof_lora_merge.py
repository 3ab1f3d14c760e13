"""Where the stage-2 LoRA transformer build spends its time (config 3): file load, operand packing, merge GEMMs, panel packing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathlib import Path
from mlx_video_amd.ltx_model import LTXModel, LTXModelConfig
from mlx_video_amd import lora as L
dev = torch.device("cuda:0"); BF = torch.bfloat16
layers = int(os.environ.get("LAYERS", "48"))
cfg = LTXModelConfig(num_layers=layers)
Wt = LTXModel.random_weights(cfg, dev)
from safetensors.torch import save_file
gl = torch.Generator().manual_seed(5); sd = {}
for i in range(layers):
    for raw in ("attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0"):
        sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_A.weight"] = (torch.randn(64, 4096, generator=gl) * 0.02).to(BF)
        sd[f"diffusion_model.transformer_blocks.{i}.{raw}.lora_B.weight"] = (torch.randn(4096, 64, generator=gl) * 0.02).to(BF)
save_file(sd, "/tmp/l.safetensors")
def T(f, name):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{name:28s} {1e3*(time.perf_counter()-t):8.1f} ms", flush=True); return r
for rep in range(2):
    st = T(lambda: L.load_lora_state_device(Path("/tmp/l.safetensors"), dev), "load_lora_state_device")
    merged = T(lambda: L.apply_lora_to_weights(Wt, [L.LoraSpec(Path("/tmp/l.safetensors"), 0.8)], lora_states={Path("/tmp/l.safetensors"): st}), "apply (tensors on device)")
    m0 = T(lambda: L.apply_lora_to_weights(Wt, [L.LoraSpec(Path("/tmp/l.safetensors"), 0.8)]), "apply_lora_to_weights (file)")
    assert all(torch.equal(merged[k], m0[k]) for k in merged)
    del m0
    m = T(lambda: LTXModel(cfg, merged), "LTXModel(...) panel packing")
    del merged, m
