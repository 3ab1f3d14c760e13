"""LoRA merge (mlx_video/lora.py:18-127): W += strength * (B @ A), product in fp32 on the matrix cores
(ltxk_gemm_bf16 with the SCALE_RES epilogue), result rounded like the reference:
bf16(W + bf16(strength * (B@A))).  One-time work at load (config 5 "merged LoRA")."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterable, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


@dataclass(frozen=True)
class LoraSpec:
    path: Path
    strength: float = 1.0


def _sanitize_lora_prefix(prefix: str) -> str:
    """lora.py:18-33."""
    for p in ("model.diffusion_model.", "diffusion_model."):
        if prefix.startswith(p):
            prefix = prefix[len(p):]
    for a, b in ((".to_out.0.", ".to_out."), (".ff.net.0.proj.", ".ff.proj_in."), (".ff.net.2.", ".ff.proj_out."),
                 (".audio_ff.net.0.proj.", ".audio_ff.proj_in."), (".audio_ff.net.2.", ".audio_ff.proj_out."),
                 (".linear_1.", ".linear1."), (".linear_2.", ".linear2.")):
        prefix = prefix.replace(a, b)
    return prefix


def _iter_lora_pairs(lora_sd: Dict[str, torch.Tensor]) -> Iterable[Tuple[str, str, torch.Tensor, torch.Tensor]]:
    """lora.py:60-74: yield (base_key_raw, base_key_sanitized, A, B)."""
    for key in lora_sd:
        if not key.endswith(".lora_A.weight"):
            continue
        prefix = key[: -len(".lora_A.weight")]
        kb = f"{prefix}.lora_B.weight"
        if kb not in lora_sd:
            continue
        base = f"{prefix}.weight"
        yield base, _sanitize_lora_prefix(base), lora_sd[key], lora_sd[kb]


def _candidate_weight_keys(base_raw: str, base_sanitized: str) -> Tuple[str, ...]:
    """lora.py:77-91."""
    cand = [base_sanitized, base_raw]
    if base_raw.startswith("diffusion_model."):
        cand.append(f"model.{base_raw}")
    if base_sanitized and not base_sanitized.startswith("model."):
        cand += [f"diffusion_model.{base_sanitized}", f"model.diffusion_model.{base_sanitized}"]
    return tuple(dict.fromkeys(cand))


def load_lora_state(path: Path) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(str(path))


_ST_DTYPES = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64}


def load_lora_state_device(path: Path, device) -> Dict[str, torch.Tensor]:
    """The LoRA file's whole data section in ONE host->device copy (a rank-64 LoRA over 8 projections x 48 blocks is 768
    tensors of 0.5 MB: copied one by one from the memory-mapped file they cost 1.2 s, as one 400 MB block 40 ms); the
    tensors are views of that device block, cut by the safetensors header (8-byte length + JSON: dtype, shape, offsets)."""
    import json
    import struct
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        header = json.loads(f.read(n))
        header.pop("__metadata__", None)
        size = max((m["data_offsets"][1] for m in header.values()), default=0)
        pad = (-size) % 8
        host = torch.empty(size + pad, dtype=torch.uint8, pin_memory=torch.device(device).type == "cuda")
        got = f.readinto(memoryview(host.numpy())[:size])
    if got != size:
        raise ValueError(f"{path}: truncated safetensors file ({got} of {size} data bytes)")
    blob = host.to(device, non_blocking=True)
    out = {}
    for k, m in header.items():
        dt = _ST_DTYPES.get(m["dtype"])
        b0, b1 = m["data_offsets"]
        if dt is None or b0 % torch.empty((), dtype=dt).element_size():
            out[k] = None                                   # an exotic / unaligned tensor: fetched from the file on demand
            continue
        out[k] = blob[b0:b1].view(dt).reshape(m["shape"])
    if any(v is None for v in out.values()):
        slow = load_lora_state(path)
        out = {k: (v if v is not None else slow[k].to(device)) for k, v in out.items()}
    return out


def merge_lora_pair(w: torch.Tensor, A: torch.Tensor, B: torch.Tensor, strength: float) -> torch.Tensor:
    """w (out,in), A (r,in), B (out,r) bf16 on the device -> bf16(w + bf16(strength * B@A)).
    The rank axis is zero-padded to a multiple of 64 (the GEMM's K-step); zeros add nothing."""
    Bp, At = _pack_group([A], [B], w.device)
    return ops.gemm(Bp[0], At[0], None, epilogue=ops.EPI_SCALE_RES, resid=w.contiguous(), alpha=float(strength))


def _pack_group(As, Bs, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """G pairs of one shape -> the two GEMM operand stacks on the device, built with ONE host->device copy and ONE
    pad / transpose kernel each (not per pair): Bp (G,out,rp) and A^T (G,in,rp), rank zero-padded to rp = 64*ceil(r/64).
    Deviation from the reference, stated: lora.py:114 multiplies B @ A in fp32 from the stored dtype; here the operands go to
    the matrix cores as bf16 (exact for bf16 LoRA files - the published LTX-2 LoRAs; F16 / F32 files lose operand bits beyond
    bf16's 8 before the fp32-accumulated product, parity unpinned for those) and the delta is rounded once, as in the reference."""
    r = As[0].shape[0]
    rp = (r + 63) // 64 * 64
    A_all = torch.stack([a.to(BF16) for a in As]).to(device, non_blocking=True)          # (G,r,in)
    B_all = torch.stack([b.to(BF16) for b in Bs]).to(device, non_blocking=True)          # (G,out,r)
    if rp == r:
        return B_all.contiguous(), A_all.transpose(1, 2).contiguous()
    Bp = torch.zeros((B_all.shape[0], B_all.shape[1], rp), dtype=BF16, device=device)
    Bp[:, :, :r] = B_all
    At = torch.zeros((A_all.shape[0], A_all.shape[2], rp), dtype=BF16, device=device)
    At[:, :, :r] = A_all.transpose(1, 2)
    return Bp, At


GROUP_MAX = 64      # pairs packed per host->device copy (bounds the staging memory: 64 x (16384+4096) x 128 x 2 B = 336 MB)


def apply_lora_to_weights(weights: Dict[str, torch.Tensor], lora_specs: Iterable[LoraSpec], verbose: bool = False,
                          lora_states: Dict[Path, Dict[str, torch.Tensor]] = None, in_place: bool = False) -> Dict[str, torch.Tensor]:
    """lora.py:94-127.  ``lora_states`` lets callers pass already-loaded LoRA tensors.  Pairs of one (A, B) shape are
    packed together (one copy + one pad/transpose kernel per group), then merged with one EPI_SCALE_RES GEMM launch per
    weight; a weight touched by several LoRAs is merged in spec order, as the reference does.  ``in_place``: the merged
    values are written into the given tensors themselves (the GEMM's output aliases its residual operand: same launches,
    same bits, no second copy of the model) - for callers that own ``weights`` and will not need the un-merged values again."""
    updated = dict(weights)
    for spec in lora_specs:
        sd = (lora_states or {}).get(spec.path)
        if sd is None:
            dev0 = next((w.device for w in updated.values() if torch.is_tensor(w)), torch.device("cpu"))
            sd = load_lora_state_device(spec.path, dev0) if dev0.type == "cuda" else load_lora_state(spec.path)
        applied = skipped = 0
        groups: Dict[tuple, list] = {}
        for base_raw, base_san, A, B in _iter_lora_pairs(sd):
            key = next((k for k in _candidate_weight_keys(base_raw, base_san) if k in updated), None)
            if key is None:
                skipped += 1
                continue
            groups.setdefault((tuple(A.shape), tuple(B.shape), str(updated[key].device)), []).append((key, A, B))
        for items in groups.values():
            for c0 in range(0, len(items), GROUP_MAX):
                chunk = items[c0:c0 + GROUP_MAX]
                dev = updated[chunk[0][0]].device
                Bp, At = _pack_group([a for _, a, _ in chunk], [b for _, _, b in chunk], dev)
                for i, (key, _, _) in enumerate(chunk):
                    w = updated[key]
                    if in_place:
                        if w.dim() != 2 or not w.is_contiguous():
                            raise ValueError(f"in-place LoRA merge needs a contiguous 2-D weight, got {key} {tuple(w.shape)}")
                        ops.gemm(Bp[i], At[i], None, epilogue=ops.EPI_SCALE_RES, resid=w, out=w, alpha=float(spec.strength))
                    else:
                        updated[key] = ops.gemm(Bp[i], At[i], None, epilogue=ops.EPI_SCALE_RES, resid=w.contiguous(),
                                                alpha=float(spec.strength)).reshape(w.shape)
                    applied += 1
        if verbose:
            print(f"[LoRA] {spec.path} applied={applied} skipped={skipped}")
        elif applied == 0:
            print(f"[LoRA] Warning: no weights applied for {spec.path}. Check key mapping.")
    return updated
