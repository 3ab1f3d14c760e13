"""One-off randomized sweep of the attention launch forms (A/B build): for random (B, H, Tq, Tk), with and without the fused query
preparation, the mixed 192 / 128-row grid, 192-row tiles only and the 128-row kernel without its key-split tail must agree BIT FOR
BIT, and the default launch (which may take the key-split tail on small grids) must stay within 5e-4 rel-L2 of them (the stated tolerance of the key-split tail).
  python scripts/fuzz_attn_forms.py [cases] [seed]"""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_video_amd import _lib, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
with _lib.use_library(_lib.AB_LIB_PATH):
    for ci in range(cases):
        H = rnd.choice([1, 3, 8, 16, 32])
        B = rnd.choice([1, 2, 3, 4])
        Tq = rnd.choice([rnd.randint(1, 200), rnd.randint(129, 800), rnd.randint(800, 2400)])
        Tk = rnd.choice([rnd.randint(1, 130), rnd.randint(64, 700), 1024, rnd.randint(700, 1500)])
        D = H * 128
        g = torch.Generator(device=dev).manual_seed(ci)
        q = torch.randn((B * Tq, D), generator=g, device=dev).to(BF)
        k = torch.randn((B * Tk, D), generator=g, device=dev).to(BF)
        vt = torch.randn((B, D, (Tk + 63) // 64 * 64), generator=g, device=dev).to(BF)
        prep = H >= 4 and rnd.random() < 0.5           # (the fused preparation takes H*128/64 partial sums, a multiple of 8)
        kw = {}
        if prep:
            ss = (q.float() ** 2).reshape(B * Tq, D // 64, 64).sum(-1).contiguous()
            w = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(BF)
            kw = dict(q_sumsq=ss, q_norm_weight=w, eps=1e-6)
            if rnd.random() < 0.7:
                kw.update(cos=torch.randn((H, Tq, 64), generator=g, device=dev), sin=torch.randn((H, Tq, 64), generator=g, device=dev))
        outs = {}
        for form, env, ts in (("two", "2", False), ("mixed", "0", False), ("three", "3", False), ("default", "0", True)):
            os.environ["LTXK_FA_QB"] = env
            o = torch.full((B * Tq + 1, D), 7.0, dtype=BF, device=dev)
            ops.flash_attn(q, k, vt, o[:B * Tq], B, H, Tq, Tk, 1.0 / math.sqrt(128), tail_split=ts, **kw)
            torch.cuda.synchronize()
            if not bool((o[B * Tq:] == 7.0).all()):
                bad += 1; print("WROTE PAST THE END", ci, form, flush=True)
            outs[form] = o[:B * Tq]
        tag = f"case {ci}: B={B} H={H} Tq={Tq} Tk={Tk} prep={prep} rope={'cos' in kw}"
        if not (torch.equal(outs["mixed"], outs["two"]) and torch.equal(outs["three"], outs["two"])):
            bad += 1; print("FORM MISMATCH", tag, int((outs['mixed'] != outs['two']).sum()), int((outs['three'] != outs['two']).sum()), flush=True)
        rel = float((outs["default"].float() - outs["two"].float()).norm() / outs["two"].float().norm())
        if not rel <= 5e-4:
            bad += 1; print("DEFAULT FAR FROM THE NO-SPLIT FORM", tag, rel, flush=True)
        if ci % 25 == 24:
            print(f"[progress] {ci + 1} cases, {bad} failures", flush=True)
print(f"done: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
