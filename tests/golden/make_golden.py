#!/usr/bin/env python3
"""Generates the golden vectors of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

PARITY UNPINNED: the reference's algorithm for this path lives in deepspeed==0.17.5, which is neither vendored under
/root/reference nor installable offline, and the reference tree holds no fixtures for it.  These vectors
are therefore produced by THIS repository's literal dense-mask restatement (oracle/ds_moe_literal.py) on the CPU, in
this container, with fixed seeds -- they pin the oracle and the kernels against silent drift, not against upstream.
The one exception is known_answer.json, which is hand-derived in SURVEY.md 8a'' and only re-checked here.

Files written:
  known_answer.json        S=4, E=3 hand-derived top-2 case
  top2_S704_E4.npz         config-1 routing shape: logits, Gumbel noise, dense-form results (deterministic and noisy)
  top2_S1000_E8_drop.npz   8 experts, capacity factor 0.5 (heavy drops), min_capacity 4
  top1_S512_E4.npz         top-1 with RTS uniforms, used_token, RSample noise
  block_stablelm.npz       whole block, StableLM experts (M 128, I 320, E 4, S 333): x, weights, fp32 logits, outputs
  block_phi.npz            whole block, Phi experts with biases (M 128, I 256, E 4, S 260)
  block_stablelm_fp16.npz  the StableLM block in fp16 (the reference's inference dtype, moellava/model/builder.py:70), gate
                           weight in fp16 too as after model.to(torch.float16); written by `make_golden.py --fp16-only`
                           (round 2) so that the round-1 files keep their bytes
bf16 / fp16 tensors are stored as their uint16 bit patterns."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ds_moe_literal as lit  # noqa: E402
from oracle.expert_mlps import PhiMLPRef, StableLMMLPRef  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def dense_to_index(combine: torch.Tensor):
    """[S,E,C] combine weights -> per token list of (expert, slot, weight), sorted by expert."""
    S, E, C = combine.shape
    idx = np.full((S, 2), -1, np.int32)
    slot = np.full((S, 2), -1, np.int32)
    w = np.zeros((S, 2), np.float32)
    nz = torch.nonzero(combine)
    fill = np.zeros(S, np.int64)
    for s, e, c in nz.tolist():
        j = fill[s]
        idx[s, j], slot[s, j], w[s, j] = e, c, float(combine[s, e, c])
        fill[s] += 1
    return idx, slot, w


def top2_case(S, E, cf, min_cap, seed, noise_seed):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(S, E, generator=g)
    logits[: S // 2, 0] += 0.7
    out = {"logits": logits.numpy(), "capacity_factor": np.float32(cf), "min_capacity": np.int32(min_cap)}
    for tag, noise in (("det", None), ("noisy", lit.gumbel_noise((S, E), noise_seed))):
        l_aux, combine, dispatch, counts = lit.top2gating_dense(logits.clone(), cf, min_cap, True, noise)
        idx, slot, w = dense_to_index(combine)
        out.update({f"{tag}_expert": idx, f"{tag}_slot": slot, f"{tag}_weight": w, f"{tag}_l_aux": np.float32(l_aux),
                    f"{tag}_exp_counts": counts.numpy().astype(np.int64),
                    f"{tag}_capacity": np.int32(combine.shape[2])})
        if noise is not None:
            out["noise"] = noise.numpy()
    return out


def top1_case():
    S, E = 512, 4
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(S, E, generator=g)
    logits[:, 2] += 1.0
    u = torch.rand(S, E, generator=g)
    used = (torch.rand(S, generator=g) > 0.25).long()
    noise = lit.gumbel_noise((S, E), 5)
    out = {"logits": logits.numpy(), "rts_uniform": u.numpy(), "used_token": used.numpy().astype(np.int32),
           "noise": noise.numpy(), "capacity_factor": np.float32(0.75), "min_capacity": np.int32(0)}
    for tag, kw in (("rts", dict(used_token=used, use_rts=True, rts_uniform=u)),
                    ("first", dict(used_token=None, use_rts=False)),
                    ("rsample", dict(used_token=used, use_rts=True, rts_uniform=u, noise=noise))):
        l_aux, combine, dispatch, counts = lit.top1gating_dense(logits.clone(), 0.75, 0, drop_tokens=True, **kw)
        idx, slot, w = dense_to_index(combine)
        out.update({f"{tag}_expert": idx[:, 0], f"{tag}_slot": slot[:, 0], f"{tag}_weight": w[:, 0],
                    f"{tag}_l_aux": np.float32(l_aux), f"{tag}_exp_counts": counts.numpy().astype(np.int64)})
    return out


def block_case(kind, M, I, E, B, T, cf, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    proto = (StableLMMLPRef(M, I) if kind == "stablelm" else PhiMLPRef(M, I)).to(dtype)
    moe = lit.MoERef(M, proto, num_experts=E, k=2, capacity_factor=cf, eval_capacity_factor=2.0, min_capacity=0,
                     top2_2nd_expert_sampling=False)
    for p in moe.deepspeed_moe.experts.parameters():
        p.data = (torch.randn(p.shape, generator=g) * 0.05).to(dtype)
    moe.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M, generator=g) * 0.1
    if dtype == torch.float16:   # model.to(torch.float16) casts the gate too (builder.py:475); values stay fp16-exact
        moe.deepspeed_moe.gate.wg.weight.data = moe.deepspeed_moe.gate.wg.weight.data.to(dtype).float()
    x = torch.randn(B, T, M, generator=g).to(dtype)
    moe.train()
    with torch.no_grad():
        out, l_aux, counts = moe(x)
    res = {"x": bits(x), "out": bits(out), "l_aux": np.float32(l_aux), "exp_counts": counts.numpy().astype(np.int64),
           "logits": moe.deepspeed_moe.gate.last_logits.numpy(), "capacity_factor": np.float32(cf),
           "wg": moe.deepspeed_moe.gate.wg.weight.detach().numpy()}
    for name, p in moe.deepspeed_moe.experts.named_parameters():
        res["param:" + name] = bits(p.data)
    return res


def main():
    if "--fp16-only" in sys.argv:
        np.savez_compressed(os.path.join(OUT, "block_stablelm_fp16.npz"),
                            **block_case("stablelm", 128, 320, 4, 3, 111, 1.5, 23, torch.float16))
        print("block_stablelm_fp16.npz", os.path.getsize(os.path.join(OUT, "block_stablelm_fp16.npz")))
        return
    # hand-derived case of SURVEY 8a'' (l_aux there is printed as 1.493633; exact fp32 arithmetic gives 1.4936405)
    logits = torch.tensor([[3., 2., 1.], [3., 1., 2.], [2., 3., 1.], [3., 2., 1.]])
    l_aux, combine, dispatch, counts = lit.top2gating_dense(logits.clone(), 0.75, 0, True, None)
    idx, slot, w = dense_to_index(combine)
    ka = {"logits": logits.tolist(), "capacity_factor": 0.75, "min_capacity": 0, "capacity": 2,
          "exp_counts": [4, 3, 1], "kept_per_expert": [2, 2, 1],
          "token_assignments": [[[0, 0, 0.731059], [1, 1, 0.268941]], [[0, 1, 0.731059], [2, 0, 0.268941]],
                                [[1, 0, 1.0]], []],
          "l_aux": 1.4936405}
    assert counts.tolist() == ka["exp_counts"] and abs(float(l_aux) - ka["l_aux"]) < 1e-6
    for s, want in enumerate(ka["token_assignments"]):
        got = [(int(idx[s, j]), int(slot[s, j]), float(w[s, j])) for j in range(2) if idx[s, j] >= 0]
        assert len(got) == len(want) and all(a[0] == b[0] and a[1] == b[1] and abs(a[2] - b[2]) < 1e-5
                                              for a, b in zip(got, want)), (s, got, want)
    json.dump(ka, open(os.path.join(OUT, "known_answer.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(OUT, "top2_S704_E4.npz"), **top2_case(704, 4, 1.5, 0, 11, 99))
    np.savez_compressed(os.path.join(OUT, "top2_S1000_E8_drop.npz"), **top2_case(1000, 8, 0.5, 4, 12, 98))
    np.savez_compressed(os.path.join(OUT, "top1_S512_E4.npz"), **top1_case())
    np.savez_compressed(os.path.join(OUT, "block_stablelm.npz"), **block_case("stablelm", 128, 320, 4, 3, 111, 1.5, 21))
    np.savez_compressed(os.path.join(OUT, "block_phi.npz"), **block_case("phi", 128, 256, 4, 2, 130, 1.0, 22))
    np.savez_compressed(os.path.join(OUT, "block_stablelm_fp16.npz"),
                        **block_case("stablelm", 128, 320, 4, 3, 111, 1.5, 23, torch.float16))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
