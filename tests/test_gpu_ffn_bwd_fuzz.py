"""Seeded fuzz of the grouped expert MLP's backward (`moe_expert_ffn_fwd_train` + `moe_expert_ffn_bwd`)
against torch autograd in fp32 with the forward's 16-bit rounding points: group sizes around the 64-row K slices of the
weight-gradient kernel and the row tiles of the data-gradient launches, empty groups (zero weight gradients), both
activations, biases, the RePaMLP mask, rows read through the gather index."""
import random

import pytest
import torch

from moe_llava_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
ROWS = [0, 0, 1, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 400]


def _close(tag, got, want, tol=2e-2):
    got, want = got.detach().float(), want.detach().float()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= tol * scale + 1e-5, f"{tag}: max err {err:.3e} at scale {scale:.3e}"
    if float(want.norm()) > 0:
        nrm = float((got - want).norm() / want.norm())
        assert nrm <= tol, f"{tag}: relative l2 error {nrm:.3e}"


def test_expert_ffn_backward_fuzz_vs_autograd():
    rnd = random.Random(4242)
    for case in range(30):
        act = rnd.choice([ops.ACT_SWIGLU, ops.ACT_GELU_NEW])
        M = rnd.choice([64, 128, 256, 512])
        I = rnd.choice([64, 128, 320, 512])
        E = rnd.choice([1, 2, 4, 5])
        rows = [rnd.choice(ROWS) for _ in range(E)]
        if sum(rows) == 0:
            rows[-1] = 70
        R = sum(rows)
        bias = rnd.random() < 0.5
        n_gated = rnd.choice([-1, -1, I // 2]) if act == ops.ACT_SWIGLU else -1
        gathered = rnd.random() < 0.6
        g = torch.Generator().manual_seed(300 + case)
        n1 = 2 * I if act == ops.ACT_SWIGLU else I
        S = (R + 1) // 2 + 2 if gathered else R
        x = torch.randn(S, M, generator=g).to(torch.bfloat16).to(DEV)
        w1 = (torch.randn(E, n1, M, generator=g) * (1.2 / M ** 0.5)).to(torch.bfloat16).to(DEV)
        w2 = (torch.randn(E, M, I, generator=g) * (1.2 / I ** 0.5)).to(torch.bfloat16).to(DEV)
        b1 = (torch.randn(E, n1, generator=g) * 0.2).to(torch.bfloat16).to(DEV) if bias else None
        b2 = (torch.randn(E, M, generator=g) * 0.2).to(torch.bfloat16).to(DEV) if bias else None
        offs = torch.zeros(E + 1, dtype=torch.int32)
        offs[1:] = torch.cumsum(torch.tensor(rows), 0).int()
        offs = offs.to(DEV)
        perm = torch.randint(0, 2 * S, (R,), generator=g).int().to(DEV) if gathered else None
        dy = torch.randn(R, M, generator=g).to(torch.bfloat16).to(DEV)
        tag = f"case {case}: act {act} M {M} I {I} rows {rows} bias {bias} n_gated {n_gated} gathered {gathered}"

        y, ab, h = ops.expert_ffn_train(act, x, perm, 1 if gathered else 0, w1, b1, w2, b2, offs, R, n_gated=n_gated)
        da, dw1, db1, dw2, db2 = ops.expert_ffn_bwd(act, x, perm, 1 if gathered else 0, w1, w2, ab, h, dy, offs, R,
                                                    bias, bias, n_gated=n_gated)
        torch.cuda.synchronize()

        o = offs.cpu().tolist()
        for e in range(E):
            lo, hi = o[e], o[e + 1]
            if hi == lo:
                assert float(dw1[e].float().abs().max()) == 0 and float(dw2[e].float().abs().max()) == 0, tag + ": idle expert"
                if bias:
                    assert float(db1[e].abs().max()) == 0 and float(db2[e].abs().max()) == 0, tag + ": idle expert bias"
                continue
            xr = (x[(perm[lo:hi] >> 1).long()] if gathered else x[lo:hi]).float().requires_grad_(True)
            W1, W2 = w1[e].float().requires_grad_(True), w2[e].float().requires_grad_(True)
            B1 = b1[e].float().requires_grad_(True) if bias else None
            B2 = b2[e].float().requires_grad_(True) if bias else None
            pre = xr @ W1.t() + (B1 if bias else 0)
            pre = pre + (pre.bfloat16().float() - pre).detach()          # 16-bit output of the Linear, identity gradient
            if act == ops.ACT_SWIGLU:
                a_, b_ = pre[:, :I], pre[:, I:]
                sg = torch.nn.functional.silu(a_)
                if 0 <= n_gated < I:
                    sg = torch.cat([sg[:, :n_gated], torch.ones_like(sg[:, n_gated:])], 1)
                hh = sg * b_
            else:
                hh = torch.nn.functional.gelu(pre, approximate="tanh")
            hh = hh + (hh.bfloat16().float() - hh).detach()
            yy = hh @ W2.t() + (B2 if bias else 0)
            _close(tag + f" y[{e}]", y[lo:hi], yy, tol=1.5e-2)
            (yy * dy[lo:hi].float()).sum().backward()
            _close(tag + f" da[{e}]", da[lo:hi], xr.grad)
            _close(tag + f" dw1[{e}]", dw1[e], W1.grad)
            _close(tag + f" dw2[{e}]", dw2[e], W2.grad)
            if bias:
                _close(tag + f" db1[{e}]", db1[e], B1.grad)
                _close(tag + f" db2[{e}]", db2[e], B2.grad)
