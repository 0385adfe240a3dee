"""Seeded fuzz of `moe_grouped_gemm_fwd` through the C ABI (include/moe_gfx950.h) against fp32 torch with the reference's
16-bit rounding points: group sizes on and around every tile height (32 / 64 / 128 / 256 rows) incl. empty groups, K from one
slice to 2048, column counts that are not a multiple of any tile, the three forward epilogues, biases, the RePaMLP mask,
rows read directly or through a gather index, an output leading dimension wider than N.  Every output element is compared;
rows past the last group and the padding columns must stay untouched.  Launches of at most 64 rows take the
weight-streaming kernel, the rest the tiled one: both are covered by the draw."""
import random

import pytest
import torch

from moe_llava_amd import _lib

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SENT = 0x7b7b  # bf16 bit pattern written where nothing may be stored

ROWS = [0, 0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 700]


def _reference(epi, xr, w, b, N, n_gated):
    acc = xr.float() @ w.float().t()
    if b is not None:
        acc = acc + b.float()
    if epi == 1:
        a16, b16 = acc[:, :N].bfloat16().float(), acc[:, N:].bfloat16().float()
        sg = torch.nn.functional.silu(a16).bfloat16().float()
        if 0 <= n_gated < N:
            sg[:, n_gated:] = 1.0                      # RePaMLP: channels past the mask bypass the SiLU
        return sg * b16
    if epi == 2:
        return torch.nn.functional.gelu(acc.bfloat16().float(), approximate="tanh")
    return acc


def test_grouped_gemm_fuzz_vs_fp32_reference():
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    rnd = random.Random(31337)
    for case in range(60):
        epi = rnd.choice([0, 1, 2])
        K = rnd.choice([64, 128, 192, 512, 1024, 2048])
        N = rnd.choice([4, 60, 64, 100, 128, 256, 260, 516, 1000, 2048])
        E = rnd.choice([1, 2, 4, 4, 5, 8])
        if rnd.random() < 0.25:                         # decode-size launch: at most 64 rows in all
            rows = [0] * E
            for _ in range(rnd.randint(1, 64)):
                rows[rnd.randrange(E)] += 1
        else:
            rows = [rnd.choice(ROWS) for _ in range(E)]
        R = sum(rows)
        if R == 0:
            rows[0] = 1
            R = 1
        gathered = rnd.random() < 0.6
        shift = rnd.choice([0, 1]) if gathered else 0
        has_bias = rnd.random() < 0.5
        n_gated = rnd.choice([-1, -1, N // 2, 4]) if epi == 1 else -1
        pad = rnd.choice([0, 4, 8])
        g = torch.Generator().manual_seed(900 + case)
        S = max(1, (R >> shift) + 3) if gathered else R
        x = torch.randn(S, K, generator=g).to(torch.bfloat16).to(DEV)
        NW = 2 * N if epi == 1 else N
        w = (torch.randn(E, NW, K, generator=g) * (1.5 / K ** 0.5)).to(torch.bfloat16).to(DEV)
        b = (torch.randn(E, NW, generator=g) * 0.2).to(torch.bfloat16).to(DEV) if has_bias else None
        offs = torch.zeros(E + 1, dtype=torch.int32)
        offs[1:] = torch.cumsum(torch.tensor(rows), 0).int()
        offs = offs.to(DEV)
        perm = torch.randint(0, S << shift, (R,), generator=g).int().to(DEV) if gathered else None
        out = torch.full((R + 2, N + pad), SENT, dtype=torch.int16, device=DEV).view(torch.bfloat16)
        rc = lib.moe_grouped_gemm_fwd(epi, x.data_ptr(), K, perm.data_ptr() if gathered else None, shift, w.data_ptr(),
                                      b.data_ptr() if b is not None else None, offs.data_ptr(), E, E, R, K, N, n_gated,
                                      out.data_ptr(), N + pad, None, 0, 0, st)
        tag = f"case {case}: epi {epi} K {K} N {N} rows {rows} gathered {gathered} shift {shift} bias {has_bias} n_gated {n_gated} pad {pad}"
        assert rc == 0, tag + ": " + lib.moe_last_error().decode()
        torch.cuda.synchronize()
        assert bool((out[R:].view(torch.int16) == SENT).all()), tag + ": wrote past the last row"
        if pad:
            assert bool((out[:, N:].view(torch.int16) == SENT).all()), tag + ": wrote into the padding columns"
        o = offs.cpu().tolist()
        for e in range(E):
            lo, hi = o[e], o[e + 1]
            if hi == lo:
                continue
            xr = x[(perm[lo:hi] >> shift).long()] if gathered else x[lo:hi]
            ref = _reference(epi, xr, w[e], b[e] if b is not None else None, N, n_gated)
            got = out[lo:hi, :N].float()
            err = (got - ref).abs()
            bar = 4e-3 * float(ref.abs().max()) + 2e-2 * ref.abs()
            bad = err > bar
            assert not bool(bad.any()), (tag + f": expert {e}: {int(bad.sum())} outputs off, max err {float(err.max()):.3e} "
                                         f"at {[int(v) for v in torch.nonzero(bad)[0]]} (rows {lo}..{hi})")
