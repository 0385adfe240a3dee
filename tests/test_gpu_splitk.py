"""Split-K tail of the grouped GEMM launches (ABI v8, include/moe_gfx950.h): the tiles of an XCD's last partial round are
cut along K, their fp32 partial sums added in K order by a fix-up launch.  Checked through the C ABI: the result with a
workspace equals the result without one up to the fp32 summation order (a few last-bit differences of the 16-bit
outputs, and at least one -- which shows that the shape really was split), is bit-identical from call to call, and
matches an fp32 torch reference on sampled rows, for the three forward epilogues, the 256- and 128-row tiles, ragged
groups, an empty group and a column count that is not a multiple of the tile."""
import pytest
import torch

from moe_llava_amd import _lib
from tests._parity import knob

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _run(lib, epi, x, perm, w, b, offs, E, R, K, N, ws):
    out = torch.zeros(R, N, dtype=torch.bfloat16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.moe_grouped_gemm_fwd(epi, x.data_ptr(), K, perm.data_ptr(), 1, w.data_ptr(), b.data_ptr() if b is not None else None,
                                  offs.data_ptr(), E, E, R, K, N, -1, out.data_ptr(), N,
                                  ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, 0, st)
    assert rc == 0, lib.moe_last_error()
    torch.cuda.synchronize()
    return out


# (epilogue, K, N, rows per group): tile counts chosen so that some XCD's last round is partial with <= 16 tiles
CASES = [
    (1, 2048, 5632, [426, 437, 266, 279], None),        # config 1's fc1 (128-row tiles, SwiGLU): 14 x 44 tiles, 77 per XCD
    (0, 5632, 2048, [426, 437, 266, 279], "128,256"),   # config 1's fc2 (plain) on 128 x 256 tiles: 14 per XCD
    (2, 2560, 10240, [3000, 2500, 0, 1357], None),      # Phi fc1 (gelu_new + bias), 256-row tiles, one empty group
    (0, 10240, 2560, [3000, 2500, 1, 1356], None),      # Phi fc2 (plain + bias), K = 160 slices
    (1, 2048, 5504, [1300, 1200, 1100, 1264], None),    # Qwen fc1: N = 5504 is 43 column tiles of 128
    (0, 1024, 512, [700, 300, 20, 4], "64,128"),        # 9 small tiles per XCD: its only round is partial, 3 pieces each
]


@pytest.mark.parametrize("epi,K,N,rows,tile", CASES)
def test_split_tail_matches_whole_tiles_and_reference(epi, K, N, rows, tile):
    # MOE_SPLITK_MIN=4: pieces of 4 K slices are allowed, so that the short-K shapes are split too (the default, 24, keeps
    # the tail for launches whose tiles are long enough for it to pay)
    with knob("MOE_GEMM_TILE", tile), knob("MOE_SPLITK_MIN", "4"):
        _check(epi, K, N, rows)


def test_default_rule_splits_long_k_only():
    """K = 10240 (config 3's fc2) is split by default, K = 2048 is not."""
    _check(0, 10240, 2560, [3000, 2500, 1, 1356])
    with pytest.raises(AssertionError, match="meant to be split"):
        _check(1, 2048, 5632, [426, 437, 266, 279])


def _check(epi, K, N, rows):
    lib = _lib.load()
    g = torch.Generator().manual_seed(K + N)
    E, R = len(rows), sum(rows)
    S = R // 2 + 1
    x = torch.randn(S, K, generator=g).to(torch.bfloat16).to(DEV)
    NW = 2 * N if epi == 1 else N
    w = (torch.randn(E, NW, K, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
    b = (torch.randn(E, NW, generator=g) * 0.1).to(torch.bfloat16).to(DEV) if epi != 1 else None
    offs = torch.zeros(E + 1, dtype=torch.int32)
    offs[1:] = torch.cumsum(torch.tensor(rows), 0).int()
    offs = offs.to(DEV)
    perm = torch.randint(0, 2 * S, (R,), generator=g).int().to(DEV)      # token * 2 + choice
    ws = torch.empty(lib.moe_splitk_workspace_bytes(), dtype=torch.uint8, device=DEV)
    whole = _run(lib, epi, x, perm, w, b, offs, E, R, K, N, None)
    split = _run(lib, epi, x, perm, w, b, offs, E, R, K, N, ws)
    ws.fill_(0x7f)                                                       # stale workspace contents must not matter
    split2 = _run(lib, epi, x, perm, w, b, offs, E, R, K, N, ws)
    assert torch.equal(split, split2)                                    # fixed summation order: bit-reproducible
    d = (split.float() - whole.float()).abs()
    scale = float(whole.float().abs().max())
    ndiff = int((d > 0).sum())
    print(f"[split-K] epi {epi} K {K} N {N} rows {rows}: {ndiff} of {d.numel()} outputs differ from the whole-tile launch, "
          f"max |d| {float(d.max()):.3e} (scale {scale:.3e})")
    assert ndiff > 0, "this shape was meant to be split"
    assert float(d.max()) <= 2 ** -7 * scale                            # one bf16 ulp at the tensor's scale
    assert ndiff < 0.05 * d.numel()
    with knob("MOE_SPLITK", "0"):                                        # the knob switches the tail off in the library
        off = _run(lib, epi, x, perm, w, b, offs, E, R, K, N, ws)
    assert torch.equal(off, whole)
    # fp32 reference on sampled rows: first / last row of every group, rows around the tile edges, random ones
    o = offs.cpu().tolist()
    pick = set()
    for e in range(E):
        lo, hi = o[e], o[e + 1]
        if hi > lo:
            pick |= {lo, hi - 1, (lo + hi) // 2}
            for t in (128, 256):
                pick |= {r for r in (lo + t - 1, lo + t, hi - (hi - lo) % t - 1, hi - (hi - lo) % t) if lo <= r < hi}
    pick |= set(torch.randint(0, R, (16,), generator=g).tolist())
    for r in sorted(pick):
        e = max(i for i in range(E) if o[i] <= r)
        xr = x[int(perm[r]) >> 1].float()
        acc = w[e].float() @ xr + (b[e].float() if b is not None else 0)
        if epi == 1:
            a16, b16 = acc[:N].bfloat16().float(), acc[N:].bfloat16().float()
            ref = (torch.nn.functional.silu(a16).bfloat16().float() * b16)
        elif epi == 2:
            ref = torch.nn.functional.gelu(acc.bfloat16().float(), approximate="tanh")
        else:
            ref = acc
        err = (split[r].float() - ref).abs()
        bar = 4e-3 * float(ref.abs().max()) + 2e-2 * ref.abs()
        assert bool((err <= bar).all()), (r, e, float(err.max()))
