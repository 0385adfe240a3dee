"""The output bar of the parity tests, in one place (BASELINE.json: "outputs within bf16 tolerance, rtol 1e-2, max-abs
reported").

    |got - want| <= ATOL_OF_SCALE * max|want|  +  RTOL * |want|        for every element

The absolute part exists because an output element is a sum of products that can cancel: its error is set by the
largest terms, not by the element.  Round 1 used 1e-2 * max|want| there, ten times what the data shows (measured
maximum errors are 0.17 - 0.24 % of max|want|: single bf16 ulp flips of h / y elements whose fp32 sums are accumulated
in another order than torch's CPU GEMM); 4e-3 leaves less than a factor of two over what was measured, so a wrong
rounding point (an extra bf16 rounding, fp32 instead of bf16 SiLU, ...) on small elements no longer hides."""
import torch

RTOL = 1e-2
ATOL_OF_SCALE = 4e-3


def report(name, got, want):
    d = (got.float() - want.float()).abs()
    scale = float(want.float().abs().max())
    big = want.float().abs() > 0.05 * scale
    rel_big = float((d[big] / want.float().abs()[big]).max()) if bool(big.any()) else 0.0
    print(f"[parity] {name}: max_abs={float(d.max()):.4e} ref_max={scale:.4e} "
          f"max_abs/ref_max={float(d.max()) / max(scale, 1e-30):.3e} max_rel(|ref|>5% of max)={rel_big:.3e}")
    return float(d.max()), scale


def assert_out_close(name, got, want, atol_of_scale=ATOL_OF_SCALE, rtol=RTOL):
    """Print the error statistics and assert the bar; returns (max_abs, scale)."""
    err, scale = report(name, got, want)
    g, w = got.float(), want.float()
    bad = (g - w).abs() > atol_of_scale * scale + rtol * w.abs()
    assert not bool(bad.any()), (f"{name}: {int(bad.sum())} of {bad.numel()} elements outside "
                                 f"{atol_of_scale:g} * max|ref| + {rtol:g} * |ref| (max_abs {err:.4e}, max|ref| {scale:.4e})")
    return err, scale


class knob:
    """``with knob("MOE_NO_SKINNY", "1"):`` -- set a diagnostic environment knob of the library for the block.  The library
    reads its knobs once, so the change (and the restore) is followed by ``ops.reload_knobs()``."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        import os
        from moe_llava_amd import ops
        self.prev = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = str(self.value)
        ops.reload_knobs()
        return self

    def __exit__(self, *a):
        import os
        from moe_llava_amd import ops
        if self.prev is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.prev
        ops.reload_knobs()
        return False


def edge_tokens(expert_offsets, perm, top_k, bm, n_random=0, seed=0, per_expert=8):
    """Tokens whose compact rows sit where a tiling bug would show: the first and last row of every expert, the rows on
    either side of row-tile boundaries (tiles of ``bm`` rows start at the expert's first row: the first two and last two
    boundaries plus ``per_expert`` random ones), and the first / middle / last row of the expert's last partial tile.
    ``perm[p] = token * top_k + choice`` (the kernels' dispatch index); returns a sorted LongTensor of distinct tokens,
    plus ``n_random`` random compact rows' tokens."""
    import torch
    offs = [int(v) for v in expert_offsets]
    g = torch.Generator().manual_seed(seed)
    rows = set()
    for e in range(len(offs) - 1):
        lo, hi = offs[e], offs[e + 1]
        if hi <= lo:
            continue
        rows |= {lo, hi - 1}
        nb = (hi - lo - 1) // bm                           # boundaries strictly inside the expert's rows
        bs = list(range(1, nb + 1))
        pick = set(bs[:2] + bs[-2:])
        if len(bs) > 4:
            pick |= {bs[int(i)] for i in torch.randint(0, len(bs), (per_expert,), generator=g)}
        for t in pick:
            rows |= {lo + t * bm - 1, lo + t * bm}
        last = lo + nb * bm                                # the last (possibly partial) tile
        rows |= {last, (last + hi - 1) // 2, hi - 1}
    n = offs[-1]
    if n_random and n > 0:
        rows |= {int(i) for i in torch.randint(0, n, (n_random,), generator=g)}
    idx = torch.tensor(sorted(rows), dtype=torch.long)
    return torch.unique(perm.cpu().long()[idx] // top_k)
