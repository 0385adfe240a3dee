"""The entry point that allocates nothing: moe_routing_top2_fwd_noalloc (scratch in a device array built into the
library) against moe_routing_top2_fwd (caller-provided workspace), through the C ABI, bit for bit, on two calls in a
row, the second with a smaller input.  Collected last on purpose (file name): a failure here must not hide other tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
_FIELDS = (("expert_idx", torch.int32, 2), ("slot", torch.int32, 2), ("weight", torch.float32, 2))


def _outputs(S, E, max_rows):
    o = {n: torch.full((S, k), -7, dtype=dt, device=DEV) for n, dt, k in _FIELDS}
    o["exp_counts"] = torch.full((E,), -7, dtype=torch.int64, device=DEV)
    o["kept_counts"] = torch.full((E,), -7, dtype=torch.int32, device=DEV)
    o["expert_offsets"] = torch.full((E + 1,), -7, dtype=torch.int32, device=DEV)
    o["perm"] = torch.full((max(max_rows, 1),), -7, dtype=torch.int32, device=DEV)
    o["l_aux"] = torch.full((1,), -7.0, dtype=torch.float32, device=DEV)
    return o


def _args(o):
    return [o[n].data_ptr() for n in ("expert_idx", "slot", "weight", "exp_counts", "kept_counts", "expert_offsets",
                                      "perm", "l_aux")]


def _both(lib, logits, noise, seed, S, E, cap, first_on_device=False):
    st = torch.cuda.current_stream().cuda_stream
    max_rows = min(2 * S, E * cap)
    a, b = _outputs(S, E, max_rows), _outputs(S, E, max_rows)
    nws = lib.moe_routing_workspace_bytes(S, E)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=DEV)
    npt = noise.data_ptr() if noise is not None else None
    rc = lib.moe_routing_top2_fwd(logits.data_ptr(), npt, seed, S, E, cap, 1, *_args(a), ws.data_ptr(), nws, st)
    assert rc == 0, lib.moe_last_error()
    if first_on_device:
        # the first call on a device does the one-off hipGetSymbolAddress lookup; the no-allocation claim is about every
        # call after it
        rc = lib.moe_routing_top2_fwd_noalloc(logits.data_ptr(), npt, seed, S, E, cap, 1, *_args(b), st)
        assert rc == 0, lib.moe_last_error()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before = torch.cuda.mem_get_info()[0]               # hipMemGetInfo: also sees a hipMalloc torch's allocator does not
    rc = lib.moe_routing_top2_fwd_noalloc(logits.data_ptr(), npt, seed, S, E, cap, 1, *_args(b), st)
    assert rc == 0, lib.moe_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory changed during the no-allocation entry point"
    torch.cuda.synchronize()
    n = int(a["expert_offsets"][-1])
    for name in a:
        x, y = (a[name][:n], b[name][:n]) if name == "perm" else (a[name], b[name])
        assert torch.equal(x, y), name
    return a


def test_no_alloc_routing_matches_workspace_entry_point_twice():
    from moe_llava_amd import _lib
    from moe_llava_amd.sharded_moe import capacity_of
    from oracle import ds_moe_index as ix
    lib = _lib.load()
    E = 8
    g = torch.Generator().manual_seed(5)
    big = torch.randn(40000, E, generator=g)
    big[:20000, 0] += 2.0                                              # capacity drops
    noise = torch.rand(40000, E, generator=g)
    # two calls in a row on the three-kernel path (S > 4096: the path that uses the scratch), the second smaller, then
    # a single-launch size; the shared array keeps the previous call's contents in between
    for S, with_noise, seed in ((40000, True, 0), (9001, False, 77), (700, False, 0)):
        lg = big[:S].contiguous().to(DEV)
        nz = noise[:S].contiguous().to(DEV) if with_noise else None
        cap = capacity_of(S, E, 2.0, 0)
        got = _both(lib, lg, nz, seed, S, E, cap, first_on_device=(S == 40000))
        if seed == 0:                                                  # and against the oracle
            want = ix.route_top2(big[:S].clone(), cap, noise[:S] if with_noise else None, True)
            assert torch.equal(got["slot"].cpu(), want["slot"]) and torch.equal(got["exp_counts"].cpu(), want["exp_counts"])
    # beyond the built-in array: a clear error, nothing launched
    S_big = 3_000_000
    assert lib.moe_routing_workspace_bytes(S_big, E) > (1 << 20)
    rc = lib.moe_routing_top2_fwd_noalloc(16, None, 0, S_big, E, 10, 1, 16, 16, 16, 16, 16, 16, 16, 16, None)
    assert rc == -4 and b"built-in array" in lib.moe_last_error()
