"""Worker for the expert-parallel tests (spawned with torch.multiprocessing; not collected by pytest).

mode "cpu": world_size-2 gloo rehearsal of MOELayer._expert_parallel on the CPU.  The HIP ops cannot run here,
so the test swaps ``moe_llava_amd.ops`` entry points for oracle-backed stand-ins (test infrastructure only) and
checks the exchange logic: split sizes, (source, expert) group layout, reverse all-to-all, combine.
mode "gpu": the real kernels, two ranks sharing cuda:0, gloo group (payload staged through the host, because
RCCL refuses two ranks on one device); output must equal the ep_size=1 block bit for bit.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _install_cpu_stand_ins():
    """Oracle-backed replacements of the HIP ops, for the CPU rehearsal only."""
    import torch.nn.functional as F
    from moe_llava_amd import ops
    from oracle import ds_moe_index as ix

    def gate_logits(x, wg):
        return F.linear(x.float(), wg.float())

    def routing_top2(logits, capacity, noise=None, drop_tokens=True, noise_seed=0):
        S, E = logits.shape
        r = ix.route_top2(logits, capacity, noise, drop_tokens)
        offs, perm = ix.dispatch_order(r, E)
        max_rows = min(2 * S, E * capacity) if drop_tokens else 2 * S
        p = torch.zeros(max(max_rows, 1), dtype=torch.int32)
        p[: perm.numel()] = perm
        return ops.Routing(r["expert_idx"], r["slot"], r["weight"], r["exp_counts"], r["kept_counts"], offs, p,
                           r["l_aux"], max_rows, capacity)

    def gather_rows(src, gather, shift, n_rows_dev, max_rows):
        n = int(n_rows_dev[0])
        dst = torch.zeros(max(max_rows, 1), src.shape[1], dtype=src.dtype)
        dst[:n] = src[(gather[:n] >> shift).long()]
        return dst

    def expert_ffn(act, a, gather, shift, w1, b1, w2, b2, group_offsets, max_rows, timer=None):
        M = a.shape[1]
        n_local, _, I = w2.shape
        y = torch.zeros(max(max_rows, 1), M, dtype=torch.bfloat16)
        offs = group_offsets.tolist()
        for g in range(len(offs) - 1):
            lo, hi = offs[g], offs[g + 1]
            if hi == lo:
                continue
            e = g % n_local
            rows = a[(gather[lo:hi] >> shift).long()] if gather is not None else a[lo:hi]
            h = F.linear(rows, w1[e], None if b1 is None else b1[e])
            if act == ops.ACT_SWIGLU:
                h = F.silu(h[:, :I]) * h[:, I:]
            else:
                from oracle.expert_mlps import gelu_new
                h = gelu_new(h)
            y[lo:hi] = F.linear(h, w2[e], None if b2 is None else b2[e])
        return y

    def combine(y, r, S, M):
        out = torch.zeros(S, M, dtype=torch.float32)
        wq = r.weight.to(torch.bfloat16).float()
        for j in range(2):
            on = (r.slot[:, j] >= 0) & (wq[:, j] != 0)
            pos = (r.expert_offsets[r.expert_idx[:, j].long()] + r.slot[:, j]).long()
            out[on] += wq[on, j, None] * y[pos[on]].float()
        return out.to(torch.bfloat16)

    ops.gate_logits, ops.routing_top2, ops.gather_rows = gate_logits, routing_top2, gather_rows
    ops.expert_ffn, ops.combine = expert_ffn, combine


def _build(M, I, E, ep, seed, kind):
    import moe_llava_amd
    from moe_llava_amd.expert_mlps import PhiMLP, StableLMMLP
    g = torch.Generator().manual_seed(seed)
    mk = (lambda: StableLMMLP(M, I)) if kind == "stablelm" else (lambda: PhiMLP(M, I))
    proto = mk().to(torch.bfloat16)
    full = moe_llava_amd.MoE(M, proto, num_experts=E, ep_size=1, k=2, capacity_factor=1.0, eval_capacity_factor=2.0,
                             min_capacity=0, top2_2nd_expert_sampling=False)
    for p in full.deepspeed_moe.experts.parameters():
        p.data = (torch.randn(p.shape, generator=g) * 0.05).to(torch.bfloat16)
    full.deepspeed_moe.gate.wg.weight.data = torch.randn(E, M, generator=g) * 0.1
    part = moe_llava_amd.MoE(M, proto, num_experts=E, ep_size=ep, k=2, capacity_factor=1.0, eval_capacity_factor=2.0,
                             min_capacity=0, top2_2nd_expert_sampling=False)
    return full, part


def run(rank: int, world: int, port: int, mode: str, kind: str, result_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu") if mode == "cpu" else torch.device("cuda", 0)
        if mode == "cpu":
            _install_cpu_stand_ins()
        else:
            torch.cuda.set_device(dev)
        M, I, E, ep = (64, 128, 4, 2) if mode == "cpu" else (128, 256, 4, 2)
        full, part = _build(M, I, E, ep, seed=5, kind=kind)
        el = E // ep
        ep_rank = rank % ep
        # this rank's shard of the experts
        for le, mod in enumerate(part.deepspeed_moe.experts.deepspeed_experts):
            mod.load_state_dict(full.deepspeed_moe.experts.deepspeed_experts[ep_rank * el + le].state_dict())
        part.deepspeed_moe.gate.wg.weight.data = full.deepspeed_moe.gate.wg.weight.data.clone()
        full, part = full.to(dev).train(), part.to(dev).train()
        part.set_deepspeed_parallelism()
        assert part.deepspeed_moe.ep_group is not None and part.expert_group_name == f"ep_size_{ep}"
        # different token counts per rank are NOT allowed upstream (equal-split a2a); ours handles them
        S = 200 + 56 * rank
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(1, S, M, generator=g).to(torch.bfloat16)
        x[:, : S // 2] += 0.7                      # skew so that some choices are dropped
        x = x.to(dev)
        with torch.no_grad():
            want, l_ref, c_ref = full(x)
            got, l_aux, cnt = part(x)
        if mode == "gpu":
            torch.cuda.synchronize()
        assert torch.equal(cnt.cpu(), c_ref.cpu())
        assert float(l_aux) == float(l_ref)
        dropped = int((part.deepspeed_moe.last_routing.slot.cpu() < 0).sum())
        if mode == "gpu":
            assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
        else:
            assert torch.allclose(got.float(), want.float(), rtol=1e-2, atol=1e-3)
        dist.barrier()
        with open(os.path.join(result_dir, f"ok_{rank}"), "w") as f:
            f.write(f"S={S} dropped={dropped}\n")
    finally:
        dist.destroy_process_group()
