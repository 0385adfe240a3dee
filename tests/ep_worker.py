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

    def gate_logits(x, wg, jitter_seed=0):
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

    def chunk_pos(offs, C):
        """chunk-major position of every compact row (host restatement of chunk_rows_kernel's map)."""
        offs = offs.tolist()
        E = len(offs) - 1
        k = [offs[e + 1] - offs[e] for e in range(E)]
        pos = torch.zeros(max(offs[-1], 1), dtype=torch.long)
        nxt = 0
        for c in range(C):
            for e in range(E):
                lo, hi = k[e] * c // C, k[e] * (c + 1) // C
                for i in range(lo, hi):
                    pos[offs[e] + i] = nxt
                    nxt += 1
        return pos[: offs[-1]]

    def ep_chunk_rows(src, gather, shift, expert_offsets, n_chunks, unpack, max_rows, out=None):
        n = int(expert_offsets[-1])
        pos = chunk_pos(expert_offsets, n_chunks)
        if out is None:
            out = torch.zeros(max(max_rows, 1), src.shape[1], dtype=src.dtype)
        if unpack:
            out[:n] = src[pos]
        else:
            out[pos] = src[(gather[:n] >> shift).long()] if gather is not None else src[:n]
        return out

    def expert_ffn(act, a, gather, shift, w1, b1, w2, b2, group_offsets, max_rows, timer=None, n_gated=-1, out=None):
        M = a.shape[1]
        n_local, _, I = w2.shape
        y = out[0] if out is not None else torch.zeros(max(max_rows, 1), M, dtype=torch.bfloat16)
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

    # ---- backward stand-ins (torch autograd per group), so the reverse exchange can be rehearsed on the CPU ----
    def _ffn(act, rows, w1e, b1e, w2e, b2e, I):
        ab = F.linear(rows, w1e, b1e)
        if act == ops.ACT_SWIGLU:
            h = F.silu(ab[:, :I]) * ab[:, I:]
        else:
            from oracle.expert_mlps import gelu_new
            h = gelu_new(ab)
        return ab, h, F.linear(h, w2e, b2e)

    def expert_ffn_train(act, a, gather, shift, w1, b1, w2, b2, group_offsets, max_rows, n_gated=-1, out=None):
        M = a.shape[1]
        n_local, _, I = w2.shape
        rows_n = max(max_rows, 1)
        if out is not None:
            y, ab, h = out
        else:
            y = torch.zeros(rows_n, M, dtype=torch.bfloat16)
            ab = torch.zeros(rows_n, w1.shape[1], dtype=torch.bfloat16)
            h = torch.zeros(rows_n, I, dtype=torch.bfloat16)
        offs = group_offsets.tolist()
        for g in range(len(offs) - 1):
            lo, hi = offs[g], offs[g + 1]
            if hi > lo:
                e = g % n_local
                rows = a[(gather[lo:hi] >> shift).long()] if gather is not None else a[lo:hi]
                ab[lo:hi], h[lo:hi], y[lo:hi] = _ffn(act, rows, w1[e], None if b1 is None else b1[e], w2[e],
                                                     None if b2 is None else b2[e], I)
        return y, ab, h

    def expert_ffn_bwd(act, a, gather, shift, w1, w2, ab, h, dy, group_offsets, max_rows, has_b1, has_b2, n_gated=-1):
        M = a.shape[1]
        n_local, _, I = w2.shape
        da = torch.zeros(max(max_rows, 1), M, dtype=torch.bfloat16)
        dw1, dw2 = torch.zeros(w1.shape), torch.zeros(w2.shape)
        db1 = torch.zeros(n_local, w1.shape[1]) if has_b1 else None
        db2 = torch.zeros(n_local, M) if has_b2 else None
        offs = group_offsets.tolist()
        for g in range(len(offs) - 1):
            lo, hi = offs[g], offs[g + 1]
            if hi == lo:
                continue
            e = g % n_local
            rows = (a[(gather[lo:hi] >> shift).long()] if gather is not None else a[lo:hi]).clone().requires_grad_(True)
            W1, W2 = w1[e].clone().requires_grad_(True), w2[e].clone().requires_grad_(True)
            # biases do not change any gradient except their own (= column sums), the pre-activation is rebuilt from ab
            B1 = torch.zeros(w1.shape[1], dtype=torch.bfloat16, requires_grad=True)
            B2 = torch.zeros(M, dtype=torch.bfloat16, requires_grad=True)
            with torch.enable_grad():
                resid = (ab[lo:hi].float() - F.linear(rows, W1).float()).to(torch.bfloat16).detach()   # = b1 (+rounding)
                pre = F.linear(rows, W1) + resid + B1
                if act == ops.ACT_SWIGLU:
                    hh = F.silu(pre[:, :I]) * pre[:, I:]
                else:
                    from oracle.expert_mlps import gelu_new
                    hh = gelu_new(pre)
                yy = F.linear(hh, W2, B2)
                yy.backward(dy[lo:hi])
            da[lo:hi] = rows.grad
            dw1[e] += W1.grad.float()
            dw2[e] += W2.grad.float()
            if has_b1:
                db1[e] += B1.grad.float()
            if has_b2:
                db2[e] += B2.grad.float()
        return da, dw1.to(torch.bfloat16), db1, dw2.to(torch.bfloat16), db2

    def _pos(r):
        return (r.expert_offsets[r.expert_idx.long()] + r.slot).long()

    def combine_bwd(dout, y, r, S, M):
        dy = torch.zeros_like(y)
        k = r.expert_idx.shape[1]
        dweight = torch.zeros(S, k)
        wq = r.weight.to(torch.bfloat16).float()
        pos = _pos(r)
        for j in range(k):
            on = r.slot[:, j] >= 0
            dy[pos[on, j]] = (wq[on, j, None] * dout[on].float()).to(torch.bfloat16)
            dweight[on, j] = (dout[on].float() * y[pos[on, j]].float()).sum(1)
        return dy, dweight

    def gate_bwd(x, logits, r, dweight, dl_aux, jitter_seed=0):
        S, E = logits.shape
        lg = logits.clone().requires_grad_(True)
        with torch.enable_grad():
            gates = torch.softmax(lg, dim=1)
            g = gates.gather(1, r.expert_idx.long()) * (r.slot >= 0).float()
            den = torch.clamp(g.sum(1, keepdim=True), min=torch.finfo(torch.float32).eps)
            w = g / den
            ce = torch.bincount(r.expert_idx[:, 0].long(), minlength=E).float() / S
            l_aux = (gates.mean(0) * ce).mean() * E * E
            tot = (w * dweight).sum() + (0.0 if dl_aux is None else dl_aux) * l_aux
            tot.backward()
        return lg.grad, lg.grad.t() @ x.float()

    def dispatch_bwd(da_rows, r, dlogits, wg_weight, S, M, jitter_seed=0):
        dx = torch.zeros(S, M)
        pos = _pos(r)
        for j in range(r.expert_idx.shape[1]):
            on = r.slot[:, j] >= 0
            dx[on] += da_rows[pos[on, j]].float()
        if dlogits is not None:
            dx += dlogits @ wg_weight.float()
        return dx.to(torch.bfloat16)

    ops.gate_logits, ops.routing_top2, ops.gather_rows = gate_logits, routing_top2, gather_rows
    ops.ep_chunk_rows = ep_chunk_rows
    ops.expert_ffn, ops.combine = expert_ffn, combine
    def block_top2(x, wg, cap, noise, seed, drop, act, w1, b1, w2, b2, n_gated=-1):
        lg = gate_logits(x, wg)
        r = routing_top2(lg, cap, noise, drop)
        y = expert_ffn(act, x, r.perm, 1, w1, b1, w2, b2, r.expert_offsets, r.max_rows)
        return combine(y, r, x.shape[0], x.shape[1]), lg, r

    ops.block_top2 = block_top2
    ops.expert_ffn_routed = (lambda act, x, r, w1, b1, w2, b2, timer=None, n_gated=-1:
                             expert_ffn(act, x, r.perm, r.top_k - 1, w1, b1, w2, b2, r.expert_offsets, r.max_rows))
    ops.expert_ffn_train, ops.expert_ffn_bwd, ops.combine_bwd = expert_ffn_train, expert_ffn_bwd, combine_bwd
    ops.gate_bwd, ops.dispatch_bwd = gate_bwd, dispatch_bwd


def _build(M, I, E, ep, seed, kind):
    import moe_llava_amd
    from tests.expert_mlps import PhiMLP, StableLMMLP
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


def run_single_rank_rccl(rank: int, port: int, result_dir: str):
    """The exchange code on the real RCCL backend with a one-rank group (RCCL refuses two ranks on one GPU):
    asynchronous all_to_all_single on the communicator's stream, chunked pipeline, results equal the local path."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        full, _ = _build(256, 512, 4, 1, seed=8, kind="stablelm")
        full = full.to(dev).eval()
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 700, 256, generator=g).to(torch.bfloat16).to(dev)
        with torch.no_grad():
            want, l_ref, c_ref = full(x)
            layer = full.deepspeed_moe
            layer.force_exchange = True
            layer._set_ep_group(dist.group.WORLD)
            for chunks in ("1", "4"):
                os.environ["MOE_EP_CHUNKS"] = chunks
                got, l_aux, cnt = full(x)
                torch.cuda.synchronize()
                assert layer.last_ffn_launches == int(chunks)
                assert torch.equal(got, want) and torch.equal(cnt, c_ref) and float(l_aux) == float(l_ref)
        # training step through the exchange path (unchunked) on RCCL
        del os.environ["MOE_EP_CHUNKS"]
        full.train()
        xg = x.clone().requires_grad_(True)
        o, la, _ = full(xg)
        (o.float().sum() + la).backward()
        torch.cuda.synchronize()
        assert torch.isfinite(xg.grad.float()).all()
        with open(os.path.join(result_dir, "ok_0"), "w") as f:
            f.write("rccl single rank ok\n")
    finally:
        dist.destroy_process_group()


def run_chunk_agreement(rank: int, world: int, port: int, result_dir: str):
    """The DEFAULT chunk rule (MOE_EP_CHUNKS unset) with per-rank token counts on either side of its thresholds, and a
    bandwidth override present on rank 0 only: every rank must issue the same number of exchanges (a rank that chose
    another count would hang the group or mis-slice the rows), and the result must not depend on the count."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("MOE_EP_CHUNKS", None)
    os.environ.pop("MOE_EP_BW_GBS", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _install_cpu_stand_ins()
        from moe_llava_amd import sharded_moe
        M, I, E, ep = 64, 64, 4, 2
        full, part = _build(M, I, E, ep, seed=11, kind="stablelm")
        el = E // ep
        for le, mod in enumerate(part.deepspeed_moe.experts.deepspeed_experts):
            mod.load_state_dict(full.deepspeed_moe.experts.deepspeed_experts[(rank % ep) * el + le].state_dict())
        part.deepspeed_moe.gate.wg.weight.data = full.deepspeed_moe.gate.wg.weight.data.clone()
        full, part = full.eval(), part.eval()          # eval capacity 2.0: nothing dropped, rows = 2 S
        part.set_deepspeed_parallelism()
        layer = part.deepspeed_moe
        # rank 0 sends 2 * 9000 = 18000 rows (>= the 16384-row threshold of the no-measurement rule), rank 1 2 * 7000
        S = 9000 if rank == 0 else 7000
        x = torch.randn(1, S, M, generator=torch.Generator().manual_seed(40 + rank)).to(torch.bfloat16)
        seen = []
        for phase in ("no bandwidth known", "override on rank 0 only"):
            if phase.startswith("override"):
                sharded_moe._EP_BW.clear()             # as on a fresh process group: the probe runs again, on every rank
                if rank == 0:
                    os.environ["MOE_EP_BW_GBS"] = "0.5"   # slow links: the rule wants several chunks
            with torch.no_grad():
                want, _, c_ref = full(x)
                got, _, cnt = part(x)
            launches = [None] * world
            dist.all_gather_object(launches, int(layer.last_ffn_launches))
            assert len(set(launches)) == 1, (phase, launches)
            assert torch.equal(cnt, c_ref)
            assert torch.allclose(got.float(), want.float(), rtol=1e-2, atol=1e-3)
            seen.append(launches[0])
        assert seen[0] == 1, seen                      # min over ranks of (send, recv) is below 16384 rows
        assert seen[1] > 1, seen                       # rank 0's override was adopted by rank 1 too
        with open(os.path.join(result_dir, f"ok_{rank}"), "w") as f:
            f.write(f"launches={seen}\n")
    finally:
        os.environ.pop("MOE_EP_BW_GBS", None)
        dist.destroy_process_group()


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
        if mode == "gpu":
            # the EP = 2 output DIRECTLY against the literal dense-mask oracle (all E experts on the CPU, same injected
            # logits): routing is per rank and an expert is the same function wherever it runs, so upstream's EP output
            # for this rank's tokens is the ep_size = 1 oracle's
            from oracle import ds_moe_literal as lit
            from oracle.expert_mlps import PhiMLPRef, StableLMMLPRef
            from tests._parity import assert_out_close
            ref = lit.MoERef(M, (StableLMMLPRef if kind == "stablelm" else PhiMLPRef)(M, I).to(torch.bfloat16),
                             num_experts=E, ep_size=1, k=2, capacity_factor=1.0, eval_capacity_factor=2.0, min_capacity=0,
                             top2_2nd_expert_sampling=False)
            ref.load_state_dict({k: v.cpu() for k, v in full.state_dict().items()}, strict=True)
            ref.train()
            ref.deepspeed_moe.gate.logits_override = part.deepspeed_moe.gate.last_logits.cpu()
            with torch.no_grad():
                r_out, r_laux, r_cnt = ref(x.cpu())
            assert torch.equal(cnt.cpu(), r_cnt) and abs(float(l_aux) - float(r_laux)) <= 1e-5
            assert_out_close(f"EP=2 rank {rank} ({kind}) vs literal oracle", got.cpu(), r_out)
        # ---- pipelined exchange (3 chunks, odd on purpose): same rows, same results ----
        os.environ["MOE_EP_CHUNKS"] = "3"
        try:
            with torch.no_grad():
                got3, l3, c3 = part(x)
        finally:
            del os.environ["MOE_EP_CHUNKS"]
        assert part.deepspeed_moe.last_ffn_launches == 3
        assert torch.equal(c3.cpu(), c_ref.cpu()) and float(l3) == float(l_ref)
        if mode == "gpu":
            torch.cuda.synchronize()
            assert torch.equal(got3, want), float((got3.float() - want.float()).abs().max())
        else:
            assert torch.allclose(got3.float(), want.float(), rtol=1e-2, atol=1e-3)
        # ---- backward through the exchange: dy rows travel the reverse route ----
        g2 = torch.Generator().manual_seed(300 + rank)
        R = torch.randn(1, S, M, generator=g2).to(torch.bfloat16).to(dev)
        grads = []
        for model in (full, part):
            model.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            o, la, _ = model(xg)
            torch.autograd.backward([o, la], [R, torch.tensor(0.25, device=dev)])
            grads.append((xg.grad, model.deepspeed_moe.gate.wg.weight.grad))
        if mode == "gpu":
            torch.cuda.synchronize()
        (dx_f, dwg_f), (dx_p, dwg_p) = grads
        if mode == "gpu":
            assert torch.equal(dx_p, dx_f), float((dx_p.float() - dx_f.float()).abs().max())
        else:
            assert torch.allclose(dx_p.float(), dx_f.float(), rtol=2e-2, atol=2e-3)
        assert torch.allclose(dwg_p.float(), dwg_f.float(), rtol=1e-4, atol=1e-5)
        # ---- the same training step with the exchange pipelined in 2 chunks (forward and both backward exchanges):
        #      rows are the same rows, only their order in the receive buffer changes (chunk, source, expert) ----
        os.environ["MOE_EP_CHUNKS"] = "2"
        try:
            part.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            o2, la2, _ = part(xg)
            assert part.deepspeed_moe.last_ffn_launches == 2
            torch.autograd.backward([o2, la2], [R, torch.tensor(0.25, device=dev)])
        finally:
            del os.environ["MOE_EP_CHUNKS"]
        if mode == "gpu":
            torch.cuda.synchronize()
            assert torch.equal(o2, want) and torch.equal(xg.grad, dx_p)
        else:
            assert torch.allclose(xg.grad.float(), dx_p.float(), rtol=2e-2, atol=2e-3)
        chunked_grads = {n: p.grad.detach().float().cpu().clone() for n, p in part.deepspeed_moe.experts.named_parameters()}
        part.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        o, la, _ = part(xg)
        torch.autograd.backward([o, la], [R, torch.tensor(0.25, device=dev)])
        for n, p in part.deepspeed_moe.experts.named_parameters():
            g1 = p.grad.detach().float().cpu()
            sc = float(g1.abs().max())
            assert float((chunked_grads[n] - g1).abs().max()) <= 1e-2 * sc + 1e-6, n   # another summation order over rows
        # an expert's weight gradient sums over the tokens of BOTH ranks: all-reduce the ep=1 model's gradients
        fe = full.deepspeed_moe.experts.deepspeed_experts
        for le, mod in enumerate(part.deepspeed_moe.experts.deepspeed_experts):
            for q in range(ep):
                ref_mod = fe[q * el + le]
                for (n, p_ref) in ref_mod.named_parameters():
                    tot = p_ref.grad.detach().float().cpu().clone()
                    dist.all_reduce(tot, group=part.deepspeed_moe.ep_group)   # the replicas of the shard inside ONE EP group
                    if q == ep_rank:
                        got = dict(mod.named_parameters())[n].grad.float().cpu()
                        scale = float(tot.abs().max())
                        err = float((got - tot).abs().max())
                        assert err <= 2e-2 * scale + 1e-6, (n, err, scale)
        dist.barrier()
        with open(os.path.join(result_dir, f"ok_{rank}"), "w") as f:
            f.write(f"S={S} dropped={dropped}\n")
    finally:
        dist.destroy_process_group()
