import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moe_llava_amd
from tests.expert_mlps import StableLMMLP
dev = torch.device("cuda", 0)
moe = moe_llava_amd.MoE(128, StableLMMLP(128, 256), num_experts=4, k=2, capacity_factor=1.5, min_capacity=0).to(torch.bfloat16).to(dev)
for shape in ((0, 7, 128), (2, 0, 128)):
    for train in (False, True):
        moe.train(train)
        try:
            x = torch.zeros(shape, dtype=torch.bfloat16, device=dev, requires_grad=train)
            out, l_aux, cnt = moe(x)
            torch.cuda.synchronize()
            print(shape, train, tuple(out.shape), float(l_aux), cnt.tolist())
            if train:
                (out.float().sum() + l_aux).backward()
                torch.cuda.synchronize()
                print("  grad", tuple(x.grad.shape), [None if p.grad is None else float(p.grad.abs().sum()) for p in list(moe.parameters())[:2]])
        except Exception as e:
            print(shape, train, "raised", type(e).__name__, str(e)[:200])
