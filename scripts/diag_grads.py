import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O
B, size, net = 2, 64, "resnet18"
eng = DPCEngine(net, size, 8, 5, 3, B, "cuda:0", torch.float32)
p = O.make_params_pcg(net)
eng.load_params(p)
x = O.make_input_pcg(B, 8, 5, size)
score = eng.forward(x.cuda(), train=False).cpu()
res = eng.loss_topk(True).cpu()
eng.backward()
torch.cuda.synchronize()
loss, accs, grads, ref = O.train_step_reference(p, x, net, 3, None)
print("score err", (score-ref).abs().max().item())
p64 = {k: v.double() for k,v in p.items()}
l64, _, g64, _ = O.train_step_reference(p64, x.double(), net, 3, None)
print("rel-to-maxabs error vs fp64 oracle:   mine | fp32-oracle | maxabs | name")
for k, g in g64.items():
    mine = eng.G[k].cpu().double(); o32 = grads[k].double()
    s = max(g.abs().max().item(),1e-12)
    print(f"{(mine-g).abs().max().item()/s:10.3e} {(o32-g).abs().max().item()/s:10.3e} {s:10.3e} {k}")
# ---- ReLU-mask flips: recompute every block's bn1 pre-activation in fp64 from the engine's own block input
import torch.nn.functional as F
names = [f"backbone.layer{li+1}.{bi}." for li in range(4) for bi in range(2)]
for blk, pre in zip(eng.blocks, names):
    xin = blk.x_in.cpu().double().permute(0,4,1,2,3)
    is3d = pre.startswith("backbone.layer3") or pre.startswith("backbone.layer4")
    s = blk.c1.s; pd = blk.c1.p
    z = O.bn_batch(F.conv3d(xin, p64[pre+"conv1.weight"], None, s, pd), p64[pre+"bn1.weight"], p64[pre+"bn1.bias"])
    mine = blk.act1.cpu().double().permute(0,4,1,2,3)
    flips = ((z > 0) != (mine > 0))
    print(pre, "act1 numel", z.numel(), "mask flips", int(flips.sum()), "|z| at flips", z[flips].abs().tolist()[:5],
          "max |act1-relu(z)|", (mine - F.relu(z)).abs().max().item())
# ---- mask flips against the FULL-chain fp64 / fp32 oracle (their own activations)
def chain(pp, xx):
    acts = []
    h = F.conv3d(xx.reshape(-1, 3, 5, size, size), pp["backbone.conv1.weight"], None, (1,2,2), (0,3,3))
    h = F.relu(O.bn_batch(h, pp["backbone.bn1.weight"], pp["backbone.bn1.bias"]))
    h = F.max_pool3d(h, (1,3,3), (1,2,2), (0,1,1))
    for li in range(4):
        for bi in range(2):
            pre = f"backbone.layer{li+1}.{bi}."
            is3d = li >= 2; stride = 2 if (li > 0 and bi == 0) else 1
            s1 = (stride,)*3 if is3d else (1,stride,stride); pad = (1,1,1) if is3d else (0,1,1)
            a1 = F.relu(O.bn_batch(F.conv3d(h, pp[pre+"conv1.weight"], None, s1, pad), pp[pre+"bn1.weight"], pp[pre+"bn1.bias"]))
            h = O.basic_block(h, pp, pre, is3d, stride, final_relu=not (li == 3 and bi == 1))
            acts.append((pre, a1, h))
    return acts
with torch.no_grad():
    a64 = chain(p64, x.double()); a32 = chain(p, x)
for blk, (pre, a1, out), (_, b1, bout) in zip(eng.blocks, a64, a32):
    m1 = blk.act1.cpu().permute(0,4,1,2,3); mo = blk.out.cpu().permute(0,4,1,2,3)
    print(pre, "flips vs fp64 chain: act1", int(((m1>0)!=(a1>0)).sum()), "out", int(((mo>0)!=(out>0)).sum()),
          "| fp32-oracle vs fp64 chain: act1", int(((b1>0)!=(a1>0)).sum()), "out", int(((bout>0)!=(out>0)).sum()))
