"""GPU tier of SURVEY.md section 8 f2: checkpoints written from a HIP engine after REAL train steps.

* train 2 steps on MI355X -> save (the dictionary of dpc/main.py:166-174) -> the reference's own resume lines restated
  (dpc/main.py:91-98: torch.load -> DataParallel(model).load_state_dict strict -> optimizer.load_state_dict) accept the file
  and hold the engine's weights / Adam moments bit for bit;
* a fresh engine resumed from that file runs the third step BIT-IDENTICAL to an engine that was never interrupted (weights,
  moments, loss / top-k, and the dropout stream, which continues from the saved step);
* ``--pretrain`` of the DPC file into the LC classifier's engine (key intersection, backbone/resnet_2d3d.py:310-333):
  backbone + ConvGRU weights arrive, head and running statistics stay at their initial values, and the classifier steps."""
import os

import pytest
import torch
import torch.nn as nn

from dpc_amd import checkpoint as ckpt
from dpc_amd.engine import DPCEngine
from dpc_amd.lc import LC, LCEngine
from dpc_amd.model import DPC_RNN
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(dtype, seed=233):
    eng = DPCEngine("resnet18", 64, 8, 5, 3, 4, DEV, dtype, seed=seed)
    eng.load_params(O.init_params_reference_style("resnet18", seed=0))
    return eng


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_save_resume_is_bit_identical_to_an_uninterrupted_run(tmp_path, dtype):
    xs = [torch.randn(4, 8, 3, 5, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(10 + i)) for i in range(3)]
    a = _engine(dtype)
    for i in range(2):
        a.train_step(xs[i])
    fn = str(tmp_path / "k400_64_r18_dpc-rnn_bs4_lr0.001_seq8_pred3_len5_ds3" / "epoch1.pth.tar")
    os.makedirs(os.path.dirname(fn))
    ckpt.save_checkpoint(ckpt.build_state(a, 1, "resnet18", 0.125, 2), is_best=False, filename=fn)
    res_a = a.train_step(xs[2]).clone()  # the uninterrupted run goes on
    torch.cuda.synchronize()

    # ---- reference side (dpc/main.py:58-66,80-81,91-98), CPU: the file loads strictly, model AND optimizer
    model = nn.DataParallel(DPC_RNN(64, 8, 5, 3, "resnet18"))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    checkpoint = torch.load(fn, map_location=torch.device("cpu"), weights_only=False)
    assert (checkpoint["epoch"], checkpoint["iteration"], checkpoint["best_acc"], checkpoint["net"]) == (1, 2, 0.125, "resnet18")
    model.load_state_dict(checkpoint["state_dict"])
    optimizer.load_state_dict(checkpoint["optimizer"])
    named = dict(model.module.named_parameters())  # (DataParallel moves a single-device module to cuda:0)
    assert int(optimizer.state[named["backbone.conv1.weight"]]["step"]) == 2

    # ---- resumed engine: third step bit-identical to the uninterrupted one
    b = _engine(dtype)
    b.flat_p.add_(1.0)  # make sure the weights really come from the file
    info = ckpt.resume(b, fn)
    assert info == {"epoch": 1, "iteration": 2, "best_acc": 0.125} and b.step_count == 2
    for k in b.PRM:  # what the reference loader holds == what the resumed engine holds
        assert torch.equal(named[k].detach().cpu(), b.PRM[k].cpu()), k
        o, n = b.offsets[k]
        assert torch.equal(optimizer.state[named[k]]["exp_avg"].flatten().cpu(), b.flat_m[o:o + n].cpu()), k
    res_b = b.train_step(xs[2]).clone()
    torch.cuda.synchronize()
    assert torch.equal(res_a, res_b)
    assert torch.equal(a.flat_p, b.flat_p) and torch.equal(a.flat_m, b.flat_m) and torch.equal(a.flat_v, b.flat_v)
    assert torch.equal(a.dropout_masks_of_step(), b.dropout_masks_of_step())
    # --reset_lr: weights only, Adam starts over -> a different third step
    c = _engine(dtype)
    ckpt.resume(c, fn, reset_lr=True)
    assert c.step_count == 0 and c.flat_m.abs().max().item() == 0
    c.train_step(xs[2])
    assert not torch.equal(c.flat_p, a.flat_p)


def test_pretrain_dpc_file_into_the_lc_engine(tmp_path):
    a = _engine(torch.bfloat16)
    x = torch.randn(4, 8, 3, 5, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    a.train_step(x)
    fn = str(tmp_path / "epoch1.pth.tar")
    ckpt.save_checkpoint(ckpt.build_state(a, 1, "resnet18", 0.0, 1), filename=fn)
    lc = LCEngine("resnet18", 64, 8, 5, 4, DEV, torch.bfloat16, num_class=11)
    init = LC(64, 8, 5, "resnet18", 0.5, 11, seed=5)
    lc.load_params({k: v.detach() for k, v in init.state_dict().items()})
    head0 = lc.PRM["final_fc.1.weight"].clone()
    lines = []
    info = ckpt.pretrain(lc, fn, log=lines.append)
    assert info["epoch"] == 1
    # unused from the file: the predictor; not in the file: the classifier head and every running buffer
    assert "network_pred.0.weight" in lines and "final_fc.1.weight" in lines and "backbone.bn1.running_mean" in lines
    for k in ("backbone.conv1.weight", "backbone.layer3.1.conv2.weight", "agg.ConvGRUCell_00.out_gate.weight"):
        assert torch.equal(lc.PRM[k], a.PRM[k]), k
    assert torch.equal(lc.PRM["final_fc.1.weight"], head0)
    assert lc.BUF["backbone.bn1.running_var"].eq(1).all() and int(lc.BUF["backbone.bn1.num_batches_tracked"]) == 0
    y = torch.tensor([1, 5, 0, 10], device=DEV)
    r0 = lc.train_step(x, y).clone()
    for _ in range(5):
        r = lc.train_step(x, y).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(r).all() and r[0].item() < r0[0].item() and int(lc.BUF["backbone.bn1.num_batches_tracked"]) == 6
