"""The command-line entries end to end: `python -m dpc_amd.main` (dpc/main.py:50-174: epochs of train() + validate(), checkpoint
rotation, --resume with the `_lr(.+?)_` regex of :93, --reset_lr) and `python -m dpc_amd.lc_main` (eval/test.py:50-216: --pretrain of
a DPC-RNN checkpoint by key intersection, train / validate, --test).

CPU tier: the same `main(argv)` functions on the host-side SIMT simulator at narrow widths (the block classes are width-parametric,
backbone/resnet_2d3d.py:50,86) -- every line of the entries except the device selection runs.  GPU tier: the real command lines as
child processes on the MI355X (tests/test_entries.py::test_*_gpu)."""
import os
import re
import subprocess
import sys

import pytest
import torch

from dpc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDTHS = (8, 16, 32, 32)


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    return L.load_emulator()


def _check_run1(d, out, nb=2):
    """what one epoch of `main` over nb batches leaves behind (dpc/main.py:150-174, utils/utils.py:14-26)"""
    assert re.search(r"Epoch: \[0\]\[0/%d\]\s+Loss [0-9.]+ \([0-9.]+\)\s+Acc: top1 [0-9.]+; top3 [0-9.]+; top5 [0-9.]+ T:" % nb, out), out
    assert re.search(r"\[0/1\] Loss [0-9.]+\s+Acc: top1", out), out          # validate()'s line, dpc/main.py:280
    assert "Training from ep 0 to ep 1 finished" in out
    f1 = os.path.join(d, "epoch1.pth.tar")
    ck = torch.load(f1, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "net", "state_dict", "best_acc", "optimizer", "iteration"}   # dpc/main.py:167-172
    assert ck["epoch"] == 1 and ck["net"] == "resnet18" and ck["iteration"] == nb           # every train step is logged (print_freq 1)
    assert all(k.startswith("module.") for k in ck["state_dict"]) and "module.agg.cell_list.0.out_gate.weight" in ck["state_dict"]
    assert float(ck["optimizer"]["state"][0]["step"]) == float(nb)
    return ck


def _check_resume(d, d2, out, ck1, nb=2):
    assert f"=> loaded resumed checkpoint '{os.path.join(d2, 'epoch1.pth.tar')}' (epoch 1)" in out
    assert "==== Change lr from 0.001000 to 0.000500 ====" in out                        # the `_lr(.+?)_` regex, dpc/main.py:93,99
    assert "Epoch: [1][0/%d]" % nb in out and "Epoch: [0]" not in out                          # start_epoch restored
    assert "Training from ep 1 to ep 2 finished" in out
    assert not os.path.exists(os.path.join(d2, "epoch1.pth.tar"))                        # utils/utils.py:18-21: the previous epoch's file goes
    ck2 = torch.load(os.path.join(d2, "epoch2.pth.tar"), map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 2 and ck2["iteration"] == ck1["iteration"] + nb and ck2["best_acc"] >= ck1["best_acc"]
    # --reset_lr: the optimizer state of the file is NOT loaded (dpc/main.py:97-100): moments restart, the step counter too
    assert float(ck2["optimizer"]["state"][0]["step"]) == float(nb) and ck2["optimizer"]["param_groups"][0]["lr"] == 5e-4
    moved = [k for k in ck1["state_dict"] if not torch.equal(ck1["state_dict"][k], ck2["state_dict"][k])]
    assert len(moved) >= 0.9 * len(ck1["state_dict"])                                      # the run trained on from the loaded values


def _check_lc(out, path, nb=2):
    assert f"=> loaded pretrain checkpoint '{path}' (epoch 2;" in out
    m = re.search(r"\(epoch 2; (\d+) keys not in the file, (\d+) keys of the file unused\)", out)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0     # LC's head / running buffers are new; DPC's predictor is not used
    assert re.search(r"Epoch: \[0\]\[0/%d\]\s+Loss [0-9.]+\s+Acc: [0-9.]+\s+lr 0.001" % nb, out), out
    assert "Training from ep 0 to ep 1 finished" in out


def test_main_and_lc_main_on_the_simulator(emu, tmp_path, capsys):
    from dpc_amd import lc_main, main as dpc_main
    d = str(tmp_path / "run")
    common = ["--net", "resnet18", "--img_dim", "64", "--batch_size", "1", "--gpu", "0", "--synthetic", "1", "--print_freq", "1",
              "--dtype", "f32", "--num_seq", "4", "--pred_step", "1"]   # half the sequence, one batch per epoch: this tier is about the host side
    dpc_main.main(common + ["--epochs", "1", "--save_dir", d], _simulator=emu, _widths=WIDTHS)
    ck1 = _check_run1(d, capsys.readouterr().out, nb=1)
    # resume from a path that spells the old learning rate the way the reference's experiment folders do (dpc/main.py:52-56,93)
    d2 = str(tmp_path / "dpc-rnn_lr0.001_wd1e-05" / "model")
    os.makedirs(d2)
    os.rename(os.path.join(d, "epoch1.pth.tar"), os.path.join(d2, "epoch1.pth.tar"))
    dpc_main.main(common + ["--epochs", "2", "--save_dir", d2, "--resume", os.path.join(d2, "epoch1.pth.tar"), "--reset_lr", "--lr", "5e-4"],
                  _simulator=emu, _widths=WIDTHS)
    _check_resume(d, d2, capsys.readouterr().out, ck1, nb=1)
    # the downstream classifier starts from that DPC-RNN checkpoint (eval/test.py:150-166)
    f2 = os.path.join(d2, "epoch2.pth.tar")
    lc_main.main(["--net", "resnet18", "--img_dim", "64", "--batch_size", "1", "--gpu", "0", "--synthetic", "1", "--print_freq", "1",
                  "--dtype", "f32", "--num_seq", "4", "--epochs", "1", "--pretrain", f2],
                 _simulator=emu, _widths=WIDTHS)
    _check_lc(capsys.readouterr().out, f2, nb=1)


def test_main_two_ranks(emu, tmp_path, capfd):
    """`--gpu 0,1`: the rank processes of dpc_amd.main (mp.spawn, init_process_group, per-rank dropout seeds, the two-bucket gradient
    exchange inside train_step, the averaged metrics, rank-0-only checkpointing) -- what replaces nn.DataParallel (dpc/main.py:65,
    torch data_parallel.py:173-198).  On this tier the ranks run the simulator and exchange over gloo; on an 8-GPU node the same
    lines run over RCCL."""
    from dpc_amd import main as dpc_main
    d, pr = str(tmp_path / "run"), str(tmp_path / "probe")
    os.makedirs(pr)
    common = ["--net", "resnet18", "--img_dim", "64", "--batch_size", "2", "--gpu", "0,1", "--synthetic", "2", "--print_freq", "1",
              "--dtype", "f32", "--num_seq", "4", "--pred_step", "1"]
    dpc_main.main(common + ["--epochs", "1", "--save_dir", d], _simulator=emu, _widths=WIDTHS, _probe=pr)
    out = capfd.readouterr().out
    ck1 = _check_run1(d, out, nb=2)
    assert out.count("Training from ep 0 to ep 1 finished") == 1 and out.count("Epoch: [0][0/2]") == 1   # rank 0 alone reports
    assert [f for f in sorted(os.listdir(d)) if f.startswith("epoch")] == ["epoch1.pth.tar"]                # ... and alone writes the file
    r = [torch.load(os.path.join(pr, f"rank{i}.pt")) for i in range(2)]
    assert [x["rank"] for x in r] == [0, 1] and all(x["world"] == 2 and x["per_gpu"] == 1 and x["step"] == 2 for x in r)
    assert r[0]["seed"] == 233 and r[1]["seed"] == 234                      # independent dropout streams per replica (dpc/model_3d.py:18)
    assert torch.equal(r[0]["flat_p"], r[1]["flat_p"]) and torch.equal(r[0]["flat_m"], r[1]["flat_m"])   # averaged gradients, same update
    assert torch.isfinite(r[0]["flat_p"]).all() and r[0]["flat_m"].abs().sum() > 0
    # the saved state_dict is rank 0's arena
    sd = ck1["state_dict"]
    assert torch.equal(sd["module.backbone.conv1.weight"].flatten(), r[0]["flat_p"][:sd["module.backbone.conv1.weight"].numel()])
    # ... and it resumes on two ranks: strict load + optimizer state, one more epoch, both ranks still identical
    dpc_main.main(common + ["--epochs", "2", "--save_dir", d, "--resume", os.path.join(d, "epoch1.pth.tar")],
                  _simulator=emu, _widths=WIDTHS, _probe=pr)
    out = capfd.readouterr().out
    assert out.count(f"=> loaded resumed checkpoint '{os.path.join(d, 'epoch1.pth.tar')}' (epoch 1)") == 1
    assert "Epoch: [1][0/2]" in out and "Training from ep 1 to ep 2 finished" in out
    r2 = [torch.load(os.path.join(pr, f"rank{i}.pt")) for i in range(2)]
    assert all(x["step"] == 4 for x in r2) and torch.equal(r2[0]["flat_p"], r2[1]["flat_p"]) and not torch.equal(r2[0]["flat_p"], r[0]["flat_p"])
    ck2 = torch.load(os.path.join(d, "epoch2.pth.tar"), map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 2 and float(ck2["optimizer"]["state"][0]["step"]) == 4.0 and [f for f in sorted(os.listdir(d)) if f.startswith("epoch")] == ["epoch2.pth.tar"]


def test_entry_errors_follow_the_reference():
    from dpc_amd import main as dpc_main
    a = dpc_main.build_parser().parse_args(["--model", "dpc", "--gpu", "0"])
    a._simulator, a._widths = object(), WIDTHS   # never reached: the model check comes first
    with pytest.raises(ValueError, match="wrong model!"):       # dpc/main.py:63
        dpc_main._worker(0, 1, a, 0)
    a = dpc_main.build_parser().parse_args(["--train_what", "last", "--gpu", "0"])
    a._simulator, a._widths = object(), WIDTHS
    with pytest.raises(AttributeError, match="resnet"):         # dpc/main.py:71
        dpc_main._worker(0, 1, a, 0)


def _run(launcher, argv, timeout=600):
    """`python -m <argv>` from the repository root.  In the GPU tier the child is forked by the pre-GPU launcher process
    (tests/conftest.py: forking from a process whose HIP runtime is up segfaulted once in five runs)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-c", "import os, runpy, sys; os.chdir(sys.argv[1]); sys.argv = sys.argv[2:]; runpy.run_module(sys.argv[0], run_name='__main__')",
           ROOT] + argv
    if launcher is not None:
        rc, out, err = launcher.run(cmd, env, timeout)
    else:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        rc, out, err = r.returncode, r.stdout, r.stderr
    assert rc == 0, out[-3000:] + "\n" + err[-3000:]
    return out


@pytest.mark.gpu
def test_main_and_lc_main_gpu(tmp_path, clean_launcher):
    """the real command lines on the MI355X (bf16, full widths): train + validate + save, --resume --reset_lr, then the classifier
    --pretrain'ed from the file, then its --test of the file it wrote"""
    d = str(tmp_path / "run")
    common = ["--net", "resnet18", "--img_dim", "64", "--batch_size", "4", "--gpu", "0", "--synthetic", "2", "--print_freq", "1"]
    out = _run(clean_launcher, ["dpc_amd.main"] + common + ["--epochs", "1", "--save_dir", d])
    ck1 = _check_run1(d, out)
    d2 = str(tmp_path / "dpc-rnn_lr0.001_wd1e-05" / "model")
    os.makedirs(d2)
    os.rename(os.path.join(d, "epoch1.pth.tar"), os.path.join(d2, "epoch1.pth.tar"))
    out = _run(clean_launcher, ["dpc_amd.main"] + common + ["--epochs", "2", "--save_dir", d2, "--resume", os.path.join(d2, "epoch1.pth.tar"), "--reset_lr",
                                           "--lr", "5e-4"])
    _check_resume(d, d2, out, ck1)
    f2 = os.path.join(d2, "epoch2.pth.tar")
    d3 = str(tmp_path / "lc")
    out = _run(clean_launcher, ["dpc_amd.lc_main"] + common + ["--epochs", "1", "--pretrain", f2, "--save_dir", d3])
    _check_lc(out, f2)
    f3 = os.path.join(d3, "epoch1.pth.tar")
    ck3 = torch.load(f3, map_location="cpu", weights_only=False)
    assert ck3["epoch"] == 1 and any("final_fc" in k for k in ck3["state_dict"])
    out = _run(clean_launcher, ["dpc_amd.lc_main"] + common + ["--test", f3])
    assert f"=> loaded test checkpoint '{f3}' (epoch 1; 0 keys not in the file, 0 keys of the file unused)" in out
    assert re.search(r"Loss [0-9.]+\s+Acc top1: [0-9.]+ Acc top5: [0-9.]+", out) and "(test checkpoint epoch 1)" in out
