"""Training entry with the reference's command line (dpc/main.py:27-47), MI355X-native.

    python -m dpc_amd.main --net resnet18 --img_dim 128 --batch_size 128 --gpu 0,1,2,3 --synthetic 50

Same flags and defaults as the reference's ``main.py``.  What differs is the execution model:
``--gpu 0,1,..`` starts ONE PROCESS PER GPU (the reference wraps the model in nn.DataParallel,
dpc/main.py:65); each process owns a DPCEngine; ``--batch_size`` is the GLOBAL batch, as in the
reference (DataParallel scatters it along dim 0), so each GPU sees batch_size / n_gpu clips; gradients
are averaged with one RCCL all-reduce (dpc_amd/parallel.py).  Datasets / augmentation / tensorboard are outside this build's scope
(SURVEY.md §2 rows 8,9,11): the input is the synthetic N(0,1) video of ``--synthetic`` batches per
epoch with the dataset's tensor layout [B, num_seq, 3, seq_len, H, W] (dpc/dataset_3d.py:109-111), or -- ``--frames clips.npy`` --
decoded uint8 frames [clips, F, H0, W0, 3] on which the training transform of ``--dataset`` (dpc/main.py:114-132) runs on the GPU
(dpc_amd/data.py FrameSource -> engine.load_recipe -> train_step(None): SURVEY.md §8 f4).
Checkpoints are the reference's dictionary (``module.``-prefixed state_dict incl. alias keys, torch-Adam-layout
``optimizer``; dpc/main.py:166-174, utils/utils.py:14-26) written and read by ``dpc_amd/checkpoint.py``: a file
written here resumes in the reference and vice versa (tests/test_checkpoint.py).
"""
from __future__ import annotations

import argparse
import os
import re
import time

import torch


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument('--net', default='resnet18', type=str)
    parser.add_argument('--model', default='dpc-rnn', type=str)
    parser.add_argument('--dataset', default='ucf101', type=str)
    parser.add_argument('--seq_len', default=5, type=int, help='number of frames in each video block')
    parser.add_argument('--num_seq', default=8, type=int, help='number of video blocks')
    parser.add_argument('--pred_step', default=3, type=int)
    parser.add_argument('--ds', default=3, type=int, help='frame downsampling rate')
    parser.add_argument('--batch_size', default=4, type=int)
    parser.add_argument('--lr', default=1e-3, type=float, help='learning rate')
    parser.add_argument('--wd', default=1e-5, type=float, help='weight decay')
    parser.add_argument('--resume', default='', type=str, help='path of model to resume')
    parser.add_argument('--pretrain', default='', type=str, help='path of pretrained model')
    parser.add_argument('--epochs', default=10, type=int, help='number of total epochs to run')
    parser.add_argument('--start-epoch', default=0, type=int, help='manual epoch number (useful on restarts)')
    parser.add_argument('--gpu', default='0,1', type=str)
    parser.add_argument('--print_freq', default=5, type=int, help='frequency of printing output during training')
    parser.add_argument('--reset_lr', action='store_true', help='Reset learning rate when resume training?')
    parser.add_argument('--prefix', default='tmp', type=str, help='prefix of checkpoint filename')
    parser.add_argument('--train_what', default='all', type=str)
    parser.add_argument('--img_dim', default=128, type=int)
    # additions of this build
    parser.add_argument('--synthetic', default=20, type=int, help='synthetic batches per epoch (the data source unless --frames is given)')
    parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'], help='compute dtype (f32 = parity mode)')
    parser.add_argument('--save_dir', default='', type=str, help='where to write checkpoints (default: none)')
    parser.add_argument('--frames', default='', type=str, help='uint8 .npy [clips, F, H0, W0, 3] of decoded frames: the training transform of '
                        '--dataset runs on the GPU (dpc_amd/data.py); replaces --synthetic')
    parser.add_argument('--val_frames', default='', type=str, help='frames for validate() (default: the --frames array; the reference validates '
                        'with the training transform too, dpc/main.py:135-136)')
    parser.add_argument('--crop', default=224, type=int, help='RandomCrop size of the ucf101 recipe (dpc/main.py:117)')
    return parser


class AverageMeter:
    """utils/utils.py:77-113 (value, running average, 5-step local average)"""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0.0
        self.local = []

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.local = (self.local + [val])[-5:]

    @property
    def local_avg(self):
        return sum(self.local) / max(len(self.local), 1)


def average_over_ranks(dist, vals: torch.Tensor, world: int) -> torch.Tensor:
    """mean over ranks of a small device tensor, in place (the logged loss / top-k: equal shards, so the mean of per-rank means is the
    reference's mean over the gathered rows, dpc/main.py:211-218).  RCCL averages in the collective; gloo (CPU tier) has no AVG."""
    if dist is None or world <= 1:
        return vals
    if dist.get_backend() == 'nccl':
        dist.all_reduce(vals, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        vals.div_(world)
    return vals


def _worker(rank: int, world: int, args, port: int):
    gpus = [int(g) for g in str(args.gpu).split(',') if g != '']
    torch.manual_seed(0)  # dpc/main.py:50
    sim = getattr(args, '_simulator', None)   # tests only (CPU tier): the host-side SIMT simulator handle; the command line cannot set it
    if sim == 'emu':   # a spawned rank of the CPU tier loads its own handle (a ctypes library does not pickle)
        from . import _lib
        sim = _lib.load_emulator()
    dev = torch.device('cpu') if sim is not None else torch.device('cuda', gpus[rank] if world > 1 else gpus[0])
    from .parallel import configure_rccl, default_reserve_cus
    if world > 1:
        configure_rccl(world)   # before this process's first HIP call: few long-lived RCCL channels beside the backward pass (parallel.py)
    if sim is None:
        torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if sim is None:
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world, device_id=dev)
        else:   # CPU tier: the same rank processes over gloo
            torch.set_num_threads(max(1, (os.cpu_count() or 2) // (2 * world)))
            dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    from .engine import DPCEngine
    from .model import DPC_RNN
    from .parallel import make_allreduce
    from . import checkpoint as ckpt

    if args.model != 'dpc-rnn':
        raise ValueError('wrong model!')  # dpc/main.py:63
    if args.train_what == 'last':
        # dpc/main.py:69-72 freezes `model.module.resnet`, an attribute DPC_RNN never had (the backbone is `.backbone`,
        # dpc/model_3d.py:29): the reference dies right there with this error, and so does the drop-in -- loudly, not by
        # silently training everything
        raise AttributeError("'DPC_RNN' object has no attribute 'resnet'")
    if args.batch_size % world:
        raise ValueError('batch_size must be divisible by the number of GPUs (drop_last semantics, dpc/main.py:313)')
    per_gpu = args.batch_size // world
    cdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    from .plan import LAYER_WIDTH
    widths = getattr(args, '_widths', None) or LAYER_WIDTH
    eng = DPCEngine(args.net, args.img_dim, args.num_seq, args.seq_len, args.pred_step, per_gpu, dev, cdt, widths, lib=sim, lr=args.lr, wd=args.wd,
                    seed=233 + rank, reserve_cus=default_reserve_cus(world))  # dropout stream: the reference seeds 233 (dpc/model_3d.py:18); independent per replica
    init = DPC_RNN(args.img_dim, args.num_seq, args.seq_len, args.pred_step, args.net, widths=widths, seed=0)  # same on every rank
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    best_acc, iteration = 0.0, 0
    if args.resume:  # dpc/main.py:88-102: strict model load + optimizer state (unless --reset_lr)
        if os.path.isfile(args.resume):
            m = re.search('_lr(.+?)_', args.resume)
            info = ckpt.resume(eng, args.resume, reset_lr=args.reset_lr)
            args.start_epoch, iteration, best_acc = info['epoch'], info['iteration'], info['best_acc']
            if args.reset_lr:
                eng.lr, eng.wd = args.lr, args.wd
                if rank == 0 and m:
                    print('==== Change lr from %f to %f ====' % (float(m.group(1)), args.lr))
            if rank == 0:
                print("=> loaded resumed checkpoint '{}' (epoch {})".format(args.resume, info['epoch']))
        elif rank == 0:
            print("[Warning] no checkpoint found at '{}'".format(args.resume))
    if args.pretrain:  # dpc/main.py:104-112: key intersection (neq_load_customized)
        if os.path.isfile(args.pretrain):
            info = ckpt.pretrain(eng, args.pretrain, log=print if rank == 0 else (lambda *a: None))
            if rank == 0:
                print("=> loaded pretrained checkpoint '{}' (epoch {})".format(args.pretrain, info['epoch']))
        elif rank == 0:
            print("=> no checkpoint found at '{}'".format(args.pretrain))
    allreduce = make_allreduce(dist, world)
    gen = torch.Generator(dev).manual_seed(1000 + rank)
    shape = (per_gpu, args.num_seq, 3, args.seq_len, args.img_dim, args.img_dim)
    src_train = src_val = None
    if args.frames:   # decoded uint8 frames in, the reference's training transform on the GPU (dpc/dataset_3d.py:97-111, dpc/main.py:114-136)
        import random
        import numpy as np
        from .data import FrameSource
        random.seed(rank)
        np.random.seed(rank)      # dpc/main.py:51 seeds 0; one stream per rank like one per loader worker
        mk = lambda path: FrameSource(path, args.dataset, args.num_seq, args.seq_len, args.ds, args.img_dim, per_gpu, rank, world, args.crop)  # noqa: E731
        src_train = mk(args.frames)
        src_val = mk(args.val_frames) if args.val_frames else src_train

    def run_epoch(train: bool, epoch: int):
        losses, accs = AverageMeter(), [AverageMeter() for _ in range(3)]
        nonlocal iteration
        src = src_train if train else src_val
        n_batches = len(src) if src is not None else args.synthetic
        feed = src.epoch(dev) if src is not None else None
        for idx in range(n_batches):
            tic = time.time()
            if feed is not None:
                frames, starts, clips = next(feed)
                eng.load_recipe(frames, starts, clips, ds=src.ds)   # fills the stem's operand; no f32 video in between
                block = None
            else:
                block = torch.randn(shape, device=dev, generator=gen)
            if train:
                res = eng.train_step(block, allreduce=allreduce)
            else:  # validate(): dropout off, BN still batch statistics (dpc/main.py:249-282, model_3d.py:28)
                eng.forward(block, train=False, materialise=False)  # loss / top-k only: the score is never written (bf16)
                res = eng.loss_topk(with_grad=False)
            if idx % args.print_freq == 0 or not train:
                vals = average_over_ranks(dist, res.clone(), world)
                loss, t1, t3, t5 = vals.cpu().tolist()  # ONE packed D2H (the reference does five .item() syncs per step)
                losses.update(loss, per_gpu)
                for a, v in zip(accs, (t1, t3, t5)):
                    a.update(v, per_gpu)
                if train and rank == 0:
                    print('Epoch: [{0}][{1}/{2}]\t Loss {3:.6f} ({4:.4f})\t Acc: top1 {5:.4f}; top3 {6:.4f}; top5 {7:.4f} T:{8:.2f}\t'.format(
                        epoch, idx, n_batches, loss, losses.local_avg, t1, t3, t5, time.time() - tic), flush=True)
                if train:
                    iteration += 1
        return losses.local_avg, accs[0].local_avg, [a.local_avg for a in accs]

    for epoch in range(args.start_epoch, args.epochs):
        run_epoch(True, epoch)
        val_loss, val_acc, val_list = run_epoch(False, epoch)
        if rank == 0:
            print('[{0}/{1}] Loss {2:.4f}\t Acc: top1 {3:.4f}; top3 {4:.4f}; top5 {5:.4f} \t'.format(epoch, args.epochs, val_loss, *val_list), flush=True)
            if args.save_dir:
                os.makedirs(args.save_dir, exist_ok=True)
                is_best = val_acc > best_acc
                best_acc = max(val_acc, best_acc)
                fn = os.path.join(args.save_dir, 'epoch%s.pth.tar' % str(epoch + 1))
                # dpc/main.py:166-174 + utils/utils.py:14-26: same dictionary, same file rotation
                ckpt.save_checkpoint(ckpt.build_state(eng, epoch + 1, args.net, best_acc, iteration), is_best, filename=fn,
                                     keep_all=False)
    if rank == 0:
        print('Training from ep %d to ep %d finished' % (args.start_epoch, args.epochs))
    probe = getattr(args, '_probe', None)   # tests only: every rank leaves its parameter arena behind (identical on all ranks by construction)
    if probe:
        torch.save({'flat_p': eng.flat_p.detach().cpu(), 'flat_m': eng.flat_m.detach().cpu(), 'step': eng.step_count,
                    'rank': rank, 'world': world, 'per_gpu': per_gpu, 'seed': eng.seed}, os.path.join(probe, f'rank{rank}.pt'))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None, _simulator=None, _widths=None, _probe=None):
    args = build_parser().parse_args(argv)
    args._simulator, args._widths, args._probe = _simulator, _widths, _probe   # tests/test_entries.py: the CPU tier runs the entry on the simulator
    gpus = [g for g in str(args.gpu).split(',') if g != '']
    world = max(len(gpus), 1)
    if _simulator is not None and world != 1:
        args._simulator = 'emu'   # the rank processes (gloo instead of RCCL) load the simulator themselves
    if world == 1:
        _worker(0, 1, args, 0)
    else:
        import torch.multiprocessing as mp
        port = 29500 + (os.getpid() % 2000)
        mp.spawn(_worker, args=(world, args, port), nprocs=world, join=True)


if __name__ == '__main__':
    main()
