"""Entry of the downstream classifier with the reference's command line (eval/test.py:26-48), MI355X-native.

    python -m dpc_amd.lc_main --net resnet18 --img_dim 128 --batch_size 128 --gpu 0 --pretrain <dpc checkpoint> --synthetic 20

train() / validate() / test() follow eval/test.py:218-343 on the LCEngine (one process per GPU, RCCL gradient
all-reduce as in dpc_amd.main); datasets, augmentation and tensorboard are outside this build's scope, the input is
synthetic N(0,1) video with random labels in the dataset's tensor layout.  ``--pretrain`` loads a DPC-RNN checkpoint
by key intersection (neq_load_customized, backbone/resnet_2d3d.py:310-333): backbone + ConvGRU weights are taken, the
running buffers and the head stay at their initial values -- exactly what the reference does with its own checkpoints.

Learning rate: the reference wraps Adam in ``LambdaLR(MultiStepLR_Restart_Multiplier)`` and calls ``scheduler.step(epoch)`` after
every epoch (eval/test.py:93-100,197,408-423): epoch e > start runs at ``lr * multiplier(e - 1)``, the first epoch of a run at
the constructor's ``lr * multiplier(0)`` (or the checkpoint's saved lr on ``--resume`` without ``--reset_lr``).  ``lr_multiplier``
below restates the multiplier; the fused Adam takes ``eng.lr`` per step.
``--train_what ft``: the reference gives parameters whose NAME contains 'resnet' or 'rnn' a 10x smaller lr
(eval/test.py:76-84) -- but LC's parameters are called ``backbone.*`` / ``agg.*`` / ``final_*`` (model_3d_lc.py:29-45), so the
filter never matches and every parameter trains at ``--lr``.  This entry does what the reference effectively does: one lr.
"""
from __future__ import annotations

import argparse
import os

import torch


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument('--net', default='resnet18', type=str)
    parser.add_argument('--model', default='lc', type=str)
    parser.add_argument('--dataset', default='ucf101', type=str)
    parser.add_argument('--split', default=1, type=int)
    parser.add_argument('--seq_len', default=5, type=int)
    parser.add_argument('--num_seq', default=8, type=int)
    parser.add_argument('--num_class', default=101, type=int)
    parser.add_argument('--dropout', default=0.5, type=float)
    parser.add_argument('--ds', default=3, type=int)
    parser.add_argument('--batch_size', default=4, type=int)
    parser.add_argument('--lr', default=1e-3, type=float)
    parser.add_argument('--wd', default=1e-3, type=float, help='weight decay')
    parser.add_argument('--resume', default='', type=str)
    parser.add_argument('--pretrain', default='random', type=str)
    parser.add_argument('--test', default='', type=str)
    parser.add_argument('--epochs', default=10, type=int, help='number of total epochs to run')
    parser.add_argument('--start-epoch', default=0, type=int, help='manual epoch number (useful on restarts)')
    parser.add_argument('--gpu', default='0,1', type=str)
    parser.add_argument('--print_freq', default=5, type=int)
    parser.add_argument('--reset_lr', action='store_true', help='Reset learning rate when resume training?')
    parser.add_argument('--train_what', default='last', type=str, help='Train what parameters?')
    parser.add_argument('--prefix', default='tmp', type=str)
    parser.add_argument('--img_dim', default=128, type=int)
    # additions of this build
    parser.add_argument('--synthetic', default=20, type=int, help='synthetic batches per epoch (the only data source here)')
    parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    parser.add_argument('--save_dir', default='', type=str)
    return parser


def lr_multiplier(epoch: int, gamma: float, milestones, repeat: int) -> float:
    """MultiStepLR_Restart_Multiplier (eval/test.py:408-423): gamma^(milestones passed) inside a cycle of max(milestones) epochs,
    restarting `repeat` times, then pinned at the last decay level"""
    period = max(milestones)
    if epoch // period >= repeat:
        return gamma ** (len(milestones) - 1)
    return gamma ** sum(1 for m in milestones if epoch % period >= m)


def lr_milestones(dataset: str, img_dim: int):
    """eval/test.py:93-99"""
    if dataset == 'hmdb51':
        return [150, 250, 300]
    if dataset == 'ucf101':
        return [300, 400, 500] if img_dim == 224 else [60, 80, 100]
    raise ValueError('no learning-rate schedule for dataset %r (eval/test.py:93-99 defines ucf101 and hmdb51)' % dataset)


def _worker(rank: int, world: int, args, port: int):
    gpus = [int(g) for g in str(args.gpu).split(',') if g != '']
    sim = getattr(args, '_simulator', None)   # tests only (CPU tier): the host-side SIMT simulator handle; the command line cannot set it
    dev = torch.device('cpu') if sim is not None else torch.device('cuda', gpus[rank] if world > 1 else gpus[0])
    if sim is None:
        torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world, device_id=dev)
    from . import checkpoint as ckpt
    from .lc import LC, LCEngine
    from .parallel import make_allreduce

    if args.dataset == 'ucf101':
        args.num_class = 101   # eval/test.py:55-56
    elif args.dataset == 'hmdb51':
        args.num_class = 51
    if args.model != 'lc':
        raise ValueError('wrong model!')
    if args.batch_size % world:
        raise ValueError('batch_size must be divisible by the number of GPUs')
    per_gpu = args.batch_size // world
    cdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    from .plan import LAYER_WIDTH
    widths = getattr(args, '_widths', None) or LAYER_WIDTH
    eng = LCEngine(args.net, args.img_dim, args.num_seq, args.seq_len, per_gpu, dev, cdt, widths, lib=sim, lr=args.lr, wd=args.wd,
                   dropout=args.dropout, num_class=args.num_class, seed=666 + rank)  # model_3d_lc.py:16 seeds 666
    init = LC(args.img_dim, args.num_seq, args.seq_len, args.net, args.dropout, args.num_class, widths=widths, seed=0)
    eng.load_params({k: v.detach() for k, v in init.state_dict().items()})
    log = print if rank == 0 else (lambda *a, **k: None)
    if args.train_what == 'ft':
        log("=> finetune backbone with smaller lr  [the reference's name filter ('resnet' / 'rnn') matches no LC parameter: one lr]")
    num_epoch, best_acc, iteration = 0, 0.0, 0
    base_lr = args.lr
    milestones = lr_milestones(args.dataset, args.img_dim)
    eng.lr = base_lr * lr_multiplier(0, 0.1, milestones, 1)  # LambdaLR's constructor step
    for path, what in ((args.test, 'test'), (args.resume, 'resume'), (args.pretrain, 'pretrain')):
        if not path or path == 'random' or (what == 'pretrain' and args.resume):
            continue
        if not os.path.isfile(path):
            if what == 'test':
                raise ValueError()  # eval/test.py:121-122
            log("=> no checkpoint found at '{}'".format(path))
            continue
        ck = torch.load(path, map_location='cpu', weights_only=False)
        # --resume: nn.Module.load_state_dict, strict in both directions (eval/test.py:145); --test: strict, falling back to the
        # key intersection with a warning (:113-116); --pretrain: key intersection (neq_load_customized, :160)
        try:
            missing, unexpected = ckpt.load_model_state(eng, ck['state_dict'], strict=what != 'pretrain')
        except RuntimeError:
            if what != 'test':
                raise
            log('=> [Warning]: weight structure is not equal to test model; Use non-equal load ==')
            missing, unexpected = ckpt.load_model_state(eng, ck['state_dict'], strict=False)
        log("=> loaded {} checkpoint '{}' (epoch {}; {} keys not in the file, {} keys of the file unused)".format(
            what, path, ck.get('epoch', 0), len(missing), len(unexpected)))
        num_epoch = ck.get('epoch', 0)
        if what == 'resume':
            args.start_epoch = ck['epoch']
            best_acc = float(ck.get('best_acc', 0.0))
            iteration = int(ck.get('iteration', 0))
            if not args.reset_lr and 'optimizer' in ck:
                ckpt.load_optimizer_state(eng, ck['optimizer'])  # restores the group's lr as optimizer.load_state_dict does
    allreduce = make_allreduce(dist, world)
    gen = torch.Generator(dev).manual_seed(1000 + rank)
    shape = (per_gpu, args.num_seq, 3, args.seq_len, args.img_dim, args.img_dim)

    def batch():
        return (torch.randn(shape, device=dev, generator=gen),
                torch.randint(0, args.num_class, (per_gpu,), device=dev, generator=gen))

    def reduce(res):
        vals = res.clone()
        if dist is not None:
            dist.all_reduce(vals, op=dist.ReduceOp.AVG)
        return vals.cpu().tolist()  # one packed D2H per logged step

    if args.test:  # eval/test.py:306-343: eval mode; softmax averaged over the clip's sequences, top-1 / top-5
        top1 = top5 = loss_sum = 0.0
        for _ in range(args.synthetic):
            x, y = batch()
            out, _ = eng.forward(x, y, train=False)
            prob = torch.softmax(out, 2).mean((0, 1), keepdim=False).view(1, -1)
            tgt = y[:1]
            top = prob.topk(5, 1).indices
            top1 += float((top[:, :1] == tgt[:, None]).any())
            top5 += float((top == tgt[:, None]).any())
            loss_sum += torch.nn.functional.cross_entropy(out.mean((0, 1)).view(1, -1), tgt).item()
        n = max(args.synthetic, 1)
        log('Loss {:.4f}\t Acc top1: {:.4f} Acc top5: {:.4f} \t'.format(loss_sum / n, top1 / n, top5 / n))
        log('(test checkpoint epoch {})'.format(num_epoch))
    else:
        for epoch in range(args.start_epoch, args.epochs):
            for idx in range(args.synthetic):  # train(): eval/test.py:218-271
                x, y = batch()
                res = eng.train_step(x, y, allreduce=allreduce)
                if idx % args.print_freq == 0:
                    loss, acc = reduce(res)
                    log('Epoch: [{0}][{1}/{2}]\t Loss {3:.4f}\t Acc: {4:.4f}\t lr {5:g}'.format(epoch, idx, args.synthetic, loss, acc, eng.lr),
                        flush=True)
                    iteration += 1  # advanced on logged steps only, as the reference does (eval/test.py:262-270)
            vl = va = 0.0
            for idx in range(max(args.synthetic // 4, 1)):  # validate(): eval/test.py:273-304 (eval mode, running statistics)
                x, y = batch()
                eng.forward(x, y, train=False)
                loss, acc = reduce(eng.result)
                vl += loss
                va += acc
            nv = max(args.synthetic // 4, 1)
            val_acc = va / nv
            log('Loss {:.4f}\t Acc: {:.4f} \t'.format(vl / nv, val_acc), flush=True)
            eng.lr = base_lr * lr_multiplier(epoch, 0.1, milestones, 1)  # scheduler.step(epoch), eval/test.py:197
            is_best = val_acc > best_acc  # eval/test.py:205-214
            best_acc = max(val_acc, best_acc)
            if rank == 0 and args.save_dir:
                os.makedirs(args.save_dir, exist_ok=True)
                state = {'epoch': epoch + 1, 'net': args.net, 'state_dict': {'module.' + k: v.cpu() for k, v in eng.state_dict().items()},
                         'best_acc': best_acc, 'optimizer': ckpt.optimizer_state_dict(eng), 'iteration': iteration}
                ckpt.save_checkpoint(state, is_best, filename=os.path.join(args.save_dir, 'epoch%s.pth.tar' % str(epoch + 1)))
        log('Training from ep %d to ep %d finished' % (args.start_epoch, args.epochs))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None, _simulator=None, _widths=None):
    args = build_parser().parse_args(argv)
    args._simulator, args._widths = _simulator, _widths   # tests/test_entries.py: the CPU tier runs the entry on the simulator
    gpus = [g for g in str(args.gpu).split(',') if g != '']
    world = max(len(gpus), 1)
    if _simulator is not None and world != 1:
        raise ValueError('the simulator runs one rank')
    if world == 1:
        _worker(0, 1, args, 0)
    else:
        import torch.multiprocessing as mp
        mp.spawn(_worker, args=(world, args, 29500 + (os.getpid() % 2000)), nprocs=world, join=True)


if __name__ == '__main__':
    main()
