#!/usr/bin/env python
"""Training entry point for the HOT PATH of the reference's train.py: same flags, config autoscaling, Adam param groups,
LR warm-up / steps, `<name>_<epoch>_<iter>.pth` checkpoints, resume / interrupt handling, 100-iteration console line.
Scope: the training step (model + loss + optimizer + data-parallel exchange) and the validation pass (eval.py: depth errors +
box / mask AP every `--validation_epoch` epochs over `--validation_size` frames and once after the last iteration, reference
train.py:395-402,440-448).  The annotated-dataset readers / augmentations (cv2 + pycocotools) and tensorboard logging are NOT
part of this build: real datasets exit with a message, `--log_folder` / `--batch_alloc` raise when changed from their
defaults, `--dataset synthetic` feeds seeded batches with the reference's batch contract.

What differs is the machinery underneath:
  * the model and loss run on the HIP kernels (planerecnet_amd), the device comes from `cfg.device`;
  * multi-GPU is one process per GPU (`python -m torch.distributed.run --nproc-per-node N train.py ...`): each rank takes
    batch_size // N samples, BatchNorm statistics stay per rank (as in the reference's DataParallel replicas), gradients
    are mean-all-reduced over RCCL on a side stream overlapped with backward (planerecnet_amd/parallel.py), the loss that
    is logged is the mean over ranks (reference train.py:348), and the "skip the step on a non-finite loss" decision
    (train.py:353) is taken collectively so ranks cannot diverge;
  * GT-only loss preparation (SOLOv2 target assignment, virtual-normal triplet sampling) runs on the device one batch ahead of the
    step (planerecnet_amd/targets.py; `--target_prep workers`: the host worker processes of losses.TargetPrefetcher, which draw the
    triplets from numpy's global stream like the reference);
  * `--dataset synthetic` (must be given explicitly: there is no silent fallback) feeds seeded synthetic batches with the
    reference's batch contract (data/datasets.py:54-57,250-273) -- the ScanNet / NYU readers need cv2 + pycocotools and are
    outside this hot-path build.
"""
import argparse
import datetime
import collections
import math
import os
import random
import sys
import time

import numpy as np

# Hardware queues the HIP runtime multiplexes this process's streams onto (compute stream, weight-gradient / exchange side stream, RCCL's
# own streams with N > 1).  3 is the measured optimum with ONE rank; with N > 1 it is unmeasured, hence an explicit
# knob: `--hw-queues K` (or GPU_MAX_HW_QUEUES in the environment) -- read here because it must be set before the runtime initialises.
for _i, _a in enumerate(sys.argv):
    if _a == "--hw-queues" and _i + 1 < len(sys.argv):
        os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[_i + 1]
    elif _a.startswith("--hw-queues="):
        os.environ["GPU_MAX_HW_QUEUES"] = _a.split("=", 1)[1]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
import torch  # noqa: E402
import torch.distributed as dist

import eval as eval_script  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg, set_dataset  # noqa: E402
from planerecnet_amd.datasets import SyntheticPlaneDataset, detection_collate  # noqa: E402
from planerecnet_amd.utils import MovingAverage, SavePath

parser = argparse.ArgumentParser(description="PlaneRecNet Training Script (MI355X)")
parser.add_argument("--dataset", default=None, type=str, help="Override the config's dataset ('synthetic' for seeded synthetic batches).")
parser.add_argument("--config", default="PlaneRecNet_50_config", help="The config object to use.")
parser.add_argument("--save_folder", default="./weights/", help="Directory for saving checkpoint models.")
parser.add_argument("--log_folder", default="./logs/", help="Directory for saving logs.")
parser.add_argument("--backbone_folder", default="./weights/", help="Directory for loading Backbone.")
parser.add_argument("--resume", default=None, type=str, help='Checkpoint to resume from ("interrupt" / "latest" / path).')
parser.add_argument("--start_iter", default=-1, type=int, help="Resume at this iteration (-1: parse it from the file name).")
parser.add_argument("--validation_size", default=2000, type=int)
parser.add_argument("--validation_epoch", default=1, type=int)
parser.add_argument("--no_tensorboard", dest="no_tensorboard", action="store_true")
parser.add_argument("--no_autoscale", dest="autoscale", action="store_false")
parser.add_argument("--reproductablity", dest="reproductablity", action="store_true")
parser.add_argument("--batch_size", default=8, type=int, help="GLOBAL batch size (split evenly over the ranks).")
parser.add_argument("--lr", "--learning_rate", default=None, type=float)
parser.add_argument("--momentum", default=None, type=float)
parser.add_argument("--decay", "--weight_decay", default=None, type=float)
parser.add_argument("--gamma", default=None, type=float)
parser.add_argument("--num_workers", default=2, type=int)
parser.add_argument("--save_interval", default=12500, type=int)
parser.add_argument("--keep_latest", dest="keep_latest", action="store_true")
parser.add_argument("--keep_latest_interval", default=10000, type=int)
parser.add_argument("--no_interrupt", dest="interrupt", action="store_false")
parser.add_argument("--batch_alloc", default=None, type=str, help="Accepted for CLI compatibility; ranks always take equal shares.")
parser.add_argument("--max_iter", default=None, type=int, help="(extension) stop after this many iterations.")
parser.add_argument("--synthetic_size", default=64, type=int, help="(extension) samples per synthetic epoch.")
parser.add_argument("--hw-queues", dest="hw_queues", default=None, type=int,
                    help="(extension) GPU_MAX_HW_QUEUES for this run (default 3; applied before the HIP runtime starts).")
parser.add_argument("--target_prep", default="device", choices=("device", "workers"),
                    help="(extension) where the GT-only part of the loss is prepared: HIP kernels a batch ahead, or host worker processes.")
parser.add_argument("--seed", default=0, type=int, help="Run seed of the device triplet sampler (Philox key; the rank is mixed in).")
parser.add_argument("--triplet_sampler", default="philox", choices=("philox", "numpy"),
                    help="(extension, --target_prep device) virtual-normal triplet ranks: drawn on the device, or from numpy's global stream like the reference.")
parser.add_argument("--synthetic_val_size", default=8, type=int, help="(extension) frames in the synthetic validation set.")
parser.set_defaults(keep_latest=False, interrupt=True, autoscale=True)

LOSS_TYPES = ["ins", "lav", "cat", "dpt", "pln"]


class NetLoss(torch.nn.Module):
    """net + criterion as one unit of work (reference train.py:128-150), with the GT-only targets passed in."""

    def __init__(self, net, criterion):
        super().__init__()
        self.net, self.criterion = net, criterion

    def forward(self, images, gt_instances, gt_depths, targets=None):
        return self.criterion(self.net, *self.net(images), gt_instances, gt_depths, targets=targets)


def compute_validation_metrics(epoch, iteration, prn_net, val_dataset, eval_nums=-1, rank=0, world=1):
    """Reference train.py:440-448.  Rank 0 evaluates its replica (all replicas are identical), the others wait."""
    if rank == 0:
        with torch.no_grad():
            prn_net.eval()
            print()
            print("Computing validation metrics (this may take a while)...", flush=True)
            eval_script.evaluate(prn_net, val_dataset, during_training=True, eval_nums=eval_nums)
            prn_net.train()
    if world > 1:
        dist.barrier()


def set_lr(optimizer, new_lr):
    for group in optimizer.param_groups:      # like the reference (train.py:415-417) this flattens the per-group multipliers (quirk Q7)
        group["lr"] = new_lr


def main():
    args = parser.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    set_cfg(args.config)
    if args.dataset not in (None, "synthetic"):
        set_dataset(args.dataset)
    if args.autoscale and args.batch_size != 8:
        factor = args.batch_size / 8
        if rank == 0:
            print("Scaling parameters by %.2f to account for a batch size of %d." % (factor, args.batch_size))
        cfg.lr *= factor
        cfg.max_iter //= factor
        cfg.lr_steps = [x // factor for x in cfg.lr_steps]
    for name in ("lr", "decay", "gamma", "momentum"):
        if getattr(args, name) is None:
            setattr(args, name, getattr(cfg, name))
    if args.max_iter is not None:
        cfg.max_iter = args.max_iter
    # Flags of the reference's CLI whose machinery (tensorboardX logging, DataParallel batch allocation) is outside this
    # hot-path build: accepted only at their defaults / off-values -- asking for them is an error, not a silent no-op.
    ignored = []
    if args.log_folder != "./logs/":
        ignored.append("--log_folder (no tensorboard writer)")
    if args.batch_alloc is not None:
        ignored.append("--batch_alloc (ranks always take equal shares of the batch)")
    if ignored:
        raise SystemExit("train.py: not supported by this build:\n  " + "\n  ".join(ignored))
    if rank == 0 and not args.no_tensorboard:
        print("Note: tensorboard logging is not part of this build (console log only); pass --no_tensorboard to silence this note.")
    if rank == 0 and (args.decay != cfg.decay or args.momentum != cfg.momentum):
        print("Note: --decay / --momentum are parsed but never reach Adam -- exactly as in the reference (train.py:251-256, quirk Q7).")
    if not torch.cuda.is_available():
        raise SystemExit("No GPUs detected. The HIP path has no CPU fallback.")
    if args.batch_size % world:
        raise SystemExit("batch_size must be divisible by the number of ranks")
    per_rank = args.batch_size // world
    if per_rank < 6:
        if rank == 0:
            print("Per-GPU batch size is less than the recommended limit for batch norm. Disabling batch norm.")
        cfg.freeze_bn = True
    if args.reproductablity:
        for seed_fn in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
            seed_fn(0)
    if os.environ.get("PRN_ONE_DEVICE"):                   # validation aid: N ranks time-share GPU 0 (with PRN_DIST_BACKEND=gloo; RCCL
        local = 0                                          # refuses two ranks on one device) -- exercises the N > 1 code path on a 1-GPU box
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg.device = str(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PRN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.set_num_threads(4)

    from planerecnet_amd import ops, timer
    from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher
    from planerecnet_amd.optim import FusedAdam
    from planerecnet_amd.parallel import GradAllReduce, all_reduce_mean_scalars
    from planerecnet_amd.planerecnet import PlaneRecNet
    from planerecnet_amd.staging import FrameStager

    os.makedirs(args.save_folder, exist_ok=True)
    if args.dataset != "synthetic":
        # no silent fallback: a run (and its validation tables) on seeded noise must not pass for a run on the configured dataset
        raise SystemExit("train.py: the annotated dataset readers (cv2 + pycocotools) are outside this build; pass --dataset synthetic "
                         "to train / validate on seeded synthetic batches.")
    print("NOTE: training on SYNTHETIC batches (--dataset synthetic); validation metrics are plumbing checks, not accuracy figures.")
    dataset = SyntheticPlaneDataset(args.synthetic_size)
    val_dataset = SyntheticPlaneDataset(args.synthetic_val_size)
    eval_script.parse_args(["--no_bar"])                   # (reference train.py:436-437)

    torch.manual_seed(0)
    prn_net = PlaneRecNet(cfg).train()
    timer.disable_all()
    if args.resume == "interrupt":
        args.resume = SavePath.get_interrupt(args.save_folder)
    elif args.resume == "latest":
        args.resume = SavePath.get_latest(args.save_folder, cfg.name)
    if args.resume is not None:
        print("Resuming training, loading {}...".format(args.resume))
        prn_net.load_weights(args.resume)
        if args.start_iter == -1:
            args.start_iter = SavePath.from_str(args.resume).iteration
    else:
        backbone_path = os.path.join(args.backbone_folder, cfg.backbone.path)
        if os.path.exists(backbone_path):
            prn_net.init_weights(backbone_path=backbone_path)
        else:
            if rank == 0:
                print("Pretrained backbone %s not found: random backbone init." % backbone_path)
            prn_net.init_head_weights()
    prn_net = prn_net.to(dev)
    criterion = PlaneRecNetLoss().to(dev)
    net = NetLoss(prn_net, criterion)
    optimizer = FusedAdam([
        {"params": prn_net.backbone.parameters(), "lr": 5 * args.lr}, {"params": prn_net.fpn.parameters(), "lr": args.lr},
        {"params": prn_net.inst_head.parameters(), "lr": args.lr}, {"params": prn_net.mask_head.parameters(), "lr": args.lr},
        {"params": prn_net.depth_decoder.parameters(), "lr": 2 * args.lr}], lr=args.lr)      # optim.Adam of train.py:251-256 as one launch
    exchange = GradAllReduce([p for p in prn_net.parameters()])
    optimizer.exchange = exchange                          # parameters no rank had a gradient for are skipped like a .grad of None (train.py:362)
    ops.set_wgrad_async(True)          # weight gradients on a side stream; joined by ops.wgrad_join() after every backward()

    # BN-safe warm-up forward with frozen statistics (reference train.py:270-272)
    if not cfg.freeze_bn:
        prn_net.freeze_bn()
    with torch.no_grad():
        prn_net(torch.zeros(1, 3, cfg.max_size, cfg.max_size, device=dev))
    if not cfg.freeze_bn:
        prn_net.freeze_bn(True)

    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(dataset, per_rank, num_workers=args.num_workers, shuffle=sampler is None, sampler=sampler,
                                         collate_fn=detection_collate, pin_memory=False, drop_last=True)      # (frames are staged through our own page-locked buffers, see stage())
    iteration = max(args.start_iter, 0)
    epoch_size = max(len(loader), 1)
    num_epochs = math.ceil(cfg.max_iter / epoch_size)
    step_index, last_time = 0, time.time()
    time_avg, loss_avgs = MovingAverage(), {k: MovingAverage(100) for k in LOSS_TYPES}
    save_path = lambda epoch, it: SavePath(cfg.name, epoch, it).get_path(root=args.save_folder)
    if args.target_prep == "device":
        from planerecnet_amd.targets import DeviceTargetBuilder
        # Philox key = (run seed, rank); counter = (triplet, global iteration): ranks and runs draw different streams, a resumed run continues its own
        prefetch = DeviceTargetBuilder(criterion, sampler=args.triplet_sampler, seed=(int(args.seed) << 32) | rank, first_call=iteration)
    else:
        prefetch = TargetPrefetcher(criterion)
    stager = FrameStager(dev)
    # device-side "skip the update on a non-finite loss": needs an optimizer whose step takes `found_inf` (fused Adam does)
    device_skip = bool(getattr(optimizer, "_step_supports_amp_scaling", False)) and dev.type == "cuda" and not os.environ.get("PRN_TRAIN_SYNC_LOSS")
    if device_skip:
        optimizer.grad_scale = None
    pending_stats = []

    def flush_stats():
        """Bring the queued per-step loss values to the host (one synchronisation) and feed the moving averages in order."""
        if not pending_stats:
            return
        vals = torch.stack([t for _, t in pending_stats]).tolist()
        for (names, _), row in zip(pending_stats, vals):
            for k, v in zip(names, row):
                loss_avgs[k].add(v)
        del pending_stats[:]

    if rank == 0:
        print("Begin training!\n")
    epoch = 0
    try:
        for epoch in range(num_epochs):
            if (epoch + 1) * epoch_size < iteration:
                continue
            if sampler is not None:
                sampler.set_epoch(epoch)
            it = iter(loader)
            prefetch.discard()                             # (batches of the previous epoch that were prepared but not used)
            ahead = collections.deque()                    # batches whose GT-only targets are being prepared (two in flight)

            def refill():
                b_ = next(it, None)
                if b_ is not None:
                    ahead.append(b_)
                    prefetch.submit(b_[1], tuple(b_[0][0].shape[-2:]))

            def stage():
                """Uploads of the next batch (images, depth, targets): issued one step EARLY, right after the current
                step's backward has been enqueued, so that a step starts with the forward pass even though every
                step ends in a host synchronisation (the all-reduced loss values)."""
                if not ahead:
                    return None
                images_, inst_, depths_ = ahead.popleft()
                # Frames go to HBM on the weight-gradient side stream through rotating page-locked buffers (staging.FrameStager);
                # the loader's batches stay in the workers' shared memory (the GT goes on to the target workers as handles).
                main = torch.cuda.current_stream()
                x_, d_, ev_ = stager.upload(images_, depths_, main, ops._side_stream(dev, main))
                t_ = prefetch.get(d_, dev, overlap=True)
                refill()
                return x_, d_, t_, inst_, ev_

            refill()
            refill()
            cur = stage()
            while cur is not None:
                x, d, targets, gt_instances, inputs_ready = cur
                if iteration == (epoch + 1) * epoch_size or iteration == cfg.max_iter:
                    break
                torch.cuda.current_stream().wait_event(inputs_ready)
                changed = [c for c in cfg.delayed_settings if iteration >= c[0]]
                for c in changed:
                    cfg.replace(c[1])
                    for avg in loss_avgs.values():
                        avg.reset()
                if changed:
                    cfg.delayed_settings = [x_ for x_ in cfg.delayed_settings if x_[0] > iteration]
                if cfg.lr_warmup_until > 0 and iteration <= cfg.lr_warmup_until:
                    set_lr(optimizer, (args.lr - cfg.lr_warmup_init) * (iteration / cfg.lr_warmup_until) + cfg.lr_warmup_init)
                while step_index < len(cfg.lr_steps) and iteration >= cfg.lr_steps[step_index]:
                    step_index += 1
                    set_lr(optimizer, args.lr * (args.gamma ** step_index))

                optimizer.zero_grad(set_to_none=True)
                losses = net(x, gt_instances, d, targets=targets)
                loss = sum(losses[k].sum() for k in losses)
                loss.backward()
                ops.wgrad_join()                               # deferred weight gradients (ops.set_wgrad_async)
                exchange.finish()
                cur = stage()                                  # next batch: uploads behind this step's work, before the sync below
                shown = [k for k in LOSS_TYPES if k in losses]
                stats_dev = all_reduce_mean_scalars([losses[k].detach().sum() for k in shown] + [loss.detach()], dev)
                del loss, losses                           # drop the step's autograd graph now, not when the next step's loss replaces it (+5 ms/step)
                # The reference skips the update when the loss is not finite (train.py:353) and logs every step's losses; both
                # read the loss on the host, i.e. stall the GPU once per step (74 vs 59 ms/step here).  The skip is decided ON
                # THE DEVICE instead (the fused optimizer's `found_inf` input, the one torch.amp.GradScaler drives; collective:
                # every rank sees the same all-reduced mean) and the logged values are fetched in batches.
                if device_skip:
                    optimizer.found_inf = (~torch.isfinite(stats_dev[-1])).to(torch.float32).reshape(())
                    optimizer.step()
                    pending_stats.append((shown, stats_dev))
                    if len(pending_stats) >= 100 or iteration % 100 == 0:
                        flush_stats()
                else:
                    stats = stats_dev.tolist()
                    if math.isfinite(stats[-1]):           # collective decision: every rank sees the same mean
                        optimizer.step()
                    for k, v in zip(shown, stats):
                        loss_avgs[k].add(v)
                now = time.time()
                elapsed, last_time = now - last_time, now
                if iteration != args.start_iter:
                    time_avg.add(elapsed)
                if iteration % 100 == 0 and rank == 0:
                    eta = str(datetime.timedelta(seconds=(cfg.max_iter - iteration) * time_avg.get_avg())).split(".")[0]
                    total = sum(loss_avgs[k].get_avg() for k in shown)
                    labels = sum([[k, loss_avgs[k].get_avg()] for k in shown], [])
                    print(("[%3d] %7d ||" + (" %s: %.3f |" * len(shown)) + " total: %.3f || ETA: %s || time/batch: %.3fs")
                          % tuple([epoch, iteration] + labels + [total, eta, elapsed]), flush=True)
                iteration += 1
                if iteration % args.save_interval == 0 and iteration != args.start_iter:
                    flush_stats()
                if iteration % args.save_interval == 0 and iteration != args.start_iter and rank == 0:
                    latest = SavePath.get_latest(args.save_folder, cfg.name) if args.keep_latest else None
                    print("Saving state, iter:", iteration)
                    prn_net.save_weights(save_path(epoch, iteration))
                    if latest is not None and (args.keep_latest_interval <= 0 or iteration % args.keep_latest_interval != args.save_interval):
                        print("Deleting old save...")
                        os.remove(latest)
            if args.validation_epoch > 0 and epoch % args.validation_epoch == 0 and iteration > 0 and epoch < num_epochs - 2:
                flush_stats()                              # (no validation at iteration 0 or in the last epochs: reference train.py:396-399)
                compute_validation_metrics(epoch, iteration, prn_net, val_dataset, args.validation_size, rank, world)
            if iteration >= cfg.max_iter:
                break
        flush_stats()
        compute_validation_metrics(epoch, iteration, prn_net, val_dataset, -1, rank, world)       # after training (reference train.py:401-402)
    except KeyboardInterrupt:
        if args.interrupt and rank == 0:
            print("Stopping early. Saving network...")
            SavePath.remove_interrupt(args.save_folder)
            prn_net.save_weights(save_path(epoch, repr(iteration) + "_interrupt"))
        raise SystemExit
    finally:
        prefetch.close()
    if rank == 0:
        prn_net.save_weights(save_path(epoch, iteration))
    if world > 1:
        # replica consistency: every rank applied the same all-reduced gradients to the same initial weights, so the
        # parameters must still agree bit for bit (the reference's DataParallel re-broadcasts rank 0's copy every step instead)
        chk = torch.stack([p.detach().double().sum() for p in prn_net.parameters()]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            print("Replica check over %d ranks: parameter checksum spread %.3e (%s)" % (world, float(hi - lo), "identical" if float(hi - lo) == 0.0 else "DIVERGED"))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
