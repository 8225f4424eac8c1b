#!/usr/bin/env python
"""Evaluation entry point (reference eval.py): the same flags, the same loop -- one frame at a time through the eval-mode
network, depth errors and box / mask AP at IoU .50 ... .95 accumulated over the frames, mAP table and mean depth errors
printed at the end -- and `evaluate(net, dataset, during_training, eval_nums)` for train.py's validation pass.

What differs underneath: the network and its post-process run on the HIP kernels; per frame, the depth errors are one fused
reduction and the two pairwise IoU matrices one popcount kernel (planerecnet_amd/metrics.py; include/prn.h:
prn_depth_metrics, prn_pairwise_iou); only the per-threshold matching and the AP integral -- a few hundred scalar
operations -- run on the host.

Outside this build (they need cv2 / pycocotools / tensorboardX): the annotated dataset readers -- frames come from
`--dataset synthetic` (planerecnet_amd/datasets.py) -- and `--autopsy`.  `--output_coco_json`, `--bbox_det_file` and
`--mask_det_file` are parsed and unused, exactly as in the reference.
"""
import argparse
import os
import random
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")       # before the HIP runtime initialises: see planerecnet_amd/__init__.py
import torch  # noqa: E402

from planerecnet_amd.config import cfg, set_cfg, set_dataset  # noqa: E402
from planerecnet_amd.utils import MovingAverage, SavePath  # noqa: E402

args = None


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="PlaneRecNet Evaluation (MI355X)")
    parser.add_argument("--trained_model", default=None, type=str,
                        help='Trained state_dict file path to open ("interrupt" / "latest" / path; none: random initialisation).')
    parser.add_argument("--top_k", default=100, type=int, help="Further restrict the number of predictions to parse")
    parser.add_argument("--score_threshold", default=0.15, type=float, help="Detections with a score under this threshold will not be considered.")
    parser.add_argument("--nms_mode", default="matrix", type=str, choices=["matrix", "mask"], help="Chose NMS type from matrix and mask nms.")
    parser.add_argument("--output_coco_json", dest="output_coco_json", action="store_true", help="(parsed, unused -- as in the reference)")
    parser.add_argument("--bbox_det_file", default="results/bbox_detections.json", type=str, help="(parsed, unused -- as in the reference)")
    parser.add_argument("--mask_det_file", default="results/mask_detections.json", type=str, help="(parsed, unused -- as in the reference)")
    parser.add_argument("--max_images", default=-1, type=int, help="The maximum number of images from the dataset to consider. Use -1 for all.")
    parser.add_argument("--config", default=None, help="The config object to use.")
    parser.add_argument("--no_bar", dest="no_bar", action="store_true", help="Do not output the status bar.")
    parser.add_argument("--autopsy", dest="autopsy", action="store_true", help="(tensorboard image log: not part of this build)")
    parser.add_argument("--dataset", default=None, type=str, help="Override the config's dataset ('synthetic' for seeded synthetic frames).")
    parser.add_argument("--synthetic_size", default=16, type=int, help="(extension) frames in the synthetic dataset.")
    global args
    args = parser.parse_args(argv)
    return args


def _bar(done, total, width=30):
    filled = int(round(width * done / max(total, 1)))
    return "[" + "#" * filled + " " * (width - filled) + "]"


def evaluate(net, dataset, during_training=False, eval_nums=-1):
    """Reference eval.py:63-130.  Returns (mAP table as calc_map returns it, {metric name: mean over the frames})."""
    from planerecnet_amd import metrics
    if args is None:
        parse_args(["--no_bar"])                                    # train.py's setup_eval (reference train.py:436-437)
    frame_times = MovingAverage()
    eval_nums = len(dataset) - 1 if eval_nums < 0 else min(eval_nums, len(dataset))     # (the reference's "- 1" included)
    print()
    indices = list(range(len(dataset)))
    random.shuffle(indices)
    indices = indices[:eval_nums]
    dev = next(net.parameters()).device
    infos, ap_data, all_maps = [], metrics.new_ap_data(), None
    try:
        for it, image_idx in enumerate(indices):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            image, gt_instances, gt_depth = dataset.pull_item(image_idx)
            result = net(image.unsqueeze(0).to(dev))[0]
            gt_masks, gt_boxes, gt_classes = (gt_instances[k].to(dev) for k in ("masks", "boxes", "classes"))
            infos.append([float(v) for v in metrics.compute_depth_metrics(result["pred_depth"], gt_depth.to(dev), median_scaling=True)])
            if result["pred_masks"] is not None:
                metrics.compute_segmentation_metrics(ap_data, gt_masks, gt_boxes, gt_classes, result["pred_masks"], result["pred_boxes"],
                                                     result["pred_classes"], result["pred_scores"])
            torch.cuda.synchronize(dev)
            if it > 1:                                              # the first frames include one-time setup (reference :103-106)
                frame_times.add(time.perf_counter() - t0)
            if not args.no_bar:
                fps = 1.0 / frame_times.get_avg() if it > 1 else 0
                print("\rProcessing Images  %s %6d / %6d (%5.2f%%)    %5.2f fps        "
                      % (_bar(it + 1, eval_nums), it + 1, eval_nums, (it + 1) / eval_nums * 100, fps), end="")
        all_maps = metrics.calc_map(ap_data)
        mean = np.asarray(infos, dtype=np.double).sum(axis=0) / max(len(infos), 1) if infos else np.zeros(8)
        print()
        print("Depth Metrics:")
        names = metrics.depth_metrics
        print(", ".join("{}: {:.5f}".format(n, v) for n, v in zip(names[:7], mean[:7])) + " \n{}: {:.5f}".format(names[7], mean[7]))
        return all_maps, dict(zip(names, mean.tolist()))
    except KeyboardInterrupt:
        print("Stopping...")
        return all_maps, {}


def main():
    parse_args()
    torch.set_num_threads(4)                                   # host-side tensor ops are small: a wide OpenMP team only adds fork / join latency (as train.py)
    if args.autopsy:
        raise SystemExit("eval.py: --autopsy writes tensorboard images (tensorboardX + cv2 colour maps); not part of this build")
    if args.trained_model == "interrupt":
        args.trained_model = SavePath.get_interrupt("weights/")
    elif args.trained_model == "latest":
        if args.config is None:
            raise SystemExit("eval.py: --trained_model latest needs --config")
        set_cfg(args.config)
        args.trained_model = SavePath.get_latest("weights/", cfg.name)
    if args.config is None:
        if args.trained_model is None:
            raise SystemExit("eval.py: give --config (or a --trained_model whose file name carries it)")
        args.config = SavePath.from_str(args.trained_model).model_name + "_config"
        print("Config not specified. Parsed %s from the file name.\n" % args.config)
    set_cfg(args.config)
    cfg.solov2.replace({"nms_type": args.nms_mode, "mask_thr": args.score_threshold, "update_thr": args.score_threshold, "top_k": args.top_k})
    if args.dataset not in (None, "synthetic"):
        set_dataset(args.dataset)
    if not torch.cuda.is_available():
        raise SystemExit("No GPUs detected. The HIP path has no CPU fallback.")
    from planerecnet_amd import timer
    from planerecnet_amd.datasets import SyntheticPlaneDataset
    from planerecnet_amd.planerecnet import PlaneRecNet
    if args.dataset != "synthetic":
        # never print metrics of seeded noise frames under a real dataset's name: the reference fails on a missing dataset too
        raise SystemExit("eval.py: the annotated dataset readers (cv2 + pycocotools) are outside this build; pass --dataset synthetic "
                         "to evaluate on seeded synthetic frames (the numbers then say nothing about a trained model's accuracy).")
    print("NOTE: evaluating on SYNTHETIC frames (--dataset synthetic): the metrics below are plumbing checks, not accuracy figures.")
    dataset = SyntheticPlaneDataset(args.synthetic_size)
    with torch.no_grad():
        os.makedirs("results", exist_ok=True)
        print("Loading model...", end="")
        torch.manual_seed(0)
        net = PlaneRecNet(cfg)
        if args.trained_model is not None:
            net.load_weights(args.trained_model)
        else:
            net.init_head_weights()
            print(" (no --trained_model: random initialisation)", end="")
        net.eval()
        timer.disable_all()
        print(" done.")
        cfg.device = "cuda:0"
        net = net.to("cuda:0")
        evaluate(net, dataset, during_training=False, eval_nums=args.max_images)


if __name__ == "__main__":
    main()
