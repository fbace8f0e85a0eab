#!/usr/bin/env python3
"""Importance Assessment — drop-in for the reference's importance_generation.py (:8-60).

Same flags (--dataset --data_dir --batch_size --pretrain_dir --limit --net), same output:
importance_score/<net>_limit<L>/*.npy under the current directory, which prune_cifar10.py /
prune_imagenet.py / prune_u2netp.py read through --imp_score. The DCT+score arithmetic runs
in libdctscore (HIP, gfx950); a GPU is required.

Extra, opt-in flags: --synthetic (seeded synthetic batches; also lifts the need for a
checkpoint), --input_size, --seed, --single_sweep, --device_accumulate, --deferred. Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node G importance_generation.py ...` — hook points
are sharded over the ranks and rank 0 writes the files.
"""
import argparse
import os
from collections import OrderedDict

import torch

from dct_pruning_amd import harness, nets


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Importance Assessment")
    parser.add_argument("--dataset", type=str, default="cifar10", choices=("cifar10", "imagenet", "DUTS", "synthetic"),
                        help="dataset")
    parser.add_argument("--data_dir", type=str, default="./data", help="path to dataset")
    parser.add_argument("--batch_size", type=int, default=128, help="batch size")
    parser.add_argument("--pretrain_dir", type=str, default="checkpoints/googlenet.pt",
                        help="load the model from the specified checkpoint")
    parser.add_argument("--limit", type=int, default=5, help="The num of batch to get importence score.")
    parser.add_argument("--net", type=str, default="googlenet",
                        choices=("resnet_50", "vgg_16_bn", "resnet_56", "resnet_110", "densenet_40", "googlenet", "u2netp"),
                        help="net type")
    parser.add_argument("--synthetic", action="store_true", help="seeded synthetic batches of the dataset's shape")
    parser.add_argument("--input_size", type=int, default=None, help="override H=W of synthetic inputs")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--single_sweep", action="store_true", help="score every hook point in one sweep")
    parser.add_argument("--device_accumulate", action="store_true", help="keep the running mean on the GPU")
    parser.add_argument("--deferred", action="store_true",
                        help="single sweep, one scoring launch per tile shape per batch (implies the two above)")
    return parser.parse_args(argv)


def load_checkpoint(net, args):
    """The five state-dict layouts of importance_generation.py:25-53."""
    ckpt = torch.load(args.pretrain_dir, map_location="cpu", weights_only=True)
    if args.net == "u2netp":
        own = net.state_dict()
        own.update({k: v for k, v in ckpt.items() if k in own})
        net.load_state_dict(own)
    elif args.net == "resnet_50":
        net.load_state_dict(ckpt)
    elif args.net in ("densenet_40", "resnet_110"):
        net.load_state_dict(OrderedDict((k.replace("module.", ""), v) for k, v in ckpt["state_dict"].items()))
    else:
        net.load_state_dict(ckpt["state_dict"])


def main(argv=None):
    # before the first HIP call: this pool's driver only supports dmabuf IPC, RCCL fails with hipIpcGetMemHandle otherwise
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("importance_generation.py needs a GPU: the score path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DCTS_REHEARSE=1: every rank on cuda:0 with a gloo group (to rehearse the sharded path on a
    # one-GPU box); normal runs use one GPU per rank and RCCL
    rehearse = os.environ.get("DCTS_REHEARSE") == "1"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from dct_pruning_amd import sharding
        # bounded bring-up: a rank that cannot reach the others prints {"error": ...} and exits 3 (sharding.py)
        sharding.init_process_group("gloo" if rehearse else "nccl", device=dev, what="importance_generation.py")

    torch.manual_seed(args.seed)
    net = nets.get_network(args.net)
    if args.pretrain_dir and os.path.isfile(args.pretrain_dir):
        print("==> Resuming from checkpoint..")
        load_checkpoint(net, args)
        print("Completed! ")
    elif args.synthetic:
        print("==> --synthetic without a checkpoint: random-init weights (seed %d)" % args.seed)
    else:
        print("please speicify a pretrain model ")
        raise NotImplementedError
    net = net.to(dev)

    harness.imp_score(net, args, single_sweep=args.single_sweep,
                      accumulate="device" if args.device_accumulate else "host", deferred=args.deferred)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
