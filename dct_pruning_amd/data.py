"""Data front-end of imp_score: counterpart of load_data (utils/common.py:57-161).

The score pass only enumerates the first `limit` batches of the training loader
(utils/common.py:314-316). `--synthetic` (or dataset == 'synthetic') gives a deterministic,
re-iterable loader of seeded batches with the dataset's shape — the reference's own loaders
shuffle without a seed (utils/common.py:73, :96-100, :157), so results are only comparable on
fixed batches. Real datasets go through torchvision exactly as in the reference when it is
installed; otherwise a clear error is raised (no silent substitution).
"""
import torch

INPUT_SHAPES = {"cifar10": (3, 32, 32), "imagenet": (3, 224, 224), "DUTS": (3, 288, 288)}
NET_DATASET = {"vgg_16_bn": "cifar10", "resnet_56": "cifar10", "resnet_110": "cifar10", "densenet_40": "cifar10",
               "googlenet": "cifar10", "resnet_50": "imagenet", "u2netp": "DUTS"}


class SyntheticLoader:
    """Yields `num_batches` seeded batches; every iteration yields the SAME batches.
    DUTS-style loaders yield {'image': x, 'label': y} dicts (utils/common.py:329)."""

    def __init__(self, shape, batch_size, num_batches, seed=0, as_dict=False):
        self.shape, self.batch_size, self.num_batches = tuple(shape), batch_size, num_batches
        self.seed, self.as_dict = seed, as_dict

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        for b in range(self.num_batches):
            g = torch.Generator().manual_seed(self.seed * 100003 + b)
            x = torch.randn((self.batch_size,) + self.shape, generator=g)
            y = torch.randint(0, 10, (self.batch_size,), generator=g)
            yield ({"image": x, "label": y} if self.as_dict else (x, y))


def load_data(args):
    dataset = getattr(args, "dataset", None)
    synthetic = getattr(args, "synthetic", False) or dataset == "synthetic"
    if synthetic:
        base = dataset if dataset in INPUT_SHAPES else NET_DATASET.get(getattr(args, "net", ""), "cifar10")
        shape = list(INPUT_SHAPES[base])
        size = getattr(args, "input_size", None)
        if size:
            shape[1] = shape[2] = int(size)
        loader = SyntheticLoader(shape, args.batch_size, getattr(args, "limit", 5) + 1,
                                 seed=getattr(args, "seed", 0), as_dict=(base == "DUTS"))
        return loader, None
    try:
        import torchvision
        from torchvision import datasets, transforms
    except ImportError as exc:
        raise RuntimeError(
            "dataset %r needs torchvision, which is not installed here; use --synthetic for "
            "seeded synthetic batches of the same shape" % (dataset,)) from exc
    import os
    from torch.utils.data import DataLoader
    if dataset == "cifar10":
        tf = transforms.Compose([transforms.RandomCrop(32, padding=4), transforms.RandomHorizontalFlip(),
                                 transforms.ToTensor(),
                                 transforms.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))])
        train = torchvision.datasets.CIFAR10(root=args.data_dir, train=True, download=False, transform=tf)
        return DataLoader(train, batch_size=args.batch_size, shuffle=True, num_workers=1), None
    if dataset == "imagenet":
        tf = transforms.Compose([transforms.RandomResizedCrop(224), transforms.RandomHorizontalFlip(),
                                 transforms.Resize(224), transforms.ToTensor(),
                                 transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
        train = datasets.ImageFolder(os.path.join(args.data_dir, "ILSVRC2012_img_train"), tf)
        return DataLoader(train, batch_size=args.batch_size, shuffle=True, num_workers=8,
                          pin_memory=torch.cuda.is_available()), None
    raise RuntimeError("dataset %r: only --synthetic is available in this build (the DUTS pipeline needs "
                       "skimage transforms, SURVEY.md §2 #7, out of scope)" % (dataset,))
