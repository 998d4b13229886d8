"""``create_loaders(args)`` of the search (src/data/loaders.py:21-108): training / validation datasets from
list files, the meta-train / meta-val split when both lists are the same file, torch DataLoaders
(shuffle, drop_last, pinned memory) - without torchvision."""
import logging

from torch.utils.data import DataLoader, random_split

from .datasets import CentralCrop, Compose, Normalise
from .datasets import PascalCustomDataset as Dataset
from .datasets import RandomCrop, RandomMirror, ResizeScale, ToTensor


def create_loaders(args):
    """args: train_dir, val_dir, train_list, val_list, meta_train_prct, resize_side[0], low_scale, high_scale,
    resize_longer_side, crop_size[0], val_resize_side, val_crop_size, normalise_params (scale, mean, std),
    batch_size[0], val_batch_size, num_workers.  Returns (train_loader, val_loader, do_search); do_search is
    True when train_list == val_list (the training list is then split into meta-train / meta-val)."""
    logger = logging.getLogger(__name__)
    composed_trn = Compose([
        ResizeScale(args.resize_side[0], args.low_scale, args.high_scale, args.resize_longer_side),
        RandomMirror(),
        RandomCrop(args.crop_size[0]),
        Normalise(*args.normalise_params),
        ToTensor(),
    ])
    composed_val = Compose([
        ResizeScale(args.val_resize_side, 1, 1, args.resize_longer_side),
        CentralCrop(args.val_crop_size),
        Normalise(*args.normalise_params),
        ToTensor(),
    ])
    trainset = Dataset(data_file=args.train_list, data_dir=args.train_dir, transform_trn=composed_trn,
                       transform_val=composed_val)
    do_search = False
    if args.train_list == args.val_list:
        do_search = True
        n_examples = len(trainset)
        n_train = int(n_examples * args.meta_train_prct / 100.0)
        trainset, valset = random_split(trainset, [n_train, n_examples - n_train])
    else:
        valset = Dataset(data_file=args.val_list, data_dir=args.val_dir, transform_trn=None,
                         transform_val=composed_val)
    logger.info(" Created train set = {} examples, val set = {} examples; do_search = {}".format(
        len(trainset), len(valset), do_search))
    train_loader = DataLoader(trainset, batch_size=args.batch_size[0], shuffle=True, num_workers=args.num_workers,
                              pin_memory=True, drop_last=True)
    val_loader = DataLoader(valset, batch_size=args.val_batch_size, shuffle=False, num_workers=args.num_workers,
                            pin_memory=True, drop_last=True)
    return train_loader, val_loader, do_search
