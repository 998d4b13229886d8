"""``create_loaders(args)`` of the search (src/data/loaders.py:21-108) without torchvision: the two sample
pipelines as data (operation, arguments taken from ``args``), the list-file datasets, the meta-train / meta-val
split of search mode, torch DataLoaders."""
import logging

from torch.utils.data import DataLoader, random_split

from . import datasets as D

log = logging.getLogger(__name__)

# (operation, argument getters).  The POSITIONS matter: PascalCustomDataset.set_config rewrites operation 0
# (ResizeScale) and operation 2 (RandomCrop) of the training pipeline.
_TRAIN_OPS = (
    (D.ResizeScale, lambda a: (a.resize_side[0], a.low_scale, a.high_scale, a.resize_longer_side)),
    (D.RandomMirror, lambda a: ()),
    (D.RandomCrop, lambda a: (a.crop_size[0],)),
    (D.Normalise, lambda a: tuple(a.normalise_params)),
    (D.ToTensor, lambda a: ()),
)
_VAL_OPS = (
    (D.ResizeScale, lambda a: (a.val_resize_side, 1, 1, a.resize_longer_side)),
    (D.CentralCrop, lambda a: (a.val_crop_size,)),
    (D.Normalise, lambda a: tuple(a.normalise_params)),
    (D.ToTensor, lambda a: ()),
)


def _pipeline(table, args):
    return D.Compose([op(*getter(args)) for op, getter in table])


def _loader(dataset, batch_size, shuffle, args):
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=args.num_workers,
                      pin_memory=True, drop_last=True)


def create_loaders(args):
    """-> (train_loader, val_loader, do_search).  ``args``: train_dir, val_dir, train_list, val_list,
    meta_train_prct, resize_side[0], low_scale, high_scale, resize_longer_side, crop_size[0], val_resize_side,
    val_crop_size, normalise_params = (scale, mean, std), batch_size[0], val_batch_size, num_workers.
    Search mode (``do_search``) is train_list == val_list: ``meta_train_prct`` percent of that one list train the
    candidates and the rest scores them (one ``random_split`` draw from torch's global generator); both halves
    share the dataset object, whose stage the engine switches.  Otherwise the validation list is its own dataset.
    Both loaders drop the last incomplete batch; only the training loader shuffles."""
    val_ops = _pipeline(_VAL_OPS, args)
    full = D.PascalCustomDataset(args.train_list, args.train_dir, _pipeline(_TRAIN_OPS, args), val_ops)
    do_search = args.train_list == args.val_list
    if do_search:
        n_train = int(len(full) * args.meta_train_prct / 100.0)
        train_part, val_part = random_split(full, [n_train, len(full) - n_train])
    else:
        train_part = full
        val_part = D.PascalCustomDataset(args.val_list, args.val_dir, None, val_ops)
    log.info("data: %d training / %d validation samples (%s)", len(train_part), len(val_part),
             "search split" if do_search else "separate lists")
    return (_loader(train_part, args.batch_size[0], True, args),
            _loader(val_part, args.val_batch_size, False, args), do_search)
