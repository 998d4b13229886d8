"""Data pipeline of the search (src/data): list-file dataset, augmentations, loaders - numpy / PIL / torch only."""
from .datasets import (CentralCrop, Compose, Normalise, Pad, PascalCustomDataset, RandomCrop,  # noqa: F401
                       RandomMirror, ResizeScale, ResizeShorter, ToTensor)
from .loaders import create_loaders  # noqa: F401
