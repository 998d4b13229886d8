"""The error convention of the engine entry points (src/helpers/utils.py:172-187)."""
import logging

logger = logging.getLogger(__name__)


def try_except(func):
    """Run ``func``; any RuntimeError (HIP OOM, a shape mismatch in a sampled
    architecture, a failed kernel launch) scores the candidate 0 instead of
    stopping the search."""

    def wrapper_func(*args, **kwargs):
        try:
            return func(*args, **kwargs)
        except RuntimeError as e:
            # (the reference swallows it silently; a candidate that scores 0 because of a bug on
            #  this side of the boundary should at least leave a trace)
            logger.warning(" %s failed, candidate scored 0: %s", getattr(func, "__name__", "call"), e)
            return 0

    wrapper_func.__wrapped__ = func
    wrapper_func.__name__ = getattr(func, "__name__", "wrapped")
    wrapper_func.__doc__ = func.__doc__
    return wrapper_func


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def compute_params(model):
    """(total, total without parameters whose name contains 'aux') - utils.py:16-25."""
    total = aux = 0
    for name, p in model.named_parameters():
        total += p.numel()
        if "aux" in name:
            aux += p.numel()
    return total, total - aux
