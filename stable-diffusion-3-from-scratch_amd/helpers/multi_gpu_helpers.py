"""Rank helpers: mirror of src/helpers/multi_gpu_helpers.py."""
import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def is_main_process(group=None):
    """The reference defines this twice (zero-arg at line 25, group-arg at 30 which shadows it)."""
    if not is_dist_avail_and_initialized():
        return True
    return dist.get_rank(group) == 0
