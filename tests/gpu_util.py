"""Helpers for the -m gpu parity tests: call the C ABI with torch device tensors."""
import contextlib

import torch

from calipsync_amd import _lib


def dev():
    return torch.device("cuda:0")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ok(status, what="op"):
    _lib.check(status, what)
    torch.cuda.synchronize()


def nhwc(t):     # NCHW torch tensor -> contiguous NHWC on the GPU
    return t.permute(0, 2, 3, 1).contiguous().to(dev())


def nchw(t):     # NHWC device tensor -> NCHW on the CPU
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@contextlib.contextmanager
def options(target=None, **kv):
    """Set engine switches (casync_set_option) for the duration of a block and restore them: on a
    calipsync_amd.unet.Model, or on the process defaults used by the casync_op_* calls (target None)."""
    if target is None:
        old = {k: _lib.get_option(k) for k in kv}
        for k, v in kv.items():
            _lib.set_option(k, v)
        try:
            yield
        finally:
            for k, v in old.items():
                _lib.set_option(k, v)
    else:
        old = {k: target.get_option(k) for k in kv}
        for k, v in kv.items():
            target.set_option(k, v)
        try:
            yield
        finally:
            for k, v in old.items():
                target.set_option(k, v)
