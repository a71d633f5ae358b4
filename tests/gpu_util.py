"""Helpers for the -m gpu parity tests: call the C ABI with torch device tensors."""
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
