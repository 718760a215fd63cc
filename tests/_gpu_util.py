"""Comparison helpers shared by the GPU numerics tests.

A kernel result `a` is compared with an fp32 PyTorch reference `b` of the same op fed the SAME rounded inputs:

  * per element:  |a - b| <= atol * rms(b) + rtol * |b|     (rtol ~ 2 ulp of the output dtype; the absolute term
    covers elements that are small through cancellation, where the error scales with the magnitude of the terms
    that were summed, not with the result)
  * whole tensor: ||a - b||_F / ||b||_F <= frob              (catches a systematic error that hides inside the
    per-element band, e.g. a tile of slightly wrong values)

`scale` loosens all three for ops with intermediate roundings (a mask taken from a rounded output, two-pass
statistics, ...) and says so at the call site.
"""
import torch

#            rtol      atol     frob
TOL = {
    torch.float32: (1e-4, 1e-4, 2e-5),
    torch.bfloat16: (1.6e-2, 6e-3, 4e-3),
    torch.float16: (2e-3, 1e-3, 6e-4),
}


def assert_close(a, b, dtype, what="", scale=1.0, frob_scale=None):
    rtol, atol, frob = TOL[dtype]
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a).all(), "%s: non-finite values" % what
    rms = b.pow(2).mean().sqrt().item()
    err = (a - b).abs()
    bound = scale * (atol * max(rms, 1e-30) + rtol * b.abs())
    bad = err > bound
    if bad.any():
        i = int((err - bound).argmax())
        raise AssertionError("%s: %d / %d elements out of tolerance; worst |%g - %g| = %g > %g (rms %g)" % (
            what, int(bad.sum()), a.numel(), a.flatten()[i], b.flatten()[i], err.flatten()[i], bound.flatten()[i], rms))
    nb = b.norm().item()
    rel = (a - b).norm().item() / max(nb, 1e-30)
    lim = frob * (frob_scale if frob_scale is not None else scale)
    assert rel <= lim, "%s: relative Frobenius error %g > %g" % (what, rel, lim)
