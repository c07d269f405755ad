"""The algebra behind k_inner_packed (csrc/pxr_ba_inner.hip): a lane walks several 8-channel chunks of a descriptor, so the
L2 normalisation of PixelInterpolator (base/src/interpolation.h:648-666: f / |f|, then the chain rule, then the residual
against the reference d) is applied to CHANNEL SUMS of the raw interpolated values instead of to the values:

    N^2 = f.f     r.r = 1 - 2 (f.d) / N + d.d
    Jc.Jc = (fc.fc - (f.fc)^2 / N^2) / N^2     Jc.Jr = (fc.fr - (f.fc)(f.fr) / N^2) / N^2     Jc.r = -(fc.d - (f.d)(f.fc) / N^2) / N

This test states both forms in numpy and bounds their difference at the residual sizes of a converging problem."""
import numpy as np
import pytest


def _normalise_then_subtract(f, fc, fr, d):
    n = 1.0 / np.sqrt(f @ f)
    fh, c, r = f * n, fc * n, fr * n
    c = c - (fh @ c) * fh
    r = r - (fh @ r) * fh
    res = fh - d
    return res @ res, c @ c, c @ r, r @ r, c @ res, r @ res


def _from_sums(f, fc, fr, d):
    Sgg, Sgc, Sgr, Scc, Scr, Srr = f @ f, f @ fc, f @ fr, fc @ fc, fc @ fr, fr @ fr
    Sfd, Scd, Srd, r2 = f @ d, fc @ d, fr @ d, d @ d
    ninv = 1.0 / np.sqrt(Sgg)
    n2 = ninv * ninv
    pc, pr = Sgc * n2, Sgr * n2
    s = max(0.0, 1.0 - 2.0 * Sfd * ninv + r2)
    return s, (Scc - Sgc * pc) * n2, (Scr - Sgc * pr) * n2, (Srr - Sgr * pr) * n2, -(Scd - Sfd * pc) * ninv, -(Srd - Sfd * pr) * ninv


@pytest.mark.parametrize("channels", [64, 128])
@pytest.mark.parametrize("noise", [0.3, 0.03, 0.003])
def test_normalisation_on_channel_sums_equals_normalise_then_subtract(channels, noise):
    rng = np.random.default_rng(channels + int(1000 * noise))
    worst = 0.0
    for _ in range(200):
        f = rng.normal(size=channels) * rng.uniform(0.1, 10.0)
        fc, fr = rng.normal(size=channels), rng.normal(size=channels)
        d = f / np.linalg.norm(f) + noise * rng.normal(size=channels) / np.sqrt(channels)   # a reference |r| ~ noise away
        a, b = _normalise_then_subtract(f, fc, fr, d), _from_sums(f, fc, fr, d)
        scale = (a[0], a[1], np.sqrt(a[1] * a[3]), a[3], np.sqrt(a[1] * a[0]), np.sqrt(a[3] * a[0]))
        worst = max(worst, max(abs(x - y) / sc for x, y, sc in zip(a, b, scale)))
    # r.r is a difference of O(1) quantities: its relative error grows like eps / |r|^2
    assert worst < 4e-16 * 50 / noise ** 2
