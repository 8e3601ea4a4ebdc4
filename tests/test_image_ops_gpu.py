"""The image-op layers outside the DeepID stacks through the C ABI against the CPU oracle: bed of nails both
ways, separable Gaussian blur, bilinear resize, RGB -> YUV / L*a*b*, the shift layer's pixel gather.  Sub-sampling, the gather
and the undo move values unchanged: exact; the arithmetic ones at the per-kernel tolerance 1e-4*max + 1e-6."""
import numpy as np
import pytest

import ccn_loader

ccn_loader.load()
from kutil import call, close, dev, dev_empty  # noqa: E402
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def mk(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize('N', [128, 33])
@pytest.mark.parametrize('C_,imgY,imgX,stride', [(3, 32, 32, 2), (16, 13, 11, 3), (1, 7, 7, 7), (5, 24, 24, 1)])
def test_bed_of_nails_and_undo(C_, imgY, imgX, stride, N):
    rng = np.random.default_rng(60)
    outY, outX = (imgY + stride - 1) // stride, (imgX + stride - 1) // stride   # layer.py:383-385 per axis
    start = (min(imgY - stride * (outY - 1), imgX - stride * (outX - 1))) // 2
    imgs = mk(rng, C_ * imgY * imgX, N)
    for sT in (0.0, 1.0):
        t0 = mk(rng, C_ * outY * outX, N)
        want = O.bed_of_nails(imgs, N, C_, imgY, imgX, start, stride, outY, outX, target=t0.copy(), sT=sT, sO=0.5)
        d_t = dev(t0)
        call('mcn_bed_of_nails', dev(imgs), d_t, N, C_, imgY, imgX, start, stride, outY, outX, sT, 0.5, None)
        assert np.array_equal(d_t.numpy(), want)
        g = mk(rng, C_ * outY * outX, N)
        u0 = mk(rng, C_ * imgY * imgX, N)
        want = O.bed_of_nails_undo(g, N, C_, imgY, imgX, start, stride, outY, outX, target=u0.copy(), sT=sT, sO=1.0)
        d_u = dev(u0)
        call('mcn_bed_of_nails_undo', dev(g), d_u, N, C_, imgY, imgX, start, stride, outY, outX, sT, 1.0, None)
        assert np.array_equal(d_u.numpy(), want)
        if sT == 0.0:  # the undo is the transpose of the forward map: <nails(x), g> == <x, undo(g)>
            fwd = O.bed_of_nails(imgs, N, C_, imgY, imgX, start, stride, outY, outX)
            assert abs(float((fwd.astype(np.float64) * g).sum()) - float((imgs.astype(np.float64) * want).sum())) < 1e-6 * imgs.size


def gauss(fs, stdev):
    f = np.array([np.exp(-(fs // 2 - i) ** 2 / float(2 * stdev ** 2)) for i in range(fs)], np.float32)   # layer.py:417-419
    return (f / f.sum()).astype(np.float32)


@pytest.mark.parametrize('N', [128, 20])
@pytest.mark.parametrize('C_,imgY,imgX,fs', [(3, 32, 32, 5), (3, 24, 24, 9), (2, 9, 12, 3), (1, 7, 7, 7)])
def test_gaussian_blur_both_directions(C_, imgY, imgX, fs, N):
    rng = np.random.default_rng(61)
    imgs = mk(rng, C_ * imgY * imgX, N)
    f = gauss(fs, 1.3)
    for horiz in (1, 0):
        for sT in (0.0, 0.5):
            t0 = mk(rng, C_ * imgY * imgX, N)
            want = O.gaussian_blur(imgs, f, N, C_, imgY, imgX, horiz, target=t0.copy(), sT=sT, sO=2.0)
            d_t = dev(t0)
            call('mcn_gaussian_blur', dev(imgs), dev(f), d_t, N, C_, imgY, imgX, fs, horiz, sT, 2.0, None)
            close(d_t.numpy(), want, what='blur horiz=%d sT=%g' % (horiz, sT))
    # against the definition: zero-padded correlation with the (symmetric) filter
    img0 = imgs.reshape(C_, imgY, imgX, N)
    pad = np.pad(img0, ((0, 0), (0, 0), (fs // 2, fs // 2), (0, 0)))
    ref = sum(pad[:, :, k:k + imgX, :].astype(np.float64) * float(f[k]) for k in range(fs)).reshape(C_ * imgY * imgX, N)
    d_t = dev_empty(imgs.shape, np.nan)
    call('mcn_gaussian_blur', dev(imgs), dev(f), d_t, N, C_, imgY, imgX, fs, 1, 0.0, 1.0, None)
    close(d_t.numpy(), ref.astype(np.float32), what='blur vs definition')


@pytest.mark.parametrize('N', [128, 9])
@pytest.mark.parametrize('C_,img,scale', [(3, 32, 1.3), (3, 24, 0.75), (2, 17, 2.0), (1, 16, 0.5), (4, 10, 1.0)])
def test_resize_bilinear(C_, img, scale, N):
    rng = np.random.default_rng(62)
    tgt = int(np.floor(img / scale))   # layer.py:443
    imgs = mk(rng, C_ * img * img, N)
    want = O.resize_bilinear(imgs, N, C_, img, tgt, scale)
    d_t = dev_empty(want.shape, np.nan)
    call('mcn_resize_bilinear', dev(imgs), d_t, N, C_, img, tgt, scale, None)
    close(d_t.numpy(), want, what='resize')
    if scale == 1.0:  # identity away from the last row / column (the source position is clamped to imgSize - 1.01, conv_util.cu:312-313)
        got = d_t.numpy().reshape(C_, img, img, N)[:, :-1, :-1]
        close(got, imgs.reshape(C_, img, img, N)[:, :-1, :-1], rtol=1e-5, what='identity resize')


@pytest.mark.parametrize('N', [128, 7])
def test_colour_spaces(N):
    rng = np.random.default_rng(63)
    px = 23 * 23
    imgs = rng.random((3 * px, N)).astype(np.float32)   # the LAB transform wants inputs in 0..1 (conv_util.cu:219)
    imgs[:, 0] = 0.0                                    # the linear branch of labf
    imgs[:, 1] = 1.0
    want = O.rgb_to_yuv(imgs, N, px)
    d_t = dev_empty(want.shape, np.nan)
    call('mcn_rgb_to_yuv', dev(imgs), d_t, N, px, None)
    close(d_t.numpy(), want, what='yuv')
    for center in (0, 1):
        want = O.rgb_to_lab(imgs, N, px, center)
        d_t = dev_empty(want.shape, np.nan)
        call('mcn_rgb_to_lab', dev(imgs), d_t, N, px, center, None)
        close(d_t.numpy(), want, rtol=2e-4, what='lab center=%d' % center)
        L = d_t.numpy()[:px]
        assert L.min() >= (-50.5 if center else -0.5) and L.max() <= (50.5 if center else 100.5)
    assert abs(float(d_t.numpy()[px:, 1].max())) < 0.5   # white: a* = b* = 0 up to the 0.3333 exponent


@pytest.mark.parametrize('N', [128, 5])
@pytest.mark.parametrize('C_,img,fs', [(3, 12, 3), (2, 9, 4), (1, 6, 2)])
def test_shift_gather(C_, img, fs, N):
    rng = np.random.default_rng(64)
    out = img - fs + 1
    filt = np.zeros((out * out, (fs - 1) * (fs - 1)), np.float32)   # layer.py:1183-1192
    for dx in range(fs - 1):
        for dy in range(fs - 1):
            for x in range(out):
                for y in range(out):
                    filt[x * out + y, dx * (fs - 1) + dy] = (x + dx) * img + (y + dy)
    imgs = mk(rng, C_ * img * img, N)
    want = O.shift(imgs, filt, N, C_, img * img)
    d_t = dev_empty(want.shape, np.nan)
    call('mcn_shift', dev(imgs), dev(filt), d_t, N, C_, img * img, out * out, filt.shape[1], None)
    assert np.array_equal(d_t.numpy(), want)
