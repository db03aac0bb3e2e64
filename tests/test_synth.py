"""The device-side synthetic generator (k_synth_frames) against synth.py, byte for byte:
on the CPU emulator here, on the MI355X under -m gpu."""
import numpy as np
import pytest


def _check(pkg, ctx, w, h, seeds, ts, kind="pan"):
    fb = int(pkg._lib.load().pfv_frame_bytes(w, h))
    dev = ctx.alloc(fb * len(seeds))
    try:
        for t in ts:
            ctx.synth_frames_dev(w, h, seeds, t, dev, kind=kind)
            got = np.empty((len(seeds), fb), np.uint8)
            ctx.download(got, dev)
            for k, s in enumerate(seeds):
                want = pkg.SyntheticStream(w, h, seed=int(s), kind=kind).frame(t)
                assert np.array_equal(got[k], want), (w, h, s, t, kind)
    finally:
        ctx.free(dev)


def test_emu_synth_matches_numpy(pkg, emu_ctx):
    _check(pkg, emu_ctx, 64, 48, [pkg.synth.SEED, pkg.synth.SEED + 17], [0, 1, 9])
    _check(pkg, emu_ctx, 50, 38, [12345], [4, 23])      # odd chroma width, negative floor-halved chroma motion
    _check(pkg, emu_ctx, 64, 48, [pkg.synth.SEED, 99], [0, 1, 9, 40], kind="low_motion")
    _check(pkg, emu_ctx, 50, 38, [12345], [4, 23], kind="low_motion")
    _check(pkg, emu_ctx, 2, 2, [7], [0, 3], kind="low_motion")
    _check(pkg, emu_ctx, 64, 48, [5], [0, 2], kind="static")


@pytest.mark.gpu
def test_gpu_synth_matches_numpy(pkg, gpu_ctx):
    _check(pkg, gpu_ctx, 64, 48, [pkg.synth.SEED, 1, 2 ** 40 + 5], [0, 1, 9, 299])
    _check(pkg, gpu_ctx, 50, 38, [12345], [4, 23])
    _check(pkg, gpu_ctx, 1920, 1080, [pkg.synth.SEED + 17 * 3], [7])
    _check(pkg, gpu_ctx, 64, 48, [pkg.synth.SEED, 1, 2 ** 40 + 5], [0, 1, 9, 299], kind="low_motion")
    _check(pkg, gpu_ctx, 50, 38, [12345], [4, 23], kind="low_motion")
    _check(pkg, gpu_ctx, 1920, 1080, [pkg.synth.SEED + 17 * 3], [7, 200], kind="low_motion")
