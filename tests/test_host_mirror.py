"""Host-side containers mirror src/plane.rs and src/frame.rs."""
import numpy as np
import pytest


def test_videoplane_container(pkg):
    p = pkg.VideoPlane(5, 3)
    assert p.pixels.size == 15 and not p.pixels.any()
    with pytest.raises(AssertionError):
        pkg.VideoPlane.from_slice(4, 4, np.zeros(15, np.uint8))          # plane.rs:13 assert
    src = pkg.VideoPlane.from_slice(6, 4, np.arange(24, dtype=np.uint8))
    dst = pkg.VideoPlane(8, 8)
    dst.blit(src, 2, 3, 1, 1, 4, 2)                                       # plane.rs:20-29
    assert dst.image()[3, 2:6].tolist() == [7, 8, 9, 10] and dst.image()[4, 2:6].tolist() == [13, 14, 15, 16]
    assert dst.image().sum() == sum([7, 8, 9, 10, 13, 14, 15, 16])
    sl = src.get_slice(2, 1, 3, 2)                                        # plane.rs:31-36
    assert (sl.width, sl.height) == (3, 2) and sl.pixels.tolist() == [8, 9, 10, 14, 15, 16]
    assert src.reduce().pixels.tolist() == [0, 2, 4, 12, 14, 16]         # common.rs:523-536
    assert src.reduce().double().image().shape == (4, 6)                 # common.rs:538-556


def test_videoframe_layout(pkg):
    f = pkg.VideoFrame.new(18, 10)                                        # frame.rs:12-26
    assert (f.plane_u.width, f.plane_u.height) == (9, 5) and f.plane_u.pixels.min() == 128 and not f.plane_y.pixels.any()
    with pytest.raises(AssertionError):
        pkg.VideoFrame.new(17, 10)
    fp = pkg.VideoFrame.new_padded(1920, 1080)                            # frame.rs:28-49
    assert (fp.plane_y.width, fp.plane_y.height) == (1920, 1088)
    assert (fp.plane_u.width, fp.plane_u.height) == (960, 544)
    fp2 = pkg.VideoFrame.new_padded(100, 60)                              # chroma padded from (w/2, h/2) independently
    assert (fp2.plane_y.width, fp2.plane_y.height, fp2.plane_u.width, fp2.plane_u.height) == (112, 64, 64, 32)
    y = pkg.VideoPlane.from_slice(4, 4, np.arange(16, dtype=np.uint8))
    fr = pkg.VideoFrame.from_planes(4, 4, y, y, y)                        # frame.rs:51-59
    assert fr.plane_u.pixels.tolist() == [0, 2, 8, 10]
    packed = fr.packed()
    back = pkg.VideoFrame.from_packed(4, 4, packed)
    assert np.array_equal(back.plane_v.pixels, fr.plane_v.pixels)


def test_synthetic_stream_is_deterministic(pkg):
    a = pkg.SyntheticStream(64, 48).frame(3)
    b = pkg.SyntheticStream(64, 48).frame(3)
    assert np.array_equal(a, b) and a.size == 64 * 48 * 3 // 2
    assert int(a.astype(np.int64).sum()) == int(pkg.SyntheticStream(64, 48).frame(3).astype(np.int64).sum())
    assert not np.array_equal(a, pkg.SyntheticStream(64, 48, seed=1).frame(3))
