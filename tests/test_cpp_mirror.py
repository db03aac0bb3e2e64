"""include/pfv_hip.hpp -- the C++ mirror of pfv_rs::enc::Encoder / dec::Decoder -- driven by tests/cpp/roundtrip.cpp:
the .pfv bytes it writes and the frames it decodes must equal the oracle's.  The CPU run links the program against the
emulator build of the product sources; the GPU run (-m gpu) against libpfv_hip.so."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(lib_path, exe):
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "roundtrip.cpp"), "-o", exe, lib_path,
                    "-Wl,-rpath," + os.path.dirname(lib_path)], check=True)


def _run(pkg, oracle, exe, tmp_path, w, h, quality, n_frames, gop, drop_at):
    from oracle_bind import OracleStreamDecoder, OracleStreamEncoder
    st = pkg.SyntheticStream(w, h)
    frames = [st.frame(t) for t in range(n_frames)]
    yuv_in, pfv_out, yuv_out = (str(tmp_path / n) for n in ("in.yuv", "out.pfv", "out.yuv"))
    np.concatenate(frames).tofile(yuv_in)
    r = subprocess.run([exe, str(w), str(h), "30", str(quality), str(gop), str(drop_at), yuv_in, pfv_out, yuv_out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    oenc = OracleStreamEncoder(oracle, w, h, 30, quality)
    for t, f in enumerate(frames):
        if t == drop_at:
            oenc.encode_dropframe()
        elif t % gop == 0:
            oenc.encode_iframe(f)
        else:
            oenc.encode_pframe(f)
    oenc.finish()
    want = oenc.bytes()
    assert open(pfv_out, "rb").read() == want, "C++ Encoder bytes differ from the oracle's stream"
    odec = OracleStreamDecoder(oracle, want)
    decoded = []
    while True:
        rc, fr = odec.advance_frame()
        assert rc >= 0
        if fr is not None:
            decoded.append(fr)
        if rc == 0:
            break
    got = np.fromfile(yuv_out, dtype=np.uint8)
    assert got.size == sum(d.size for d in decoded) and np.array_equal(got, np.concatenate(decoded))
    assert f"decoded {len(decoded)}" in r.stdout
    assert f"gop: {len(decoded)} frames identical" in r.stdout     # pfv::GopEncoder / pfv::GopDecoder: same bytes, same frames
    if drop_at < 0:                      # the program also ran pfv::BatchEncoder / BatchDecoder on two copies of the clip
        assert "batch: 2 streams" in r.stdout


def test_cpp_mirror_on_emulator(pkg, oracle, tmp_path):
    import conftest
    exe = str(tmp_path / "roundtrip_emu")
    _build(conftest.build_emulator(), exe)
    _run(pkg, oracle, exe, tmp_path, 48, 32, 5, n_frames=5, gop=3, drop_at=2)
    _run(pkg, oracle, exe, tmp_path, 48, 32, 7, n_frames=4, gop=2, drop_at=-1)


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(graft, pkg, oracle, tmp_path):
    exe = str(tmp_path / "roundtrip")
    _build(graft.build_hip(), exe)
    _run(pkg, oracle, exe, tmp_path, 320, 240, 6, n_frames=7, gop=3, drop_at=4)
    _run(pkg, oracle, exe, tmp_path, 1920, 1080, 5, n_frames=3, gop=15, drop_at=-1)


def _build_driver(lib_path, exe):
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "threads_driver.cpp"), "-o", exe, lib_path, "-Wl,-rpath," + os.path.dirname(lib_path)], check=True)


def test_threads_driver_on_emulator(tmp_path):
    """tests/cpp/threads_driver.cpp (the program tools/sanitize.sh runs under ThreadSanitizer), plain: every threaded object -- pfv_decoder with
    0 / 1 / 3 look-ahead threads, pfv_gop_decoder's parse pool and device windows, pfv_gop_encoder, the batch objects -- on the intact stream and on
    damaged ones, host and device entropy: all configurations deliver the same frames and the same final error"""
    import conftest
    exe = str(tmp_path / "threads_driver_emu")
    _build_driver(conftest.build_emulator(), exe)
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "threads_driver OK: 4 streams" in r.stdout and "stream 0 (intact): 12 frames, final error 0" in r.stdout


@pytest.mark.gpu
def test_threads_driver_on_gpu(graft, tmp_path):
    exe = str(tmp_path / "threads_driver")
    _build_driver(graft.build_hip(), exe)
    r = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "threads_driver OK: 9 streams" in r.stdout


def test_e2e_native_tool_on_emulator(tmp_path):
    """tools/e2e_native.cpp (bench.py's extra.config4.end_to_end.native_host, tools/gpu_byref_quality.sh) at a toy size: the GOP encoder fed from
    host memory, from device memory by copy and BY REFERENCE writes the same bytes (the program checks it and exits non-zero otherwise), both of
    its output forms parse, and the by-reference pass took every frame in place"""
    import json
    import conftest
    lib = conftest.build_emulator()
    exe = str(tmp_path / "e2e_emu")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "e2e_native.cpp"), "-o", exe, lib,
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    args = [exe, "64", "48", "12", "3", "5", "2", "4", "2"]
    r = subprocess.run(args, env=dict(os.environ, PFV_E2E_STOP_AFTER_ENCODE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["stream_bytes"] > 600 and d["by_reference"] > 0 and d["by_reference_host_seconds"]["frames_by_reference"] == 12
    assert d["by_reference_host_seconds"]["batches_redone"] == 0
    line = tmp_path / "line.jsonl"
    line.write_text(r.stdout)
    t = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "byref_table.py"), str(line)], capture_output=True, text=True, timeout=60)
    assert t.returncode == 0 and len(t.stdout.splitlines()) == 2, t.stderr
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)                  # the full program: decode modes behind the encoders
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["encode_value_frames_in_hbm_by_reference"] > 0 and "identical" in d["frames_checked"]
