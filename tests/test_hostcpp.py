"""The C++ host mirror (compress_amd/hostcpp/lzss.hpp) compiles and links against the C-ABI library on
the CPU; on the GPU box (-m gpu) a small driver runs the TestRevert-style flow through it and its
streams are checked against the oracle."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRIVER = os.path.join(HERE, "hostcpp_driver.cc")
EXE = os.path.join(HERE, "hostcpp_driver.bin")


def build_driver():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, DRIVER,
                           "-L" + os.path.join(ROOT, "compress_amd"), "-lmi355_lzss",
                           "-Wl,-rpath," + os.path.join(ROOT, "compress_amd")])


def test_hostcpp_compiles_and_links():
    build_driver()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_hostcpp_flow_matches_oracle(tmp_path, dict_naive, average_block):
    import oracle_lib as O
    if not os.path.exists(EXE):
        build_driver()
    data = average_block[:30000]
    (tmp_path / "dict").write_bytes(dict_naive)
    (tmp_path / "data").write_bytes(data)
    out = subprocess.run([EXE, str(tmp_path / "dict"), str(tmp_path / "data"), str(tmp_path / "out")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # the driver writes: one-shot stream, then the blobs of a 1000-byte Write / 4000-byte cap Revert flow
    raw = (tmp_path / "out").read_bytes()
    streams = []
    pos = 0
    while pos < len(raw):
        n = int.from_bytes(raw[pos:pos + 4], "little")
        streams.append(raw[pos + 4:pos + 4 + n])
        pos += 4 + n
    oc = O.Compressor(dict_naive)
    assert streams[0] == oc.compress(data)
    back = b"".join(O.decompress(s, dict_naive) for s in streams[1:-1])
    assert back == data
    assert streams[-1] == data                      # lzss::Decompress of the one-shot stream
    # same flow on the oracle gives the same blobs
    oc.reset()
    i0, k = 0, 1
    while i0 < len(data):
        i = i0
        while i < len(data) and oc.len() < 4000:
            oc.write(data[i:i + 1000])
            i += 1000
        if oc.len() > 4000:
            oc.revert()
            i -= 1000
        assert streams[k] == oc.bytes()
        k += 1
        oc.reset()
        i0 = i
