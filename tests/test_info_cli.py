"""CompressedStreamInfo / ToCSV (decompress.go:86-236) and the linzip CLI (main.go): host logic on the CPU, the
GPU-backed parts against the oracle's literal transcription."""
import io
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import numpy as np
import pytest

import oracle_lib as O


def run_cli(argv):
    from compress_amd.__main__ import main
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        rc = main(argv)
    return rc, out.getvalue(), err.getvalue()


def test_cli_messages_without_a_gpu(tmp_path):
    assert run_cli(["-version"]) == (0, "linzip v0.3.0\n", "")                      # main.go:44-47
    assert run_cli([]) == (1, "", "no input file specified\n")                      # main.go:49-51
    f = tmp_path / "x.bin"
    f.write_bytes(b"abc")
    assert run_cli(["-i", str(f), "-o", "y", "-no_out"]) == (1, "", "options -no_out and -o are mutually exclusive\n")
    rc, _, err = run_cli(["-i", str(tmp_path / "missing")])
    assert rc == 1 and "No such file" in err


def test_to_csv_format():
    from compress_amd import CompressionPhrase, ToCSV
    ph = [CompressionPhrase(0, 2, 7, 7, 0, b"hi"), CompressionPhrase(0xFE, 4, 3, 9, 16, b"hihi"),
          CompressionPhrase(0xFF, 5, 0, 13, 46, b"\x00" * 5)]
    want = O.to_csv([(p.Type, p.Length, p.ReferenceAddress, p.StartDecompressed, p.StartCompressed, p.Content) for p in ph])
    assert ToCSV(ph) == want
    assert want.splitlines()[2] == b"short,4,9,16,3,68696869"
    with pytest.raises(ValueError):
        ToCSV([CompressionPhrase(5, 1, 0, 0, 0, b"x")])


@pytest.mark.gpu
@pytest.mark.parametrize("dict_name", ["nil", "dict_naive"])
def test_stream_info_matches_the_transcription(dict_name, dict_naive, average_block):
    import compress_amd
    from compress_amd import synth
    dict_ = b"" if dict_name == "nil" else dict_naive
    rng = np.random.default_rng(1)
    cases = [b"", b"hi", b"hihihihihi", bytes(500), bytes([0xFE, 0xFF, 1, 0xFE]), average_block[:20000], synth.blob(2)[:30000],
             bytes(rng.integers(0, 256, 3000, dtype=np.uint8))]
    oc = O.Compressor(dict_)
    for d in cases:
        c = oc.compress(d)
        want = O.compressed_stream_info(c, dict_)
        got = compress_amd.CompressedStreamInfo(c, dict_)
        assert [(p.Type, p.Length, p.ReferenceAddress, p.StartDecompressed, p.StartCompressed, p.Content) for p in got] == want
        assert compress_amd.ToCSV(got) == O.to_csv(want)
    raw = b"\x00\x01\x01" + b"not compressed"                                   # decompress.go:109-118
    got = compress_amd.CompressedStreamInfo(raw, dict_)
    assert [(p.Type, p.Length, p.Content) for p in got] == [(0, len(raw) - 3, raw[3:])]
    with pytest.raises(compress_amd.LzssError, match="unsupported compressor version"):
        compress_amd.CompressedStreamInfo(b"\x00\x02\x00", dict_)


@pytest.mark.gpu
def test_cli_round_trip(tmp_path, dict_naive, average_block):
    src = tmp_path / "block.bin"
    src.write_bytes(average_block[:50000])
    dic = tmp_path / "dict"
    dic.write_bytes(dict_naive)
    rc, out, err = run_cli(["-i", str(src), "-dict", str(dic), "-r"])
    assert (rc, err) == (0, "")
    comp = (tmp_path / "block.bin.linzip").read_bytes()                             # main.go:74
    assert comp == O.Compressor(dict_naive).compress(average_block[:50000])
    pct = 50000 * 100 // len(comp)
    assert out == "50000 B -> %d B compression ratio %d.%02d\n" % (len(comp), pct // 100, pct % 100)
    os.remove(src)
    rc, out, err = run_cli(["-d", "-i", str(tmp_path / "block.bin.linzip"), "-dict", str(dic)])
    assert (rc, err) == (0, "")
    assert (tmp_path / "block.bin").read_bytes() == average_block[:50000]           # main.go:68-69
    rc, _, _ = run_cli(["-d", "-i", str(dic), "-no_out"])                            # not a linzip stream
    assert rc == 1
