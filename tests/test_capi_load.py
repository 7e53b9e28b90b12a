"""CPU-only: the C-ABI library loads and exports every symbol include/mi355_lzss.h declares; without a
GPU the product path fails loudly instead of falling back to any CPU code."""
import ctypes as C
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_lzss.h")).read()
    return sorted(set(re.findall(r"\b(mi355_lzss_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from compress_amd import _lib
    l = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), s
    assert sorted(_lib.SIGNATURES) == syms


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "compress_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("oracle_lib", "lzss_oracle", "liblzss_oracle", "oracle/", "import oracle", "ora_"):
                    assert needle not in text, (f, needle)
    out = subprocess.run(["ldd", os.path.join(ROOT, "compress_amd", "libmi355_lzss.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import compress_amd
    with pytest.raises(compress_amd.LzssError, match="no HIP device|hip"):
        compress_amd.Compressor(b"")


def test_host_logic_without_device():
    import compress_amd
    assert compress_amd.AugmentDict(b"") == b"\xfe\xff"
    assert compress_amd.AugmentDict(b"\xffa\xfe") == b"\xffa\xfe"
    h = compress_amd.Header.from_bytes(b"\x00\x01\x01")
    assert (h.Version, h.NoCompression) == (1, True)
    assert compress_amd.Header(1, False).to_bytes() == b"\x00\x01\x00"
    with pytest.raises(compress_amd.LzssError, match="expected 0 or 1"):
        compress_amd.Header.from_bytes(b"\x00\x01\x02")
    with pytest.raises(compress_amd.LzssError, match="dict size must be <= 4194304"):
        compress_amd.Compressor(bytes((1 << 22) - 1))


def test_synth_statistics():
    """SURVEY.md Appendix C acceptance: calldata-like byte statistics, every phrase kind in blob 0"""
    import numpy as np
    import oracle_lib as O
    from compress_amd import synth
    dict_ = synth.dictionary()
    assert len(dict_) == 65536
    c = O.Compressor(dict_)
    for b in range(3):
        d = synth.blob(b)
        assert len(d) == 131072
        a = np.frombuffer(d, np.uint8)
        assert 0.55 <= (a == 0).mean() <= 0.72
        assert 0.003 <= (a == 255).mean() <= 0.03
        out = c.compress(d)
        st = c.phrase_stats()
        assert 3 <= len(d) / len(out) <= 9
        assert st["rle"] >= 1
        if b == 0:
            assert min(st["literals"], st["short"], st["long_data"], st["long_dict"], st["rle"]) > 0
    assert synth.blob(0) == synth.blob(0) and synth.blob(0) != synth.blob(1)


def test_partition_is_host_only_and_balanced():
    """the blob -> device partition of the multi-device entry points (mi355_lzss_partition) needs no GPU: contiguous
    ranges, every blob in exactly one, balanced by bytes"""
    import numpy as np
    import compress_amd
    rng = np.random.default_rng(1)
    for n_blobs in (0, 1, 5, 1000):
        lens = rng.integers(0, 200000, n_blobs)
        off = np.concatenate([[7], 7 + np.cumsum(lens)]).astype(np.uint64)   # offsets need not start at 0
        for parts in (1, 2, 3, 8):
            first = compress_amd.partition(off, parts)
            assert first[0] == 0 and first[-1] == n_blobs and len(first) == parts + 1
            assert all(a <= b for a, b in zip(first, first[1:]))
            if n_blobs == 1000:
                sizes = [int(off[first[t + 1]] - off[first[t]]) for t in range(parts)]
                assert max(sizes) - min(sizes) <= 2 * 200000
    # equal blobs: equal shares (config 3: 10 000 blobs over 8 GPUs = 1250 each)
    off = (np.arange(10001, dtype=np.uint64) * 131072)
    assert compress_amd.partition(off, 8) == [1250 * t for t in range(9)]


# ---- the Go shim cannot be compiled in this image (no Go toolchain); two mechanical checks stand in for the compiler

def test_header_compiles_as_c99():
    """cgo hands include/mi355_lzss.h to a C compiler: it must be plain C (no C++-isms), warning-free"""
    src = '#include "mi355_lzss.h"\nint main(void) { return (int)sizeof(mi355_lzss_ctx *) > 0 ? 0 : 1; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                        "-I", os.path.join(ROOT, "include"), "-"], input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def header_prototypes():
    """name -> number of parameters, from the declarations of include/mi355_lzss.h"""
    text = open(os.path.join(ROOT, "include", "mi355_lzss.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"\b(mi355_lzss_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(split_args(args))
    return protos


def split_args(s):
    """top-level comma split of an argument list"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def go_calls():
    """(name, argument count, line) of every C.mi355_lzss_*( call in go/lzss/lzss.go"""
    text = open(os.path.join(ROOT, "go", "lzss", "lzss.go")).read()
    calls = []
    for m in re.finditer(r"\bC\.(mi355_lzss_[a-z0-9_]+)\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        args = text[m.end():i - 1].strip()
        calls.append((m.group(1), len(split_args(args)) if args else 0, text.count("\n", 0, m.start()) + 1))
    return calls


def test_go_shim_calls_match_the_header():
    protos = header_prototypes()
    assert sorted(protos) == declared_symbols()
    calls = go_calls()
    assert len(calls) >= 10
    for name, nargs, line in calls:
        assert name in protos, f"go/lzss/lzss.go:{line}: {name} is not declared in include/mi355_lzss.h"
        assert nargs == protos[name], f"go/lzss/lzss.go:{line}: {name} called with {nargs} arguments, declared with {protos[name]}"
    # constants the shim takes from the header
    text = open(os.path.join(ROOT, "go", "lzss", "lzss.go")).read()
    hdr = open(os.path.join(ROOT, "include", "mi355_lzss.h")).read()
    for c in set(re.findall(r"\bC\.(MI355_LZSS_[A-Z0-9_]+)\b", text)):
        assert re.search(r"\b%s\b" % c, hdr), c
