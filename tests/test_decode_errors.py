"""The reference decoder's messages for malformed streams (lzss/decompress.go:53-55,66-69), which the host derives from a
walk over the phrase boundaries when the device reports MI355_LZSS_E_FORMAT.  The expected texts are Go's `%+v` of the
reference's `backref` struct (lzss/backref.go:21-59) written out by hand: the reference holds no test that pins them
(compress_test.go:365-405 only asserts that an error comes back), so this is parity unpinned."""
import pytest


def streams():
    field = (1 - 255 - 1) & 0x3FFF                                   # compress_test.go:394-404
    bits = (0xFE << 22) | (4 << 14) | field
    short = bytes.fromhex("000100") + (bits << 2).to_bytes(4, "big")
    bits = (0xFF << 29) | (4 << 21) | (2 - 1)
    dyn = bytes.fromhex("000100") + (bits << 3).to_bytes(5, "big")
    return short, dyn


SHORT_MSG = ("invalid short backref {address:16130 length:5 bType:{Delimiter:254 NbBitsAddress:14 NbBitsLength:8 "
             "NbBitsBackRef:30 nbBytesBackRef:4 maxAddress:16384 maxLength:256 DictLen:0}} - output buffer is only 0 bytes long")
DYN_MSG = ("invalid dynamic backref {address:2 length:5 bType:{Delimiter:255 NbBitsAddress:21 NbBitsLength:8 "
           "NbBitsBackRef:37 nbBytesBackRef:5 maxAddress:2097152 maxLength:256 DictLen:8}} - dict is only 8 bytes long; "
           "dictStart = 6")


def test_messages_from_the_phrase_walk():
    from compress_amd import lzss
    short, dyn = streams()
    assert lzss.decode_error_text(short, 0) == SHORT_MSG
    assert lzss.decode_error_text(dyn, len(lzss.AugmentDict(b"abcdef"))) == DYN_MSG
    assert lzss.decode_error_text(bytes.fromhex("00010001fe04"), 0) == "EOF"         # truncated: the bit reader's io.EOF
    assert lzss.decode_error_text(bytes.fromhex("00010001fe040000"), 0) is None      # a valid stream
    # literals count towards the output length: one literal in front, address 2 > 1
    bits = (0xFE << 22) | (0 << 14) | 1
    s = bytes.fromhex("00010041") + (bits << 2).to_bytes(4, "big")
    assert "address:2 length:1" in lzss.decode_error_text(s, 0) and "only 1 bytes long" in lzss.decode_error_text(s, 0)


@pytest.mark.gpu
def test_decompress_raises_the_reference_texts():
    import compress_amd as amd
    short, dyn = streams()
    with pytest.raises(amd.LzssError) as e:
        amd.Decompress(short, b"")
    assert str(e.value) == SHORT_MSG
    with pytest.raises(amd.LzssError) as e:
        amd.Decompress(dyn, b"abcdef")
    assert str(e.value) == DYN_MSG


@pytest.mark.gpu
def test_truncated_reference_raises_the_readers_eof():
    """a stream that ends inside a back-reference: readFrom returns the bit reader's TryError (backref.go:77-79), io.EOF"""
    import compress_amd as amd
    for s in ("00010001fe04", "00010001ff0400", "000100fe"):
        with pytest.raises(amd.LzssError) as e:
            amd.Decompress(bytes.fromhex(s), b"")
        assert str(e.value) == "EOF"
