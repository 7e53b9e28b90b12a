// lzss.hpp - C++ host-side mirror of the reference's Go package `lzss` over the C ABI (include/mi355_lzss.h).
//
// Same names, argument meaning and error behaviour as lzss/compress.go (Compressor, NewCompressor, Write,
// Reset, Len, Written, WrittenBytes, Revert, ConsiderBypassing, Bytes, Compress, CompressedSize256k),
// lzss/decompress.go (Decompress) and lzss/header.go (Header).  Go `error` returns become lzss::Error
// exceptions carrying the reference's message text.  The index build, match search, phrase walk and bit
// emission of every Write run on the GPU (mi355_lzss_write); only the stream state of compress.go:91-129,
// 289-367 (header, re-opened last byte, Revert bookkeeping) lives here.  Header-only; link libmi355_lzss.so.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lzss_format.h"
#include "../../include/mi355_lzss.h"

namespace lzss {

constexpr int MaxInputSize = LZSS_MAX_INPUT_SIZE;  // lzss/backref.go:10
constexpr int MaxDictSize = LZSS_MAX_DICT_SIZE;    // lzss/backref.go:11
constexpr uint8_t SymbolDynamic = 0xFF;            // lzss/backref.go:15
constexpr uint8_t SymbolShort = 0xFE;              // lzss/backref.go:16
constexpr int Version = LZSS_VERSION;              // lzss/header.go:11
constexpr int HeaderSize = LZSS_HEADER_SIZE;       // lzss/header.go:12

struct Error : std::runtime_error {
    int code;
    explicit Error(const std::string &m, int c = MI355_LZSS_E_ARG) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc != MI355_LZSS_OK) throw Error(mi355_lzss_last_error(), rc);
}

using Bytes = std::vector<uint8_t>;

// lzss/compress.go:70-88
inline Bytes AugmentDict(const Bytes &dict) {
    uint8_t found = 0;
    for (uint8_t b : dict) {
        if (b == SymbolShort) found |= 2;
        else if (b == SymbolDynamic) found |= 4;
        else continue;
        if (found == 6) return dict;
    }
    Bytes out(dict);
    out.push_back(SymbolShort);
    out.push_back(SymbolDynamic);
    return out;
}

// lzss/header.go:17-44
struct Header {
    uint16_t Version = lzss::Version;
    bool NoCompression = false;
    void WriteTo(Bytes &w) const {
        w.push_back(uint8_t(Version >> 8));
        w.push_back(uint8_t(Version & 0xFF));
        w.push_back(NoCompression ? 1 : 0);
    }
    static Header ReadFrom(const uint8_t *p, size_t n) {
        if (n < size_t(HeaderSize)) throw Error("unexpected EOF");
        if (p[2] > 1) throw Error("expected 0 or 1");
        Header h;
        h.Version = uint16_t((p[0] << 8) | p[1]);
        h.NoCompression = p[2] == 1;
        return h;
    }
};

class Compressor {
  public:
    // NewCompressor, lzss/compress.go:37-67
    explicit Compressor(const Bytes &dict = {}, int device = 0) {
        check(mi355_lzss_ctx_create(dict.empty() ? nullptr : dict.data(), dict.size(), device, &ctx_));
        Reset();
    }
    ~Compressor() { mi355_lzss_ctx_destroy(ctx_); }
    Compressor(const Compressor &) = delete;
    Compressor &operator=(const Compressor &) = delete;

    // lzss/compress.go:289-304
    void Reset() {
        noCompression_ = false;
        out_.clear();
        Header{uint16_t(lzss::Version), false}.WriteTo(out_);
        in_.clear();
        lastOutLen_ = long(out_.size());
        lastNbSkippedBits_ = 0;
        nbSkippedBits_ = 0;
        lastInLen_ = 0;
    }

    // lzss/compress.go:91-129
    int Write(const uint8_t *d, size_t n) {
        lastOutLen_ = long(out_.size());
        uint8_t last = uint8_t(out_.back() >> nbSkippedBits_);
        out_.pop_back();
        unsigned pendingBits = 8u - nbSkippedBits_;
        if (pendingBits == 8) {  // a whole byte goes straight back (bitio cache stays empty)
            out_.push_back(last);
            pendingBits = 0;
        }
        lastNbSkippedBits_ = nbSkippedBits_;
        if (in_.size() + n > size_t(MaxInputSize))  // appendInput, compress.go:472-479
            throw Error("input size must be <= " + std::to_string(MaxInputSize), MI355_LZSS_E_INPUT_SIZE);
        lastInLen_ = long(in_.size());
        in_.insert(in_.end(), d, d + n);
        if (noCompression_) {
            out_.insert(out_.end(), d, d + n);
            return int(n);
        }
        const size_t start = size_t(lastInLen_);
        Bytes buf((pendingBits + 37 * (in_.size() - start) + 7) / 8 + 16);
        uint64_t nbits = 0;
        check(mi355_lzss_write(ctx_, in_.data(), in_.size(), start, pendingBits, buf.data(), buf.size(), &nbits));
        const uint64_t total = pendingBits + nbits;
        const size_t nbytes = size_t((total + 7) / 8);
        if (nbytes) {
            if (pendingBits) buf[0] |= uint8_t(last << (8 - pendingBits));
            out_.insert(out_.end(), buf.begin(), buf.begin() + long(nbytes));
        }
        nbSkippedBits_ = uint8_t((8 - total % 8) % 8);  // bw.Align(), compress.go:127
        return int(in_.size() - start);
    }
    int Write(const Bytes &d) { return Write(d.data(), d.size()); }

    size_t Len() const { return out_.size(); }           // compress.go:307
    size_t Written() const { return in_.size(); }        // compress.go:312
    const Bytes &WrittenBytes() const { return in_; }    // compress.go:318
    const Bytes &GetBytes() const { return out_; }       // Bytes(), compress.go:370

    // lzss/compress.go:324-345
    void Revert() {
        if (lastInLen_ == -1) throw Error("cannot revert twice in a row");
        in_.resize(size_t(lastInLen_));
        lastInLen_ = -1;
        if (noCompression_) {
            Bytes all(in_);
            Reset();
            Write(all);
            ConsiderBypassing();
            return;
        }
        out_.resize(size_t(lastOutLen_));
        nbSkippedBits_ = lastNbSkippedBits_;
    }

    // lzss/compress.go:348-367
    bool ConsiderBypassing() {
        if (out_.size() > in_.size() + size_t(HeaderSize)) {
            noCompression_ = true;
            nbSkippedBits_ = 0;
            lastOutLen_ = lastInLen_ + HeaderSize;
            lastNbSkippedBits_ = 0;
            out_.clear();
            Header{uint16_t(lzss::Version), true}.WriteTo(out_);
            out_.insert(out_.end(), in_.begin(), in_.end());
            return true;
        }
        return false;
    }

    // lzss/compress.go:375-379
    const Bytes &Compress(const Bytes &d) {
        Reset();
        Write(d);
        return out_;
    }

    // lzss/compress.go:384-407
    size_t CompressedSize256k(const Bytes &d) {
        if (noCompression_) return size_t(HeaderSize) + d.size();
        uint64_t size = 0;
        check(mi355_lzss_compressed_size(ctx_, d.data(), d.size(), &size));
        return size_t(size);
    }

    // Filling size-capped blobs from a stream: the Write / Revert loop of lzss/compress_test.go:337-353 with the
    // candidate Writes of a blob run as device batches (mi355_lzss_pack_blob); blobs in which ConsiderBypassing would
    // fire run the loop itself.  Returns (blob bytes, input bytes it covers) per blob.  No counterpart in the reference.
    std::vector<std::pair<Bytes, size_t>> PackStream(const Bytes &data, size_t chunk, size_t outMax, unsigned lookahead = 64) {
        std::vector<std::pair<Bytes, size_t>> res;
        Bytes buf(outMax + 16);
        for (size_t i0 = 0; i0 < data.size();) {
            uint64_t len = 0, chunks = 0;
            int loop = 0;
            check(mi355_lzss_pack_blob(ctx_, data.data() + i0, data.size() - i0, (uint32_t)chunk, outMax, lookahead, buf.data(),
                                       buf.size(), &len, &chunks, &loop));
            Bytes blob;
            size_t k = size_t(chunks);
            if (loop) {
                Reset();
                size_t i = i0;
                while (i < data.size() && Len() < outMax) {
                    Write(data.data() + i, std::min(chunk, data.size() - i));
                    const size_t uncompressed = i + chunk - i0 + 3;
                    if (Len() >= outMax && uncompressed <= outMax && Len() > uncompressed) ConsiderBypassing();
                    i += chunk;
                }
                if (Len() > outMax) {
                    Revert();
                    i -= chunk;
                }
                blob = out_;
                k = (i - i0) / chunk;
            } else {
                blob.assign(buf.begin(), buf.begin() + (long)len);
            }
            if (k == 0) throw Error("a single chunk does not fit the output cap");
            res.emplace_back(blob, std::min(i0 + k * chunk, data.size()) - i0);
            i0 += k * chunk;
        }
        Reset();
        return res;
    }

    mi355_lzss_ctx *ctx() { return ctx_; }

  private:
    mi355_lzss_ctx *ctx_ = nullptr;
    Bytes out_, in_;
    uint8_t nbSkippedBits_ = 0, lastNbSkippedBits_ = 0;
    long lastOutLen_ = 0, lastInLen_ = 0;
    bool noCompression_ = false;
};

// The reference decoder's message for a stream the device rejected as malformed (lzss/decompress.go:53-55,66-69), from a
// walk over the phrase boundaries (no bytes are decoded); "EOF" when the stream ends inside a phrase.
// dictLen: length of the augmented dictionary.  The texts are Go's %+v of the reference's backref struct.
inline std::string DecodeErrorText(const Bytes &data, size_t dictLen) {
    const size_t total = data.size() * 8;
    size_t b = size_t(HeaderSize) * 8, outLen = 0;
    auto bits = [&](size_t pos, int nb) {
        uint64_t v = 0;
        for (int k = 0; k < nb; ++k, ++pos) v = (v << 1) | ((data[pos >> 3] >> (7 - (pos & 7))) & 1u);
        return v;
    };
    auto repr = [](long address, long length, int delim, int nbAddr, size_t dl) {
        const int nb = 8 + nbAddr + 8;
        return "{address:" + std::to_string(address) + " length:" + std::to_string(length) + " bType:{Delimiter:" +
               std::to_string(delim) + " NbBitsAddress:" + std::to_string(nbAddr) + " NbBitsLength:8 NbBitsBackRef:" +
               std::to_string(nb) + " nbBytesBackRef:" + std::to_string((nb + 7) / 8) + " maxAddress:" +
               std::to_string(1L << nbAddr) + " maxLength:256 DictLen:" + std::to_string(dl) + "}}";
    };
    while (b + 8 <= total) {
        const unsigned s = (unsigned)bits(b, 8);
        if (s != 0xFE && s != 0xFF) {
            ++outLen;
            b += 8;
            continue;
        }
        const int nbAddr = s == 0xFE ? 14 : 21;
        if (b + 16 + nbAddr > total) return "EOF";  // readFrom returns the bit reader's io.EOF (backref.go:77-79)
        const long length = (long)bits(b + 8, 8) + 1, address = (long)bits(b + 16, nbAddr) + 1;
        if (s == 0xFE) {
            if (address > (long)outLen)
                return "invalid short backref " + repr(address, length, 0xFE, 14, 0) + " - output buffer is only " +
                       std::to_string(outLen) + " bytes long";
        } else if (address > (long)outLen) {
            const long dictStart = (long)dictLen - (address - (long)outLen);
            if (dictStart < 0 || dictStart > (long)dictLen || dictStart + length > (long)dictLen)
                return "invalid dynamic backref " + repr(address, length, 0xFF, 21, dictLen) + " - dict is only " +
                       std::to_string(dictLen) + " bytes long; dictStart = " + std::to_string(dictStart);
        }
        outLen += (size_t)length;
        b += 16 + (size_t)nbAddr;
    }
    return "";
}

// lzss/decompress.go:17-84
inline Bytes Decompress(const Bytes &data, const Bytes &dict, int device = 0, size_t max_out = size_t(1) << 26) {
    Header h;
    try {
        h = Header::ReadFrom(data.data(), data.size());
    } catch (const Error &e) {
        throw Error(std::string("failed to read header: ") + e.what());
    }
    if (h.Version != lzss::Version) throw Error("unsupported compressor version");
    if (h.NoCompression) return Bytes(data.begin() + HeaderSize, data.end());
    // a decode-only context: the dictionary's text on the device, no index (decompress.go:34 only augments the dictionary)
    mi355_lzss_ctx *ctx = nullptr;
    check(mi355_lzss_ctx_create_decoder(dict.empty() ? nullptr : dict.data(), dict.size(), device, &ctx));
    size_t cap = (data.size() - HeaderSize) * 8 / 30 * 256 + 512;
    if (cap > max_out) cap = max_out;
    Bytes out(cap);
    const uint64_t in_off[2] = {0, data.size()}, out_off[2] = {0, cap};
    uint32_t out_len = 0;
    int32_t status = 0;
    int rc = mi355_lzss_decompress_batch(ctx, data.data(), in_off, 1, out.data(), out_off, &out_len, &status);
    mi355_lzss_ctx_destroy(ctx);
    check(rc);
    if (status == MI355_LZSS_E_FORMAT) {
        const std::string msg = DecodeErrorText(data, AugmentDict(dict).size());
        if (!msg.empty()) throw Error(msg, status);
    }
    if (status != 0) throw Error("invalid compressed stream", status);
    out.resize(out_len);
    return out;
}

}  // namespace lzss
