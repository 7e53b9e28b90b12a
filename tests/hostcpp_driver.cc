// Test driver for compress_amd/hostcpp/lzss.hpp: one-shot Compress, a Write/Len/Revert flow, Decompress.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "../compress_amd/hostcpp/lzss.hpp"

static lzss::Bytes slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return lzss::Bytes(std::istreambuf_iterator<char>(f), {});
}
static void put(std::ofstream &o, const lzss::Bytes &b) {
    uint32_t n = (uint32_t)b.size();
    o.write((const char *)&n, 4);
    o.write((const char *)b.data(), (std::streamsize)b.size());
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    try {
        const lzss::Bytes dict = slurp(argv[1]), data = slurp(argv[2]);
        std::ofstream out(argv[3], std::ios::binary);
        lzss::Compressor c(dict);
        const lzss::Bytes one = c.Compress(data);
        put(out, one);
        c.Reset();
        const size_t chunk = 1000, cap = 4000;
        const auto packed = c.PackStream(data, chunk, cap, 8);  // must equal the loop below, blob for blob
        size_t blob_no = 0;
        for (size_t i0 = 0; i0 < data.size();) {
            size_t i = i0;
            for (; i < data.size() && c.Len() < cap; i += chunk)
                c.Write(data.data() + i, std::min(chunk, data.size() - i));
            if (c.Len() > cap) {
                c.Revert();
                i -= chunk;
            }
            put(out, c.GetBytes());
            if (blob_no >= packed.size() || packed[blob_no].first != c.GetBytes() ||
                packed[blob_no].second != std::min(i, data.size()) - i0) {
                fprintf(stderr, "PackStream differs from the Write / Revert loop at blob %zu\n", blob_no);
                return 3;
            }
            ++blob_no;
            c.Reset();
            i0 = i;
        }
        if (blob_no != packed.size()) {
            fprintf(stderr, "PackStream made %zu blobs, the loop %zu\n", packed.size(), blob_no);
            return 3;
        }
        put(out, lzss::Decompress(one, dict));
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
