"""compress_amd - MI355X-native lzss match-finding path (Consensys/compress drop-in).

The package is a thin host-side mirror of the reference's Go API (lzss/compress.go, lzss/decompress.go)
over the C ABI of libmi355_lzss.so (include/mi355_lzss.h).  It has no CPU path: without the built HIP
library and a GPU every data-path call raises.
"""
from .lzss import (  # noqa: F401
    MaxInputSize, MaxDictSize, SymbolDynamic, SymbolShort, Version, HeaderSize,
    Compressor, NewCompressor, AugmentDict, Decompress, Header, LzssError,
    compress_batch, decompress_batch, compress_batch_multi, decompress_batch_multi, partition, PinnedBuffer,
)
from .info import CompressionPhrase, CompressedStreamInfo, ToCSV  # noqa: F401,E402
