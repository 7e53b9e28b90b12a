// decompress.hip - batched per-blob decompressor (placeholder translation unit until the kernel lands).
#include "kernels.h"
