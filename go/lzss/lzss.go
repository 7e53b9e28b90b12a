// Package lzss is the cgo shim that keeps the API of github.com/consensys/compress/lzss
// (lzss/compress.go, lzss/decompress.go, lzss/header.go) and routes the hot path - index build,
// per-position longest-back-reference search, phrase walk, bit emission, decompression - to the
// MI355X library libmi355_lzss.so through its C ABI (include/mi355_lzss.h).
//
// NOTE: the build image of this repository has no Go toolchain, so this file has never been compiled;
// its logic is the same state machine as compress_amd/lzss.py and compress_amd/hostcpp/lzss.hpp, which
// the test-suite exercises byte for byte against the oracle.  See INTEGRATION.md.
package lzss

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../compress_amd -lmi355_lzss
#include <stdlib.h>
#include "mi355_lzss.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"fmt"
	"runtime"
	"unsafe"
)

const (
	MaxInputSize = 1 << 22 // lzss/backref.go:10
	MaxDictSize  = 1 << 22 // lzss/backref.go:11

	SymbolDynamic byte = 0xFF // lzss/backref.go:15
	SymbolShort   byte = 0xFE // lzss/backref.go:16

	Version    = 1 // lzss/header.go:11
	HeaderSize = 3 // lzss/header.go:12
)

// Device selects the GPU a new Compressor / Decompress call uses.
var Device = 0

func lastError(rc C.int) error {
	return fmt.Errorf("%s", C.GoString(C.mi355_lzss_last_error()))
}

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// Header: lzss/header.go:17-44
type Header struct {
	Version       uint16
	NoCompression bool
}

func (h *Header) bytes() []byte {
	b := make([]byte, HeaderSize)
	binary.BigEndian.PutUint16(b, h.Version)
	if h.NoCompression {
		b[2] = 1
	}
	return b
}

// AugmentDict: lzss/compress.go:70-88
func AugmentDict(dict []byte) []byte {
	found := uint8(0)
	for _, b := range dict {
		if b == SymbolShort {
			found |= 0b010
		} else if b == SymbolDynamic {
			found |= 0b100
		} else {
			continue
		}
		if found == 0b110 {
			return dict
		}
	}
	return append(dict, SymbolShort, SymbolDynamic)
}

// Compressor: lzss/compress.go:11-32.  Not goroutine-safe (like the reference).
type Compressor struct {
	ctx *C.mi355_lzss_ctx

	outBuf, inBuf                     []byte
	nbSkippedBits, lastNbSkippedBits  uint8
	lastOutLen, lastInLen             int
	noCompression                     bool
}

// NewCompressor: lzss/compress.go:37-67
func NewCompressor(dict []byte) (*Compressor, error) {
	if len(AugmentDict(dict)) > MaxDictSize {
		return nil, fmt.Errorf("dict size must be <= %d", MaxDictSize)
	}
	c := &Compressor{}
	if rc := C.mi355_lzss_ctx_create(ptr(dict), C.size_t(len(dict)), C.int(Device), &c.ctx); rc != 0 {
		return nil, lastError(rc)
	}
	runtime.SetFinalizer(c, func(c *Compressor) { C.mi355_lzss_ctx_destroy(c.ctx) })
	c.Reset()
	return c, nil
}

// Reset: lzss/compress.go:289-304
func (c *Compressor) Reset() {
	c.noCompression = false
	h := Header{Version: Version}
	c.outBuf = append(c.outBuf[:0], h.bytes()...)
	c.inBuf = c.inBuf[:0]
	c.lastOutLen = len(c.outBuf)
	c.lastNbSkippedBits = 0
	c.nbSkippedBits = 0
	c.lastInLen = 0
}

// Write: lzss/compress.go:91-129
func (c *Compressor) Write(d []byte) (n int, err error) {
	c.lastOutLen = len(c.outBuf)
	last := c.outBuf[len(c.outBuf)-1] >> c.nbSkippedBits
	c.outBuf = c.outBuf[:len(c.outBuf)-1]
	pending := uint(8 - c.nbSkippedBits)
	if pending == 8 {
		c.outBuf = append(c.outBuf, last)
		pending = 0
	}
	c.lastNbSkippedBits = c.nbSkippedBits
	if len(c.inBuf)+len(d) > MaxInputSize {
		return 0, fmt.Errorf("input size must be <= %d", MaxInputSize)
	}
	c.lastInLen = len(c.inBuf)
	c.inBuf = append(c.inBuf, d...)
	if c.noCompression {
		c.outBuf = append(c.outBuf, d...)
		return len(d), nil
	}
	start := c.lastInLen
	buf := make([]byte, (int(pending)+37*(len(c.inBuf)-start)+7)/8+16)
	var nbits C.uint64_t
	if rc := C.mi355_lzss_write(c.ctx, ptr(c.inBuf), C.size_t(len(c.inBuf)), C.size_t(start), C.uint(pending),
		ptr(buf), C.size_t(len(buf)), &nbits); rc != 0 {
		return 0, lastError(rc)
	}
	total := uint64(pending) + uint64(nbits)
	nbytes := int((total + 7) / 8)
	if nbytes > 0 {
		if pending > 0 {
			buf[0] |= last << (8 - pending)
		}
		c.outBuf = append(c.outBuf, buf[:nbytes]...)
	}
	c.nbSkippedBits = uint8((8 - total%8) % 8)
	return len(c.inBuf) - start, nil
}

func (c *Compressor) Len() int             { return len(c.outBuf) } // compress.go:307
func (c *Compressor) Written() int         { return len(c.inBuf) }  // compress.go:312
func (c *Compressor) WrittenBytes() []byte { return c.inBuf }       // compress.go:318
func (c *Compressor) Bytes() []byte        { return c.outBuf }      // compress.go:370

// Revert: lzss/compress.go:324-345
func (c *Compressor) Revert() error {
	if c.lastInLen == -1 {
		return fmt.Errorf("cannot revert twice in a row")
	}
	c.inBuf = c.inBuf[:c.lastInLen]
	c.lastInLen = -1
	if c.noCompression {
		in := append([]byte(nil), c.inBuf...)
		c.Reset()
		if _, err := c.Write(in); err != nil {
			return err
		}
		c.ConsiderBypassing()
		return nil
	}
	c.outBuf = c.outBuf[:c.lastOutLen]
	c.nbSkippedBits = c.lastNbSkippedBits
	return nil
}

// ConsiderBypassing: lzss/compress.go:348-367
func (c *Compressor) ConsiderBypassing() (bypassed bool) {
	if len(c.outBuf) > len(c.inBuf)+HeaderSize {
		c.noCompression = true
		c.nbSkippedBits = 0
		c.lastOutLen = c.lastInLen + HeaderSize
		c.lastNbSkippedBits = 0
		h := Header{Version: Version, NoCompression: true}
		c.outBuf = append(c.outBuf[:0], h.bytes()...)
		c.outBuf = append(c.outBuf, c.inBuf...)
		return true
	}
	return false
}

// Compress: lzss/compress.go:375-379
func (c *Compressor) Compress(d []byte) ([]byte, error) {
	c.Reset()
	_, err := c.Write(d)
	return c.Bytes(), err
}

// CompressedSize256k: lzss/compress.go:384-407
func (c *Compressor) CompressedSize256k(d []byte) (int, error) {
	if c.noCompression {
		return HeaderSize + len(d), nil
	}
	if len(d) > 1<<18 {
		return 0, fmt.Errorf("input size must be <= %d", 1<<18)
	}
	var size C.uint64_t
	if rc := C.mi355_lzss_compressed_size(c.ctx, ptr(d), C.size_t(len(d)), &size); rc != 0 {
		return 0, lastError(rc)
	}
	return int(size), nil
}

// Decompress: lzss/decompress.go:17-84
func Decompress(data, dict []byte) ([]byte, error) {
	if len(data) < HeaderSize {
		return nil, fmt.Errorf("failed to read header: %w", errors.New("unexpected EOF"))
	}
	if data[2] > 1 {
		return nil, fmt.Errorf("failed to read header: %w", errors.New("expected 0 or 1"))
	}
	if binary.BigEndian.Uint16(data[:2]) != Version {
		return nil, errors.New("unsupported compressor version")
	}
	if data[2] == 1 {
		return data[HeaderSize:], nil
	}
	var ctx *C.mi355_lzss_ctx
	if rc := C.mi355_lzss_ctx_create(ptr(dict), C.size_t(len(dict)), C.int(Device), &ctx); rc != 0 {
		return nil, lastError(rc)
	}
	defer C.mi355_lzss_ctx_destroy(ctx)
	capOut := (len(data)-HeaderSize)*8/30*256 + 512
	out := make([]byte, capOut)
	inOff := [2]C.uint64_t{0, C.uint64_t(len(data))}
	outOff := [2]C.uint64_t{0, C.uint64_t(capOut)}
	var outLen C.uint32_t
	var status C.int32_t
	if rc := C.mi355_lzss_decompress_batch(ctx, ptr(data), &inOff[0], 1, ptr(out), &outOff[0], &outLen, &status); rc != 0 {
		return nil, lastError(rc)
	}
	if status != 0 {
		return nil, fmt.Errorf("invalid compressed stream (status %d)", int(status))
	}
	return out[:outLen], nil
}

// CompressBatch compresses independent blobs that share c's dictionary in one device launch
// (one-shot Compress per blob, lzss/compress.go:375-379).  No counterpart in the reference.
func (c *Compressor) CompressBatch(blobs [][]byte) ([][]byte, error) {
	nb := len(blobs)
	inOff := make([]C.uint64_t, nb+1)
	outOff := make([]C.uint64_t, nb+1)
	var flat []byte
	for k, b := range blobs {
		flat = append(flat, b...)
		inOff[k+1] = C.uint64_t(len(flat))
		outOff[k+1] = outOff[k] + C.uint64_t(HeaderSize+(37*len(b)+7)/8+1)
	}
	out := make([]byte, int(outOff[nb])+1)
	outLen := make([]C.uint32_t, nb+1)
	status := make([]C.int32_t, nb+1)
	if rc := C.mi355_lzss_compress_batch(c.ctx, ptr(flat), &inOff[0], C.uint32_t(nb), ptr(out), &outOff[0],
		&outLen[0], &status[0]); rc != 0 {
		return nil, lastError(rc)
	}
	res := make([][]byte, nb)
	for k := range blobs {
		if status[k] != 0 {
			return nil, fmt.Errorf("blob %d: status %d", k, int(status[k]))
		}
		res[k] = out[outOff[k] : outOff[k]+C.uint64_t(outLen[k])]
	}
	return res, nil
}
