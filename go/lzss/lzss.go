// Package lzss is the cgo shim that keeps the API of github.com/consensys/compress/lzss
// (lzss/compress.go, lzss/decompress.go, lzss/header.go) and routes the hot path - index build,
// per-position longest-back-reference search, phrase walk, bit emission, decompression - to the
// MI355X library libmi355_lzss.so through its C ABI (include/mi355_lzss.h).
//
// NOTE: the build image of this repository has no Go toolchain (round 2 probed again: `go: command not found`), so
// this file has never been compiled;
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
	"bytes"
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"errors"
	"fmt"
	"io"
	"runtime"
	"strconv"
	"sync"
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

// WriteTo: lzss/header.go:22-32 (two writes, like the reference: a failing second write reports 2 bytes)
func (h *Header) WriteTo(w io.Writer) (int64, error) {
	b := h.bytes()
	if _, err := w.Write(b[:2]); err != nil {
		return 0, err
	}
	if _, err := w.Write(b[2:]); err != nil {
		return 2, err
	}
	return HeaderSize, nil
}

// ReadFrom: lzss/header.go:34-44
func (h *Header) ReadFrom(r io.Reader) (int64, error) {
	var b [HeaderSize]byte
	n, err := io.ReadFull(r, b[:])
	if err != nil {
		return int64(n), err
	}
	h.Version = binary.BigEndian.Uint16(b[:2])
	switch b[2] {
	case 0:
		h.NoCompression = false
	case 1:
		h.NoCompression = true
	default:
		return int64(n), errors.New("expected 0 or 1")
	}
	return int64(n), nil
}

// BackrefType: lzss/backref.go:21-30.  The exported fields describe the two phrase layouts of the format; the
// library works with the same constants (include/lzss_format.h).
type BackrefType struct {
	Delimiter      byte
	NbBitsAddress  uint8
	NbBitsLength   uint8
	NbBitsBackRef  uint8
	nbBytesBackRef int
	maxAddress     int
	maxLength      int
	DictLen        int
}

const (
	maxBackrefLenLog2 = 8  // lzss/backref.go:17
	shortAddrBits     = 14 // lzss/backref.go:18
	dynamicAddrBits   = 21 // lzss/backref.go:38 (fixed bound)
)

func newBackRefType(symbol byte, nbBitsAddress, nbBitsLength uint8, dictLen int) BackrefType { // backref.go:42-53
	total := 8 + nbBitsAddress + nbBitsLength
	return BackrefType{
		Delimiter:      symbol,
		NbBitsAddress:  nbBitsAddress,
		NbBitsLength:   nbBitsLength,
		NbBitsBackRef:  total,
		nbBytesBackRef: (int(total) + 7) / 8,
		maxAddress:     1 << nbBitsAddress,
		maxLength:      1 << nbBitsLength,
		DictLen:        dictLen,
	}
}

// NewShortBackrefType: lzss/backref.go:32-35
func NewShortBackrefType() BackrefType {
	return newBackRefType(SymbolShort, shortAddrBits, maxBackrefLenLog2, 0)
}

// NewDynamicBackrefType: lzss/backref.go:37-40 (addressableBytes is ignored there too: the bound is fixed at 21 bits)
func NewDynamicBackrefType(dictLen, addressableBytes int) BackrefType {
	return newBackRefType(SymbolDynamic, dynamicAddrBits, maxBackrefLenLog2, dictLen)
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
	// Close releases the device side deterministically; the finalizer is only a safety net for callers written
	// against the reference, which has nothing to release
	runtime.SetFinalizer(c, func(c *Compressor) { c.Close() })
	c.Reset()
	return c, nil
}

// Close releases the device memory and streams of the compressor (no counterpart in the reference).  The
// compressor must not be used afterwards.  Safe to call more than once.
func (c *Compressor) Close() {
	if c.ctx != nil {
		C.mi355_lzss_ctx_destroy(c.ctx)
		c.ctx = nil
		runtime.SetFinalizer(c, nil)
	}
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

// CompressedSize256k: lzss/compress.go:384-407.  "state less and thread-safe" like the reference's
// (compress.go:381-383): mi355_lzss_compressed_size works in a pool workspace of its own per call, so several
// goroutines may call it on one Compressor at once (the other methods remain single-goroutine).
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
	var dec *decoderCtx
	for {
		d, err := decoder(dict)
		if err != nil {
			return nil, err
		}
		d.mu.Lock() // a ctx is not goroutine-safe
		if d.ctx != nil {
			dec = d
			break
		}
		d.mu.Unlock() // evicted (destroyed) between the cache lookup and the lock: fetch or rebuild it
	}
	defer dec.mu.Unlock()
	ctx := dec.ctx
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
		if status == C.MI355_LZSS_E_FORMAT {
			if err := decodeError(data, len(AugmentDict(dict))); err != nil {
				return nil, err
			}
		}
		return nil, fmt.Errorf("invalid compressed stream (status %d)", int(status))
	}
	return out[:outLen], nil
}

// decodeError rebuilds the reference decoder's message for a stream the device rejected as malformed
// (lzss/decompress.go:53-55,66-69): a walk over the phrase boundaries finds the offending phrase (no bytes are
// decoded).  io.EOF when the stream ends inside a phrase.  dictLen: length of the augmented dictionary.
func decodeError(data []byte, dictLen int) error {
	r := bitWalker{data: data, pos: HeaderSize * 8}
	total := len(data) * 8
	outLen := 0
	for r.pos+8 <= total {
		s := byte(r.read(8))
		if s != SymbolShort && s != SymbolDynamic {
			outLen++
			continue
		}
		nbAddr := 14
		if s == SymbolDynamic {
			nbAddr = 21
		}
		if r.pos+8+nbAddr > total {
			return io.EOF // readFrom returns the bit reader's TryError (backref.go:77-79)
		}
		b := backref{length: int(r.read(8)) + 1, address: int(r.read(nbAddr)) + 1}
		if s == SymbolShort {
			b.bType = NewShortBackrefType()
			if b.address > outLen {
				return fmt.Errorf("invalid short backref %+v - output buffer is only %d bytes long", b, outLen)
			}
		} else if b.address > outLen {
			b.bType = NewDynamicBackrefType(dictLen, outLen)
			dictStart := dictLen - (b.address - outLen)
			if dictStart < 0 || dictStart > dictLen || dictStart+b.length > dictLen {
				return fmt.Errorf("invalid dynamic backref %+v - dict is only %d bytes long; dictStart = %d", b, dictLen, dictStart)
			}
		}
		outLen += b.length
	}
	return nil
}

// backref: lzss/backref.go:55-59 (here only for the %+v of the decoder's messages)
type backref struct {
	address int
	length  int
	bType   BackrefType
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

// ---- decode-only contexts ---------------------------------------------------------------------------------
// The reference's Decompress builds no index (decompress.go:34 only augments the dictionary), so the shim keeps one
// decode-only context (mi355_lzss_ctx_create_decoder: the dictionary's text on the device, nothing else) per
// (dictionary, device) in a small cache instead of creating a full compression context per call.
type decoderCtx struct {
	mu  sync.Mutex
	ctx *C.mi355_lzss_ctx
}

type decoderKey struct {
	sum    [sha256.Size]byte
	n      int
	device int
}

var (
	decodersMu  sync.Mutex
	decoders    = map[decoderKey]*decoderCtx{}
	decoderKeys []decoderKey // insertion order, oldest first
)

const maxDecoders = 4

func decoder(dict []byte) (*decoderCtx, error) {
	key := decoderKey{sha256.Sum256(dict), len(dict), Device}
	decodersMu.Lock()
	defer decodersMu.Unlock()
	if d, ok := decoders[key]; ok {
		return d, nil
	}
	d := &decoderCtx{}
	if rc := C.mi355_lzss_ctx_create_decoder(ptr(dict), C.size_t(len(dict)), C.int(Device), &d.ctx); rc != 0 {
		return nil, lastError(rc)
	}
	if len(decoderKeys) >= maxDecoders {
		old := decoders[decoderKeys[0]]
		delete(decoders, decoderKeys[0])
		decoderKeys = decoderKeys[1:]
		old.mu.Lock() // wait for a Decompress still using it
		C.mi355_lzss_ctx_destroy(old.ctx)
		old.ctx = nil
		old.mu.Unlock()
	}
	decoders[key] = d
	decoderKeys = append(decoderKeys, key)
	return d, nil
}

// CompressedSizes256k is CompressedSize256k for many independent inputs in one launch sequence
// (mi355_lzss_compressed_size_batch).  No counterpart in the reference.
func (c *Compressor) CompressedSizes256k(blobs [][]byte) ([]int, error) {
	res := make([]int, len(blobs))
	if c.noCompression {
		for k, b := range blobs {
			res[k] = HeaderSize + len(b)
		}
		return res, nil
	}
	inOff := make([]C.uint64_t, len(blobs)+1)
	var flat []byte
	for k, b := range blobs {
		flat = append(flat, b...)
		inOff[k+1] = C.uint64_t(len(flat))
	}
	sizes := make([]C.uint64_t, len(blobs)+1)
	if rc := C.mi355_lzss_compressed_size_batch(c.ctx, ptr(flat), &inOff[0], C.uint32_t(len(blobs)), &sizes[0], nil); rc != 0 {
		return nil, lastError(rc)
	}
	for k := range blobs {
		res[k] = int(sizes[k])
	}
	return res, nil
}

// PackedBlob is one blob of PackStream: its bytes (what Bytes() holds when the loop stops) and the input it covers.
type PackedBlob struct {
	Bytes    []byte
	Consumed int
}

// PackStream fills size-capped blobs from a stream exactly like the Write / Revert loop of the reference's callers
// (lzss/compress_test.go:337-353: Write chunk after chunk while Len() < outMax, Revert the chunk that overshot), but
// runs the candidate Writes of a blob as device batches (mi355_lzss_pack_blob, lookahead of them at a time).  Blobs in
// which ConsiderBypassing would fire run the loop itself on c.  No counterpart in the reference.
func (c *Compressor) PackStream(data []byte, chunk, outMax, lookahead int) ([]PackedBlob, error) {
	var res []PackedBlob
	out := make([]byte, outMax+16)
	for i0 := 0; i0 < len(data); {
		var outLen, chunks C.uint64_t
		var needsLoop C.int
		rest := data[i0:]
		if rc := C.mi355_lzss_pack_blob(c.ctx, ptr(rest), C.uint64_t(len(rest)), C.uint32_t(chunk), C.uint64_t(outMax),
			C.uint32_t(lookahead), ptr(out), C.uint64_t(len(out)), &outLen, &chunks, &needsLoop); rc != 0 {
			return nil, lastError(rc)
		}
		var blob []byte
		k := int(chunks)
		if needsLoop != 0 { // the loop of compress_test.go:337-353 on this compressor
			c.Reset()
			i := i0
			for i < len(data) && c.Len() < outMax {
				end := i + chunk
				if end > len(data) {
					end = len(data)
				}
				if _, err := c.Write(data[i:end]); err != nil {
					return nil, err
				}
				uncompressed := i + chunk - i0 + 3
				if c.Len() >= outMax && uncompressed <= outMax && c.Len() > uncompressed {
					c.ConsiderBypassing()
				}
				i += chunk
			}
			if c.Len() > outMax {
				if err := c.Revert(); err != nil {
					return nil, err
				}
				i -= chunk
			}
			blob = append([]byte(nil), c.Bytes()...)
			k = (i - i0) / chunk
		} else {
			blob = append([]byte(nil), out[:int(outLen)]...)
		}
		if k <= 0 {
			return nil, errors.New("a single chunk does not fit the output cap")
		}
		consumed := k * chunk
		if i0+consumed > len(data) {
			consumed = len(data) - i0
		}
		res = append(res, PackedBlob{Bytes: blob, Consumed: consumed})
		i0 += k * chunk
	}
	c.Reset()
	return res, nil
}

// CompressBatchMulti is CompressBatch over several compressors, one per GPU of the node (create them with
// Device = 0 .. n-1 and the same dictionary): contiguous blob ranges balanced by bytes, one host thread per device,
// no collective (mi355_lzss_compress_batch_multi).  No counterpart in the reference.
func CompressBatchMulti(cs []*Compressor, blobs [][]byte) ([][]byte, error) {
	nb := len(blobs)
	ctxs := make([]*C.mi355_lzss_ctx, len(cs))
	for k, c := range cs {
		ctxs[k] = c.ctx
	}
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
	if rc := C.mi355_lzss_compress_batch_multi(&ctxs[0], C.int(len(ctxs)), ptr(flat), &inOff[0], C.uint32_t(nb), ptr(out),
		&outOff[0], &outLen[0], &status[0]); rc != 0 {
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

// ---- stream diagnostics: lzss/decompress.go:86-236 -----------------------------------------------------------

// CompressionPhrase: lzss/decompress.go:86-93
type CompressionPhrase struct {
	Type              byte
	Length            int
	ReferenceAddress  int
	StartDecompressed int
	StartCompressed   int
	Content           []byte
}

// CompressionPhrases: lzss/decompress.go:95
type CompressionPhrases []CompressionPhrase

// bitWalker reads MSB-first with bitio's try-semantics: a read past the end latches eof and returns 0
type bitWalker struct {
	data []byte
	pos  int // in bits
	eof  bool
}

func (r *bitWalker) read(nb int) uint64 {
	if r.eof || r.pos+nb > len(r.data)*8 {
		r.eof = true
		return 0
	}
	var v uint64
	for k := 0; k < nb; k++ {
		bit := (r.data[r.pos>>3] >> (7 - uint(r.pos&7))) & 1
		v = v<<1 | uint64(bit)
		r.pos++
	}
	return v
}

// CompressedStreamInfo: lzss/decompress.go:97-206.  The decompressed bytes come from the GPU decompressor; the
// phrase boundaries are a walk over the bit stream with the reference's conventions (positions count the augmented
// dictionary in front of the output, consecutive literals form one phrase, StartCompressed counts bits after the header).
func CompressedStreamInfo(c, dict []byte) (CompressionPhrases, error) {
	var header Header
	sizeHeader, err := header.ReadFrom(bytes.NewReader(c))
	if err != nil {
		return nil, err
	}
	if header.Version != Version {
		panic("unsupported compressor version") // decompress.go:106-108
	}
	if header.NoCompression {
		return CompressionPhrases{{Length: len(c) - int(sizeHeader), Content: c[sizeHeader:]}}, nil
	}
	plain, err := Decompress(c, dict)
	if err != nil {
		return nil, err
	}
	aug := AugmentDict(append([]byte(nil), dict...))
	out := append(aug, plain...)
	var res CompressionPhrases
	r := &bitWalker{data: c, pos: HeaderSize * 8}
	pos, inI, litStart := len(out)-len(plain), 0, -1
	flush := func() {
		if litStart == -1 {
			return
		}
		res = append(res, CompressionPhrase{Type: 0, Length: pos - litStart, ReferenceAddress: litStart,
			StartDecompressed: litStart, StartCompressed: inI, Content: out[litStart:pos]})
		inI += (pos - litStart) * 8
		litStart = -1
	}
	for s := byte(r.read(8)); !r.eof; s = byte(r.read(8)) {
		if s == SymbolShort || s == SymbolDynamic {
			flush()
			bt := NewShortBackrefType()
			if s == SymbolDynamic {
				bt = NewDynamicBackrefType(0, pos)
			}
			length := int(r.read(int(bt.NbBitsLength))) + 1 // backref.readFrom, backref.go:70-86
			address := int(r.read(int(bt.NbBitsAddress))) + 1
			if r.eof {
				return nil, io.EOF
			}
			if address > pos {
				return nil, fmt.Errorf("invalid back reference: address %d at %d", address, pos)
			}
			pos += length
			res = append(res, CompressionPhrase{Type: s, Length: length, ReferenceAddress: pos - length - address,
				StartDecompressed: pos - length, StartCompressed: inI, Content: out[pos-length : pos]})
			inI += int(bt.NbBitsBackRef)
		} else {
			if litStart == -1 {
				litStart = pos
			}
			pos++
		}
	}
	flush()
	if pos != len(out) {
		return nil, errors.New("phrase walk and decompressor disagree on the output length")
	}
	return res, nil
}

// ToCSV: lzss/decompress.go:208-236
func (c CompressionPhrases) ToCSV() []byte {
	var b bytes.Buffer
	b.WriteString("type,length,start_decompressed (bytes),start_compressed (bits),reference_address,content (hex)\n")
	for _, p := range c {
		switch p.Type {
		case SymbolShort:
			b.WriteString("short,")
		case SymbolDynamic:
			b.WriteString("long,")
		case 0:
			b.WriteString("literal,")
		default:
			panic("unknown phrase type")
		}
		for _, v := range []int{p.Length, p.StartDecompressed, p.StartCompressed, p.ReferenceAddress} {
			b.WriteString(strconv.Itoa(v))
			b.WriteString(",")
		}
		b.WriteString(hex.EncodeToString(p.Content))
		b.WriteString("\n")
	}
	return b.Bytes()
}
