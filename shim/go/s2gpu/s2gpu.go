// Package s2gpu adapts the MI355X engine to s2.WriterCustomEncoder (s2/writer.go:1053-1064)
// and offers the batched block form.  Source only (no Go toolchain in the build image).
package s2gpu

/*
#cgo LDFLAGS: -lkcgpu
#include "kcgpu.h"
*/
import "C"

import (
	"errors"
	"sync"
	"unsafe"

	"github.com/klauspost/compress/s2"
)

// Ctx wraps one kc_ctx.  CustomEncoder's function is safe for any number of goroutines (the library micro-batches concurrent
// callers); the batched forms hold mu (one device batch per context at a time).
type Ctx struct {
	c  *C.kc_ctx
	mu sync.Mutex
}

// NewCtx creates a device context whose S2 blocks are byte-identical to what THIS build of the reference writes: on amd64 the
// reference's block encoders are generated assembly with their own table sizes, hash lengths and skip rates per input size
// (s2/encode_amd64.go), elsewhere — and under the `noasm` / `appengine` build tags — portable Go (s2/encode_all.go).  The device
// implements both (KC_OPT_S2_VARIANT); defaultVariant follows the build constraints of s2/encode_amd64.go (variant_*.go).
func NewCtx(device int) (*Ctx, error) {
	var c *C.kc_ctx
	if st := C.kc_ctx_create(&c, C.int(device), nil); st != C.KC_OK {
		return nil, errors.New("no MI355X device")
	}
	x := &Ctx{c: c}
	x.SetVariant(defaultVariant)
	return x, nil
}

// Variants of the reference's block encoders (levels default, better and their Snappy-compatible forms; the best levels are
// pure Go everywhere).
const (
	VariantGo    = 0 // s2/encode_all.go, s2/encode_better.go
	VariantAMD64 = 1 // s2/encodeblock_amd64.s
)

func (x *Ctx) SetVariant(v int) { C.kc_ctx_set_option(x.c, C.int(C.KC_OPT_S2_VARIANT), C.int64_t(v)) }

// SetHookLanes: how many batches of concurrent CustomEncoder callers run on the device at once (KC_OPT_S2_HOOK_LANES, default 4,
// at most 8; before the first call of the hook).
func (x *Ctx) SetHookLanes(n int) { C.kc_ctx_set_option(x.c, C.int(C.KC_OPT_S2_HOOK_LANES), C.int64_t(n)) }

// LastBatches reports how many device batches the last EncodeBlocks* call was cut into (0: the call did not reach the device).
func (x *Ctx) LastBatches() int {
	return int(C.kc_ctx_get_option(x.c, C.int(C.KC_OPT_LAST_BATCHES)))
}

func (x *Ctx) Close() { C.kc_ctx_destroy(x.c) }

// CustomEncoder returns the function to pass to s2.WriterCustomEncoder: bytes used, 0 = incompressible,
// <0 = use the built-in encoder.
func CustomEncoder(x *Ctx) func(dst, src []byte) int {
	return func(dst, src []byte) int {
		if len(src) == 0 || len(dst) == 0 {
			return -1
		}
		return int(C.kc_s2_encode_block(x.c, (*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)),
			(*C.uint8_t)(unsafe.Pointer(&src[0])), C.uint64_t(len(src))))
	}
}

// Levels of EncodeBlocksLevel.
const (
	LevelDefault      = 0 // s2.Encode
	LevelBetter       = 1 // s2.EncodeBetter
	LevelSnappy       = 2 // s2.EncodeSnappy
	LevelSnappyBetter = 3 // s2.EncodeSnappyBetter
	LevelBest         = 4 // s2.EncodeBest
	LevelSnappyBest   = 5 // s2.EncodeSnappyBest
)

// EncodeBlocks == N x s2.Encode(nil, src[off[i]:off[i+1]]).
func EncodeBlocks(x *Ctx, src []byte, off []uint64, dst []byte) ([]byte, []uint64, error) {
	return EncodeBlocksLevel(x, LevelDefault, src, off, dst)
}

// EncodeBlocksLevel == N x s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter / EncodeBest / EncodeSnappyBest (nil, src[off[i]:off[i+1]]).
func EncodeBlocksLevel(x *Ctx, level int, src []byte, off []uint64, dst []byte) ([]byte, []uint64, error) {
	n := len(off) - 1
	outOff := make([]uint64, n+1)
	if n <= 0 {
		return dst[:0], []uint64{0}, nil
	}
	need := 0
	for i := 0; i < n; i++ {
		need += (s2.MaxEncodedLen(int(off[i+1]-off[i])) + 15) &^ 15 // (block slots are 16-byte aligned on the device)
	}
	if cap(dst) < need+64 {
		dst = make([]byte, need+64) // dst may be nil or short, like the dst of s2.Encode
	}
	dst = dst[:cap(dst)]
	if len(src) == 0 { // n empty blocks: nothing to point the device at
		return encodeRef(level, src, off, dst, outOff)
	}
	x.mu.Lock()
	st := C.kc_s2_encode_blocks_lvl(x.c, C.int(level), (*C.uint8_t)(unsafe.Pointer(&src[0])), (*C.uint64_t)(unsafe.Pointer(&off[0])), C.uint32_t(n),
		(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
	var msg string
	if st != C.KC_OK {
		msg = C.GoString(C.kc_last_error(x.c))
	}
	x.mu.Unlock()
	if st == C.KC_OK {
		return dst[:outOff[n]], outOff, nil
	}
	if st != C.KC_ERR_UNSUPPORTED && st != C.KC_ERR_NO_DEVICE {
		return nil, nil, errors.New(msg)
	}
	// not served by the device (block above 1 GiB, device memory exhausted, ...): the reference encoder, same bytes
	return encodeRef(level, src, off, dst, outOff)
}

func encodeRef(level int, src []byte, off []uint64, dst []byte, outOff []uint64) ([]byte, []uint64, error) {
	n := len(off) - 1
	enc := [...]func(dst, src []byte) []byte{s2.Encode, s2.EncodeBetter, s2.EncodeSnappy, s2.EncodeSnappyBetter, s2.EncodeBest, s2.EncodeSnappyBest}[level]
	out := dst[:0]
	for i := 0; i < n; i++ {
		outOff[i] = uint64(len(out))
		out = append(out, enc(nil, src[off[i]:off[i+1]])...)
	}
	outOff[n] = uint64(len(out))
	return out, outOff, nil
}
