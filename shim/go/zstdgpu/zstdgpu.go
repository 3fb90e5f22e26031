// Package zstdgpu is the cgo shim that lets klauspost/compress users route batched
// EncodeAll work to the MI355X engine (libkcgpu.so, include/kcgpu.h).  Source only: the
// build image has no Go toolchain.  The GPU path and the reference path produce the same
// bytes; anything the device path does not implement falls back to the reference encoder.
package zstdgpu

/*
#cgo LDFLAGS: -lkcgpu
#include <stdlib.h>
#include "kcgpu.h"
*/
import "C"

import (
	"bytes"
	"crypto/rand"
	"errors"
	"io"
	"runtime"
	"sync"
	"unsafe"

	"github.com/klauspost/compress/zstd"
)

// Option mirrors zstd.EOption for the options that change output bytes.
type Option func(e *Encoder) error

// DefaultDeviceMinBytes is the call size from which the device path is taken.  Below it the reference encoder on the host cores
// is faster: one 128 KiB unit takes ~0.3 ms on a CPU core and ~16 ms on the device (LDS-table kernel, profiles/r03_latency*.json),
// a batch of 4096 units ~70 ms on either (16 host cores vs profiles/r03_crossover_zfast.csv), and from there on the device wins.
const DefaultDeviceMinBytes = 512 << 20

// DefaultDeviceMinUnits is the number of independent units (frames, streams) a call must carry to be device work whatever its
// size: a unit is parsed by one 8-lane group or one wave, so ONE stream of 1 GiB is minutes on the device and seconds on a host
// core, while 2048 units in flight is where the device overtakes 16 host cores (profiles/r03_crossover_zfast.csv).  A Writer's
// stream is one unit: it stays on the reference encoder unless the caller lowers this (WithDeviceMinUnits; WithDeviceMinBytes(0)
// lowers both) or asks for device jobs (WithConcurrentBlocks + WithDeviceJobs: the jobs are the units).
const DefaultDeviceMinUnits = 2048

// writerBufferCap bounds what a Writer holds before it gives up on the device and streams through the reference encoder.
const writerBufferCap = 1 << 30

// Encoder is safe for concurrent use like (*zstd.Encoder).EncodeAll (zstd/encoder.go:90-99, 722-729: every call takes an
// encoder state from a channel of `concurrent` states): every call here takes a kc_ctx from a pool of the same size
// (WithEncoderConcurrency, default GOMAXPROCS); contexts are created on first use and returned after the call.
type Encoder struct {
	device   int
	opts     C.kc_zstd_opts
	cpuOpts  []zstd.EOption
	cpu      *zstd.Encoder  // EncodeAll on it is concurrency-safe; streaming use goes through refMu
	refMu    sync.Mutex
	dictMem  unsafe.Pointer // C copy of the dictionary (kc_zstd_opts.dict points into it)
	conc     int
	minBytes int
	minUnits int
	pool     chan *C.kc_ctx // idle contexts
	poolMu   sync.Mutex
	created  int  // contexts created so far (<= conc)
	noDevice bool // kc_ctx_create failed once: reference only
	jobs     bool // WithConcurrentBlocks(true)
	devJobs  bool // WithDeviceJobs(true)
	hasDict  bool
	pad      int // WithEncoderPadding: applied on the host, behind the frames of either path
}

// WithDeviceJobs sends WithConcurrentBlocks streams to the device (kc_zstd_encode_jobs).  Off by default: the jobs of a stream are
// few and long (64 jobs of 16 MiB per GiB at SpeedFastest), and a job is parsed by ONE wave — 1 GiB takes ~2.5 s on the device
// (419 MB/s on the LDS-table kernel, DESIGN.md 4.4) where the reference's own job workers take a fraction of a second.  The bytes
// are the same either way; the device path is there for parity and for hosts without spare cores.
func WithDeviceJobs(b bool) Option {
	return func(e *Encoder) error {
		e.devJobs = b
		return nil
	}
}

// WithConcurrentBlocks mirrors zstd.WithConcurrentBlocks (zstd/encoder_options.go:340-353): a Writer's stream is cut into jobs
// of max(4 * window, 512 KiB) bytes, each encoded with the tail of the previous job as history (zstd/enc_jobs.go).  The jobs are
// independent units: this is the mode in which ONE large stream is device work (kc_zstd_encode_jobs).  As in the reference it has
// no effect with a dictionary or with WithEncoderConcurrency(1).
func WithConcurrentBlocks(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithConcurrentBlocks(b))
		e.jobs = b
		return nil
	}
}

// jobMode: the option is in force (zstd/encoder.go:81).
func (e *Encoder) jobMode() bool { return e.jobs && !e.hasDict && e.conc > 1 }

// EncodeJobs == zstd.NewWriter(w, opts..., zstd.WithConcurrentBlocks(true)); Write(src) with Flush after flushAt[i] bytes; Close().
func (e *Encoder) EncodeJobs(src []byte, flushAt []uint64) ([]byte, error) {
	var ctx *C.kc_ctx
	if len(src) > 0 && e.devJobs && e.useDevice(len(src), e.minUnits) && e.jobMode() { // (asked for: the jobs are the units)
		ctx = e.acquire()
	}
	if ctx != nil {
		defer e.release(ctx)
		js := int(C.kc_zstd_job_size(&e.opts))
		ov := int(C.kc_zstd_overlap_size(&e.opts))
		njobs := len(src)/js + len(flushAt) + 2
		dst := make([]byte, len(src)+njobs*(ov+(e.MaxEncodedSize(js+ov)-js-ov)+32)+4096)
		cuts := append([]uint64{}, flushAt...)
		cuts = append(cuts, 0) // never empty: &cuts[0] below
		var outLen C.uint64_t
		st := C.kc_zstd_encode_jobs(ctx, &e.opts, (*C.uint8_t)(unsafe.Pointer(&src[0])), C.uint64_t(len(src)),
			(*C.uint64_t)(unsafe.Pointer(&cuts[0])), C.uint64_t(len(flushAt)),
			(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), &outLen)
		if st == C.KC_OK {
			return dst[:outLen], nil
		}
		if st != C.KC_ERR_UNSUPPORTED && st != C.KC_ERR_NO_DEVICE {
			return nil, errors.New(C.GoString(C.kc_last_error(ctx)))
		}
	}
	// reference path
	var sink bytes.Buffer
	r, err := zstd.NewWriter(&sink, e.cpuOpts...)
	if err != nil {
		return nil, err
	}
	pos := uint64(0)
	for _, c := range flushAt {
		if c > uint64(len(src)) {
			c = uint64(len(src))
		}
		if c > pos {
			if _, err := r.Write(src[pos:c]); err != nil {
				return nil, err
			}
			pos = c
		}
		if err := r.Flush(); err != nil {
			return nil, err
		}
	}
	if _, err := r.Write(src[pos:]); err != nil {
		return nil, err
	}
	if err := r.Close(); err != nil {
		return nil, err
	}
	return sink.Bytes(), nil
}

// WithEncoderConcurrency mirrors zstd.WithEncoderConcurrency: how many calls may be in flight on this Encoder (device
// contexts in the pool, and the reference encoder's own concurrency).
func WithEncoderConcurrency(n int) Option {
	return func(e *Encoder) error {
		if n <= 0 {
			return errors.New("concurrency must be at least 1")
		}
		e.cpuOpts = append(e.cpuOpts, zstd.WithEncoderConcurrency(n))
		e.conc = n
		return nil
	}
}

// WithDeviceMinBytes sets the call size (sum of the unit lengths) from which the device path is taken; smaller calls go to the
// reference encoder, which gives the same bytes faster at that size.  0 sends everything to the device (tests): it also drops the
// unit-count rule (WithDeviceMinUnits).
func WithDeviceMinBytes(n int) Option {
	return func(e *Encoder) error {
		e.minBytes = n
		if n == 0 {
			e.minUnits = 0
		}
		return nil
	}
}

// WithDeviceMinUnits sets how many units a call must carry to be device work (DefaultDeviceMinUnits).
func WithDeviceMinUnits(n int) Option {
	return func(e *Encoder) error {
		e.minUnits = n
		return nil
	}
}

// acquire returns a context for one call, or nil when this call should use the reference encoder (no device).
func (e *Encoder) acquire() *C.kc_ctx {
	select {
	case c := <-e.pool:
		return c
	default:
	}
	e.poolMu.Lock()
	if e.noDevice {
		e.poolMu.Unlock()
		return nil
	}
	if e.created < e.conc {
		var c *C.kc_ctx
		if st := C.kc_ctx_create(&c, C.int(e.device), nil); st != C.KC_OK {
			e.noDevice = e.created == 0
			e.poolMu.Unlock()
			if e.noDevice {
				return nil
			}
			return <-e.pool
		}
		e.created++
		e.poolMu.Unlock()
		return c
	}
	e.poolMu.Unlock()
	return <-e.pool // all contexts busy: wait like a caller of EncodeAll waits for an encoder state
}

func (e *Encoder) release(c *C.kc_ctx) { e.pool <- c }

// useDevice: the routing rule of every entry point — enough bytes AND enough independent units to fill the chip.
func (e *Encoder) useDevice(total, units int) bool { return total >= e.minBytes && units >= e.minUnits }

// streamOnDevice: can ONE stream of a Writer be device work at all?  (Else the Writer streams through the reference encoder from
// its first byte instead of buffering.)
func (e *Encoder) streamOnDevice() bool {
	if e.minBytes > writerBufferCap {
		return false
	}
	if e.jobMode() {
		return e.devJobs
	}
	return e.minUnits <= 1
}

// trim destroys the idle device contexts (a Writer's Close: the stream is over; the next use creates them again).
func (e *Encoder) trim() {
	e.poolMu.Lock()
	defer e.poolMu.Unlock()
	for {
		select {
		case c := <-e.pool:
			C.kc_ctx_destroy(c)
			e.created--
		default:
			return
		}
	}
}

func WithEncoderLevel(l zstd.EncoderLevel) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithEncoderLevel(l))
		if C.kc_zstd_opts_level(&e.opts, C.int(l)) != 0 {
			return errors.New("unknown encoder level")
		}
		return nil
	}
}

func WithWindowSize(n int) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithWindowSize(n))
		if C.kc_zstd_opts_window(&e.opts, C.int(n)) != 0 {
			return errors.New("invalid window size")
		}
		return nil
	}
}

func WithEncoderCRC(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithEncoderCRC(b))
		C.kc_zstd_opts_crc(&e.opts, boolInt(b))
		return nil
	}
}

func WithZeroFrames(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithZeroFrames(b))
		C.kc_zstd_opts_zero_frames(&e.opts, boolInt(b))
		return nil
	}
}

func WithSingleSegment(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithSingleSegment(b))
		C.kc_zstd_opts_single_segment(&e.opts, boolInt(b))
		return nil
	}
}

// WithNoEntropyCompression mirrors zstd.WithNoEntropyCompression (zstd/encoder_options.go:297-313).
func WithNoEntropyCompression(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithNoEntropyCompression(b))
		C.kc_zstd_opts_no_entropy(&e.opts, boolInt(b))
		return nil
	}
}

// WithAllLitEntropyCompression mirrors zstd.WithAllLitEntropyCompression (zstd/encoder_options.go:285-295).
func WithAllLitEntropyCompression(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithAllLitEntropyCompression(b))
		C.kc_zstd_opts_all_lit_entropy(&e.opts, boolInt(b))
		return nil
	}
}

// WithLowerEncoderMem mirrors zstd.WithLowerEncoderMem (zstd/encoder_options.go:327-338): buffer sizing of the reference encoder
// only, no effect on the bytes and none on the device path.
func WithLowerEncoderMem(b bool) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithLowerEncoderMem(b))
		return nil
	}
}

// WithEncoderPadding mirrors zstd.WithEncoderPadding (zstd/encoder_options.go:142-158): every EncodeAll result and every stream is
// brought to a multiple of n bytes by a skippable frame of random bytes.  The padding is added on the host behind the frames of
// either path (the reference encoder held by this type does not get the option, its streaming Writer does), so a frame is the
// same bytes with and without it.
func WithEncoderPadding(n int) Option {
	return func(e *Encoder) error {
		if n <= 0 {
			return errors.New("padding must be at least 1")
		}
		if n > 1<<30 {
			return errors.New("padding must less than 1GB (1<<30 bytes) ")
		}
		if n == 1 {
			n = 0
		}
		e.pad = n
		return nil
	}
}

const skippableFrameHeader = 4 + 4

// calcSkippableFrame restates the unexported zstd/frameenc.go:100-116.
func calcSkippableFrame(written, wantMultiple int64) int {
	leftOver := written % wantMultiple
	if leftOver == 0 {
		return 0
	}
	toAdd := wantMultiple - leftOver
	for toAdd < skippableFrameHeader {
		toAdd += wantMultiple
	}
	return int(toAdd)
}

// appendPadding appends the skippable frame (zstd/frameenc.go:118-137) that brings len(dst) to a multiple of e.pad.
func (e *Encoder) appendPadding(dst []byte) []byte {
	if e.pad <= 0 {
		return dst
	}
	total := calcSkippableFrame(int64(len(dst)), int64(e.pad))
	if total == 0 {
		return dst
	}
	f := uint32(total - skippableFrameHeader)
	dst = append(dst, 0x50, 0x2a, 0x4d, 0x18, uint8(f), uint8(f>>8), uint8(f>>16), uint8(f>>24))
	start := len(dst)
	dst = append(dst, make([]byte, f)...)
	if _, err := io.ReadFull(rand.Reader, dst[start:]); err != nil {
		panic(err) // as the reference does (zstd/encoder.go:833-836)
	}
	return dst
}

// streamOpts: the options of a streaming reference Writer (the padding included: it is part of what Close writes).
func (e *Encoder) streamOpts() []zstd.EOption {
	if e.pad > 0 {
		return append(append([]zstd.EOption{}, e.cpuOpts...), zstd.WithEncoderPadding(e.pad))
	}
	return e.cpuOpts
}

// WithEncoderDict registers a dictionary in the "zstd --train" format (zstd.WithEncoderDict).
func WithEncoderDict(dict []byte) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithEncoderDict(dict))
		e.hasDict = true
		e.setDictMem(dict)
		if C.kc_zstd_opts_dict(&e.opts, (*C.uint8_t)(e.dictMem), C.uint64_t(len(dict))) != 0 {
			return errors.New("invalid dictionary")
		}
		return nil
	}
}

// WithEncoderDictRaw registers raw content as initial history (zstd.WithEncoderDictRaw).
func WithEncoderDictRaw(id uint32, content []byte) Option {
	return func(e *Encoder) error {
		e.cpuOpts = append(e.cpuOpts, zstd.WithEncoderDictRaw(id, content))
		e.hasDict = true
		e.setDictMem(content)
		if C.kc_zstd_opts_dict_raw(&e.opts, C.uint32_t(id), (*C.uint8_t)(e.dictMem), C.uint64_t(len(content))) != 0 {
			return errors.New("invalid dictionary")
		}
		return nil
	}
}

func (e *Encoder) setDictMem(b []byte) {
	if e.dictMem != nil {
		C.free(e.dictMem)
	}
	e.dictMem = C.CBytes(b) // the options struct keeps a pointer: it must not live in Go memory
}

func boolInt(b bool) C.int {
	if b {
		return 1
	}
	return 0
}

// New creates an encoder bound to GPU `device`. If no device is present the encoder still works
// through the reference implementation.
func New(device int, opts ...Option) (*Encoder, error) {
	e := &Encoder{device: device, conc: runtime.GOMAXPROCS(0), minBytes: DefaultDeviceMinBytes, minUnits: DefaultDeviceMinUnits}
	C.kc_zstd_opts_default(&e.opts)
	for _, o := range opts {
		if err := o(e); err != nil {
			return nil, err
		}
	}
	cpu, err := zstd.NewWriter(nil, e.cpuOpts...)
	if err != nil {
		return nil, err
	}
	e.cpu = cpu
	// o.concurrent: 1 selects the synchronous nextBlock form, which writes other first blocks for dictionary streams
	// (zstd/encoder.go:364-391); the default is GOMAXPROCS like the reference's (encoder_options.go:40)
	C.kc_zstd_opts_concurrency(&e.opts, C.int(e.conc))
	e.pool = make(chan *C.kc_ctx, e.conc)
	return e, nil
}

// Close releases the device contexts.  No call may be in flight.
func (e *Encoder) Close() {
	e.poolMu.Lock()
	for e.created > 0 {
		C.kc_ctx_destroy(<-e.pool)
		e.created--
	}
	e.poolMu.Unlock()
	if e.dictMem != nil {
		C.free(e.dictMem)
		e.dictMem = nil
	}
	e.cpu.Close()
}

// MaxEncodedSize == (*zstd.Encoder).MaxEncodedSize.
func (e *Encoder) MaxEncodedSize(size int) int {
	m := int(C.kc_zstd_max_encoded_size(&e.opts, C.int64_t(size)))
	if e.pad > 1 { // zstd/encoder.go:867-871
		m += calcSkippableFrame(int64(m), int64(e.pad))
	}
	return m
}

// EncodeAll == (*zstd.Encoder).EncodeAll for one unit (prefer EncodeUnits).
func (e *Encoder) EncodeAll(src, dst []byte) []byte {
	before := len(dst)
	if !e.useDevice(len(src), 1) {
		dst = e.cpu.EncodeAll(src, dst) // one unit: the reference encoder is faster (and concurrency-safe)
	} else if out, _, err := e.encodeUnits(src, []uint64{0, uint64(len(src))}, nil); err != nil {
		dst = e.cpu.EncodeAll(src, dst)
	} else {
		dst = append(dst, out...)
	}
	if len(dst) > before { // (an empty input without WithZeroFrames returns before the padding: zstd/encoder.go:732-752)
		dst = e.appendPadding(dst) // the TOTAL becomes a multiple, dst included (zstd/encoder.go:829-837)
	}
	return dst
}

// EncodeStreams encodes src[off[i]:off[i+1]] as independent STREAMS, each identical to what
// enc := zstd.NewWriter(w, opts...); enc.Write(unit); enc.Close() writes to w.
func (e *Encoder) EncodeStreams(src []byte, off []uint64, dst []byte) ([]byte, []uint64, error) {
	n := len(off) - 1
	need := 0
	for i := 0; i < n; i++ {
		need += (e.MaxEncodedSize(int(off[i+1]-off[i])) + 15) &^ 15
	}
	if cap(dst) < need+64 {
		dst = make([]byte, need+64)
	}
	dst = dst[:cap(dst)]
	outOff := make([]uint64, n+1)
	var ctx *C.kc_ctx
	if n > 0 && len(src) > 0 && e.useDevice(int(off[n]-off[0]), n) {
		ctx = e.acquire()
	}
	if ctx != nil {
		defer e.release(ctx)
		st := C.kc_zstd_encode_streams(ctx, &e.opts,
			(*C.uint8_t)(unsafe.Pointer(&src[0])), (*C.uint64_t)(unsafe.Pointer(&off[0])), C.uint32_t(n),
			(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
		if st == C.KC_OK {
			return dst[:outOff[n]], outOff, nil
		}
		if st != C.KC_ERR_UNSUPPORTED && st != C.KC_ERR_NO_DEVICE {
			return nil, nil, errors.New(C.GoString(C.kc_last_error(ctx)))
		}
	}
	// reference path: a fresh stream per unit (the streaming state of e.cpu is one at a time)
	e.refMu.Lock()
	defer e.refMu.Unlock()
	var sink bytes.Buffer
	for i := 0; i < n; i++ {
		outOff[i] = uint64(sink.Len())
		e.cpu.Reset(&sink)
		if _, err := e.cpu.Write(src[off[i]:off[i+1]]); err != nil {
			return nil, nil, err
		}
		if err := e.cpu.Close(); err != nil {
			return nil, nil, err
		}
	}
	outOff[n] = uint64(sink.Len())
	return sink.Bytes(), outOff, nil
}

// EncodeStreamsCuts is EncodeStreams with Flush points: flushAt[i] lists, ascending, how many bytes of stream i had been written
// when Flush was called (a Write followed by ReadFrom is such a point too).  The bytes equal what the reference writes for that
// Write / Flush / Close sequence; they are all delivered when the call returns (kc_zstd_encode_streams_cuts).
func (e *Encoder) EncodeStreamsCuts(src []byte, off []uint64, flushAt [][]uint64, dst []byte) ([]byte, []uint64, error) {
	n := len(off) - 1
	if len(flushAt) != n {
		return nil, nil, errors.New("zstdgpu: one flush list per stream")
	}
	cutOff := make([]uint64, n+1)
	cuts := make([]uint64, 0, 16)
	need := 0
	for i := 0; i < n; i++ {
		cuts = append(cuts, flushAt[i]...)
		cutOff[i+1] = uint64(len(cuts))
		need += (e.MaxEncodedSize(int(off[i+1]-off[i])) + 3*len(flushAt[i]) + 3 + 15) &^ 15
	}
	cuts = append(cuts, 0) // never empty: &cuts[0] below
	if cap(dst) < need+64 {
		dst = make([]byte, need+64)
	}
	dst = dst[:cap(dst)]
	outOff := make([]uint64, n+1)
	var ctx *C.kc_ctx
	if n > 0 && len(src) > 0 && e.useDevice(int(off[n]-off[0]), n) {
		ctx = e.acquire()
	}
	if ctx != nil {
		defer e.release(ctx)
		st := C.kc_zstd_encode_streams_cuts(ctx, &e.opts,
			(*C.uint8_t)(unsafe.Pointer(&src[0])), (*C.uint64_t)(unsafe.Pointer(&off[0])), C.uint32_t(n),
			(*C.uint64_t)(unsafe.Pointer(&cutOff[0])), (*C.uint64_t)(unsafe.Pointer(&cuts[0])),
			(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
		if st == C.KC_OK {
			return dst[:outOff[n]], outOff, nil
		}
		if st != C.KC_ERR_UNSUPPORTED && st != C.KC_ERR_NO_DEVICE {
			return nil, nil, errors.New(C.GoString(C.kc_last_error(ctx)))
		}
	}
	// reference path: a fresh stream per unit, Flush at the recorded positions
	e.refMu.Lock()
	defer e.refMu.Unlock()
	var sink bytes.Buffer
	for i := 0; i < n; i++ {
		outOff[i] = uint64(sink.Len())
		e.cpu.Reset(&sink)
		unit := src[off[i]:off[i+1]]
		pos := uint64(0)
		for _, c := range flushAt[i] {
			if c > uint64(len(unit)) {
				c = uint64(len(unit))
			}
			if c > pos {
				if _, err := e.cpu.Write(unit[pos:c]); err != nil {
					return nil, nil, err
				}
				pos = c
			}
			if err := e.cpu.Flush(); err != nil {
				return nil, nil, err
			}
		}
		if _, err := e.cpu.Write(unit[pos:]); err != nil {
			return nil, nil, err
		}
		if err := e.cpu.Close(); err != nil {
			return nil, nil, err
		}
	}
	outOff[n] = uint64(sink.Len())
	return sink.Bytes(), outOff, nil
}

// EncodeUnits encodes src[off[i]:off[i+1]] as independent frames, each identical to EncodeAll(unit, nil).
func (e *Encoder) EncodeUnits(src []byte, off []uint64, dst []byte) ([]byte, []uint64, error) {
	out, outOff, err := e.encodeUnits(src, off, dst)
	if err != nil || e.pad <= 0 {
		return out, outOff, err
	}
	// WithEncoderPadding: every frame followed by its skippable frame, as n x EncodeAll(unit, nil) gives them
	n := len(off) - 1
	padded := make([]byte, 0, len(out)+n*(e.pad+skippableFrameHeader))
	pOff := make([]uint64, n+1)
	for i := 0; i < n; i++ {
		pOff[i] = uint64(len(padded))
		fr := out[outOff[i]:outOff[i+1]]
		if len(fr) == 0 {
			continue
		}
		one := e.appendPadding(append([]byte{}, fr...))
		padded = append(padded, one...)
	}
	pOff[n] = uint64(len(padded))
	return padded, pOff, nil
}

// encodeUnits: the frames without padding.
func (e *Encoder) encodeUnits(src []byte, off []uint64, dst []byte) ([]byte, []uint64, error) {
	n := len(off) - 1
	need := 0
	for i := 0; i < n; i++ {
		need += (e.MaxEncodedSize(int(off[i+1]-off[i])) + 15) &^ 15
	}
	if cap(dst) < need+64 {
		dst = make([]byte, need+64)
	}
	dst = dst[:cap(dst)]
	outOff := make([]uint64, n+1)
	var ctx *C.kc_ctx
	if n > 0 && len(src) > 0 && e.useDevice(int(off[n]-off[0]), n) {
		ctx = e.acquire()
	}
	if ctx != nil {
		defer e.release(ctx)
		st := C.kc_zstd_encode_units(ctx, &e.opts,
			(*C.uint8_t)(unsafe.Pointer(&src[0])), (*C.uint64_t)(unsafe.Pointer(&off[0])), C.uint32_t(n),
			(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint64_t(len(dst)), (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
		if st == C.KC_OK {
			return dst[:outOff[n]], outOff, nil
		}
		if st != C.KC_ERR_UNSUPPORTED && st != C.KC_ERR_NO_DEVICE {
			return nil, nil, errors.New(C.GoString(C.kc_last_error(ctx)))
		}
	}
	// reference path
	out := dst[:0]
	for i := 0; i < n; i++ {
		outOff[i] = uint64(len(out))
		out = e.cpu.EncodeAll(src[off[i]:off[i+1]], out)
	}
	outOff[n] = uint64(len(out))
	return out, outOff, nil
}

// Writer is the io.WriteCloser face of the engine: the drop-in for
//
//	enc, _ := zstd.NewWriter(w, opts...); enc.Write(p) ...; enc.Close()
//
// (zstd/encoder.go:140-253, 567-649).  One stream is one device unit, parsed by one wave: the device only pays with many units
// per call, so a Writer goes to the device only when the caller lowers WithDeviceMinBytes below the stream's size (tests, or a
// future job mode); by default Write streams straight through a reference encoder with the same options.  On the device path
// Write buffers (up to 1 GiB, then the stream continues on the reference encoder), a Write followed by ReadFrom and every Flush
// are recorded as block cuts, and Close submits the stream through kc_zstd_encode_streams_cuts.  The bytes on w are the
// reference's either way.
type Writer struct {
	e      *Encoder
	w      io.Writer
	buf    []byte
	cuts   []uint64      // bytes written when a block was ended early (Flush; ReadFrom after Write)
	ref    *zstd.Encoder // set once the stream goes through the reference encoder
	closed bool
}

// NewWriter mirrors zstd.NewWriter(w, opts...) with this package's options.
func NewWriter(w io.Writer, device int, opts ...Option) (*Writer, error) {
	e, err := New(device, opts...)
	if err != nil {
		return nil, err
	}
	x := &Writer{e: e, w: w}
	if !e.streamOnDevice() {
		if err := x.fallback(); err != nil { // no stream of this Writer is device work: do not buffer it
			return nil, err
		}
	}
	return x, nil
}

// Reset discards the state and starts a new stream on w (encoder.go:103).
func (x *Writer) Reset(w io.Writer) {
	x.w, x.buf, x.cuts, x.closed = w, x.buf[:0], x.cuts[:0], false
	x.ref = nil
	if !x.e.streamOnDevice() {
		_ = x.fallback()
	}
}

// fallback moves the stream to a reference encoder, replaying what was buffered with its block cuts.
func (x *Writer) fallback() error {
	if x.ref == nil {
		r, err := zstd.NewWriter(x.w, x.e.streamOpts()...)
		if err != nil {
			return err
		}
		x.ref = r
		pos := uint64(0)
		for _, c := range x.cuts {
			if c > pos {
				if _, err := r.Write(x.buf[pos:c]); err != nil {
					return err
				}
				pos = c
			}
			if err := r.Flush(); err != nil {
				return err
			}
		}
		if uint64(len(x.buf)) > pos {
			if _, err := r.Write(x.buf[pos:]); err != nil {
				return err
			}
		}
		x.buf, x.cuts = x.buf[:0], x.cuts[:0]
	}
	return nil
}

func (x *Writer) Write(p []byte) (int, error) {
	if x.closed {
		return 0, zstd.ErrEncoderClosed
	}
	if x.ref == nil && len(x.buf)+len(p) > writerBufferCap {
		if err := x.fallback(); err != nil {
			return 0, err
		}
	}
	if x.ref != nil {
		return x.ref.Write(p)
	}
	x.buf = append(x.buf, p...)
	return len(p), nil
}

// ReadFrom mirrors (*zstd.Encoder).ReadFrom (encoder.go:444-480): it first ends the block being filled (:453-458), so bytes
// written before it and bytes read by it never share a block.
func (x *Writer) ReadFrom(r io.Reader) (int64, error) {
	if x.closed {
		return 0, zstd.ErrEncoderClosed
	}
	if x.ref != nil {
		return x.ref.ReadFrom(r)
	}
	if len(x.buf) > 0 {
		x.cuts = append(x.cuts, uint64(len(x.buf)))
	}
	var n int64
	chunk := make([]byte, 1<<20)
	for {
		k, err := r.Read(chunk)
		if k > 0 {
			if _, werr := x.Write(chunk[:k]); werr != nil {
				return n, werr
			}
			n += int64(k)
		}
		if err == io.EOF {
			return n, nil
		}
		if err != nil {
			return n, err
		}
	}
}

// Flush (encoder.go:547: "all buffered data is written out") ends the current block — in job mode, dispatches the job being
// filled — early.  The caller wants the bytes on w NOW (a peer may be waiting for them): the stream continues on the reference
// encoder, job mode included (streamOpts carries WithConcurrentBlocks; what was buffered is replayed there with its cuts, so the
// bytes are those of the same Write / Flush sequence).  A job stream stays on the device only while nobody flushes it; EncodeJobs
// takes Flush positions for callers that need the bytes of such a stream but not its latency.
func (x *Writer) Flush() error {
	if err := x.fallback(); err != nil {
		return err
	}
	return x.ref.Flush()
}

// Close finishes the stream (encoder.go:589).
func (x *Writer) Close() error {
	if x.closed {
		return nil
	}
	x.closed = true
	if x.ref == nil && !x.e.useDevice(len(x.buf), x.e.minUnits) { // (streamOnDevice has dealt with the unit rule)
		if err := x.fallback(); err != nil {
			return err
		}
	}
	if x.ref != nil {
		return x.ref.Close()
	}
	defer x.e.trim()
	var out []byte
	var err error
	if x.e.jobMode() {
		out, err = x.e.EncodeJobs(x.buf, x.cuts)
	} else {
		out, _, err = x.e.EncodeStreamsCuts(x.buf, []uint64{0, uint64(len(x.buf))}, [][]uint64{x.cuts}, nil)
	}
	if err != nil {
		return err
	}
	out = x.e.appendPadding(out) // Close pads what the stream wrote (zstd/encoder.go:637-645, 700-708)
	_, err = x.w.Write(out)
	return err
}
