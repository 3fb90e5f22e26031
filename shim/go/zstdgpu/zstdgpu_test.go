package zstdgpu

import (
	"bytes"
	"io"
	"sync"
	"crypto/sha256"
	"encoding/hex"
	"fmt"
	"os"
	"path/filepath"
	"sort"
	"testing"

	"github.com/klauspost/compress/kcgpu"
	"github.com/klauspost/compress/zstd"
)

var levels = []zstd.EncoderLevel{zstd.SpeedFastest, zstd.SpeedDefault, zstd.SpeedBetterCompression, zstd.SpeedBestCompression}

func cut(data []byte, unit int) []uint64 {
	var off []uint64
	for p := 0; p < len(data); p += unit {
		off = append(off, uint64(p))
	}
	return append(off, uint64(len(data)))
}

// checkUnits asserts gpu.EncodeUnits == N x ref.EncodeAll, frame by frame.
func checkUnits(t *testing.T, name string, gpu *Encoder, ref *zstd.Encoder, data []byte, off []uint64) {
	t.Helper()
	out, outOff, err := gpu.EncodeUnits(data, off, nil)
	if err != nil {
		t.Fatal(err)
	}
	for i := 0; i+1 < len(off); i++ {
		want := ref.EncodeAll(data[off[i]:off[i+1]], nil)
		if !bytes.Equal(out[outOff[i]:outOff[i+1]], want) {
			t.Fatalf("%s unit %d: GPU frame (%d B) differs from the reference's (%d B)", name, i, outOff[i+1]-outOff[i], len(want))
		}
	}
}

// TestBitExact: GPU frames == the reference encoder's frames (the parity criterion of this project) over the
// reference's shared fixtures cut into 128 KiB units and over the seeded corpora, at every level the device serves.
func TestBitExact(t *testing.T) {
	files, _ := filepath.Glob("../../../testdata/*")
	more, _ := filepath.Glob("../../../tests/golden/ref_inputs/*")
	files = append(files, more...)
	for _, lvl := range levels {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
		for _, f := range files {
			data, err := os.ReadFile(f)
			if err != nil || len(data) == 0 || filepath.Ext(f) == ".zip" {
				continue
			}
			checkUnits(t, fmt.Sprintf("%s level %v", f, lvl), gpu, ref, data, cut(data, 128<<10))
		}
		for _, kind := range []byte{'T', 'H', 'J', 'M'} {
			data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 96, 128<<10)
			if err != nil {
				t.Fatal(err)
			}
			checkUnits(t, fmt.Sprintf("corpus %c level %v", kind, lvl), gpu, ref, data, kcgpu.Offsets(96, 128<<10))
			// ragged units: sizes that are not multiples of anything, some above one block
			checkUnits(t, fmt.Sprintf("corpus %c ragged level %v", kind, lvl), gpu, ref, data[:3000000], cut(data[:3000000], 99991))
		}
		gpu.Close()
		ref.Close()
	}
}

// TestBitExactDict: WithEncoderDictRaw and WithEncoderDict (the reference's d0.dict fixture when present).
func TestBitExactDict(t *testing.T) {
	raw, err := kcgpu.CorpusFill('T', kcgpu.SeedD, 0, 1, 64<<10)
	if err != nil {
		t.Fatal(err)
	}
	data, _ := kcgpu.CorpusFill('M', kcgpu.SeedM, 0, 64, 128<<10)
	for _, lvl := range levels {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl), WithEncoderDictRaw(1, raw))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1), zstd.WithEncoderDictRaw(1, raw))
		checkUnits(t, fmt.Sprintf("raw dict level %v", lvl), gpu, ref, data, kcgpu.Offsets(64, 128<<10))
		checkUnits(t, fmt.Sprintf("raw dict small units level %v", lvl), gpu, ref, data[:400000], cut(data[:400000], 7001))
		gpu.Close()
		ref.Close()
	}
	full, err := os.ReadFile("../../../tests/golden/dict/d0.dict")
	if err != nil {
		t.Skip("d0.dict fixture not found")
	}
	for _, lvl := range levels {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl), WithEncoderDict(full))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1), zstd.WithEncoderDict(full))
		checkUnits(t, fmt.Sprintf("full dict level %v", lvl), gpu, ref, data[:1<<20], cut(data[:1<<20], 20011))
		gpu.Close()
		ref.Close()
	}
}

// TestBitExactStreams: EncodeStreams == NewWriter(w).Write(unit); Close() per unit.
func TestBitExactStreams(t *testing.T) {
	data, err := kcgpu.CorpusFill('T', kcgpu.SeedT, 0, 32, 128<<10)
	if err != nil {
		t.Fatal(err)
	}
	for _, lvl := range levels {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
		off := cut(data, 300001)
		out, outOff, err := gpu.EncodeStreams(data, off, nil)
		if err != nil {
			t.Fatal(err)
		}
		for i := 0; i+1 < len(off); i++ {
			var sink bytes.Buffer
			ref.Reset(&sink)
			ref.Write(data[off[i]:off[i+1]])
			ref.Close()
			if !bytes.Equal(out[outOff[i]:outOff[i+1]], sink.Bytes()) {
				t.Fatalf("stream %d level %v differs from the reference's Write+Close", i, lvl)
			}
		}
		gpu.Close()
	}
}

// TestBitExactDictStreams: streams of an encoder with a dictionary, in both of nextBlock's forms: WithEncoderConcurrency(1) (the
// synchronous form resets the block before the first Encode, so the dictionary's literal table is never used) and
// WithEncoderConcurrency(4) (the first block starts from it).  The device path follows the option.
func TestBitExactDictStreams(t *testing.T) {
	data, err := kcgpu.CorpusFill('T', kcgpu.SeedT, 0, 16, 128<<10)
	if err != nil {
		t.Fatal(err)
	}
	raw, _ := kcgpu.CorpusFill('T', kcgpu.SeedD, 0, 1, 64<<10)
	full, ferr := os.ReadFile("../../../tests/golden/dict/d0.dict")
	for _, lvl := range levels {
		for _, conc := range []int{1, 4} {
			for which := 0; which < 2; which++ {
				var gopt Option
				var ropt zstd.EOption
				if which == 0 {
					gopt, ropt = WithEncoderDictRaw(9, raw), zstd.WithEncoderDictRaw(9, raw)
				} else {
					if ferr != nil {
						continue
					}
					gopt, ropt = WithEncoderDict(full), zstd.WithEncoderDict(full)
				}
				gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl), WithEncoderConcurrency(conc), gopt)
				if err != nil {
					t.Fatal(err)
				}
				ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(conc), ropt)
				off := cut(data, 250003)
				flushAt := make([][]uint64, len(off)-1)
				for i := range flushAt {
					if i%2 == 1 {
						flushAt[i] = []uint64{uint64(100 * (i + 1)), 70000}
					}
				}
				out, outOff, err := gpu.EncodeStreamsCuts(data, off, flushAt, nil)
				if err != nil {
					t.Fatal(err)
				}
				for i := 0; i+1 < len(off); i++ {
					var sink bytes.Buffer
					ref.Reset(&sink)
					unit := data[off[i]:off[i+1]]
					pos := uint64(0)
					for _, c := range flushAt[i] {
						if c > uint64(len(unit)) {
							c = uint64(len(unit))
						}
						if c > pos {
							ref.Write(unit[pos:c])
							pos = c
						}
						ref.Flush()
					}
					ref.Write(unit[pos:])
					ref.Close()
					if !bytes.Equal(out[outOff[i]:outOff[i+1]], sink.Bytes()) {
						t.Fatalf("dictionary stream %d level %v concurrency %d dict %d differs from the reference's", i, lvl, conc, which)
					}
				}
				gpu.Close()
				ref.Close()
			}
		}
	}
}

// TestBitExactStreamsFlush: EncodeStreamsCuts == NewWriter(w); Write ...; Flush at the given positions; Close() per unit —
// flushes inside the first block (the header is written early: no EncodeAll frame), on block boundaries (no-ops), at the very
// end (empty last block), repeated, and streams of more than 32 blocks.
func TestBitExactStreamsFlush(t *testing.T) {
	data, err := kcgpu.CorpusFill('T', kcgpu.SeedT, 0, 80, 128<<10)
	if err != nil {
		t.Fatal(err)
	}
	for _, lvl := range levels {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(lvl))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
		bs := uint64(128 << 10)
		if lvl == zstd.SpeedFastest {
			bs = 64 << 10
		}
		lens := []uint64{1000, 1000, 1000, bs, 2 * bs, 2*bs + 9, 3 * bs, 4*bs + 1, 40*bs + 123} // 52 blocks + 3 KiB of the 80 x 128 KiB corpus
		flushAt := [][]uint64{{10, 500}, {1000}, {0}, {bs}, {bs}, {bs - 1, bs, bs + 1}, {7, 7, 7, 2*bs + 7}, {3*bs + 50000, 4*bs + 1, 4*bs + 9}, {bs / 2, 20*bs + 7}}
		off := []uint64{0}
		for _, l := range lens {
			off = append(off, off[len(off)-1]+l)
		}
		out, outOff, err := gpu.EncodeStreamsCuts(data[:off[len(off)-1]], off, flushAt, nil)
		if err != nil {
			t.Fatal(err)
		}
		for i := range lens {
			var sink bytes.Buffer
			ref.Reset(&sink)
			unit := data[off[i]:off[i+1]]
			pos := uint64(0)
			for _, c := range flushAt[i] {
				if c > uint64(len(unit)) {
					c = uint64(len(unit))
				}
				if c > pos {
					ref.Write(unit[pos:c])
					pos = c
				}
				ref.Flush()
			}
			ref.Write(unit[pos:])
			ref.Close()
			if !bytes.Equal(out[outOff[i]:outOff[i+1]], sink.Bytes()) {
				t.Fatalf("stream %d (len %d, flush at %v) level %v differs from the reference's Write/Flush/Close", i, lens[i], flushAt[i], lvl)
			}
		}
		gpu.Close()
	}
}

// TestWriterDropIn: zstdgpu.Writer writes what zstd.NewWriter(w).Write(p...)+Close() writes, for plain streams, for streams
// written in pieces, and for streams with a mid-stream Flush (which continue on the reference encoder).
func TestWriterDropIn(t *testing.T) {
	data, err := kcgpu.CorpusFill('T', kcgpu.SeedT, 0, 24, 128<<10)
	if err != nil {
		t.Fatal(err)
	}
	for _, lvl := range levels {
		for _, flushAt := range []int{-1, 300000} {
			var want, got bytes.Buffer
			ref, _ := zstd.NewWriter(&want, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
			gw, err := NewWriter(&got, 0, WithDeviceMinBytes(0), WithEncoderLevel(lvl))
			if err != nil {
				t.Fatal(err)
			}
			for p := 0; p < len(data); p += 70001 {
				end := p + 70001
				if end > len(data) {
					end = len(data)
				}
				ref.Write(data[p:end])
				gw.Write(data[p:end])
				if flushAt >= 0 && p <= flushAt && flushAt < end {
					ref.Flush()
					gw.Flush()
				}
			}
			ref.Close()
			if err := gw.Close(); err != nil {
				t.Fatal(err)
			}
			if !bytes.Equal(want.Bytes(), got.Bytes()) {
				t.Fatalf("level %v flushAt %d: Writer output differs from the reference's", lvl, flushAt)
			}
		}
	}
}

// TestWriteGolden pins the whole-encoder bytes of the REFERENCE (not of the GPU path) on the seeded corpora:
// one line "<name> <sha256 of the concatenated frames>" per corpus and level into tests/golden/reference_sha256.txt.
// tests/test_reference_golden.py then gates the C++ oracle and the HIP path on these hashes — the step that turns
// "parity unpinned" into a measurement (SURVEY.md §8c layer 4).  Set KC_WRITE_GOLDEN=1 to (re)write the file.
func TestWriteGolden(t *testing.T) {
	if os.Getenv("KC_WRITE_GOLDEN") == "" {
		t.Skip("set KC_WRITE_GOLDEN=1 to write tests/golden/reference_sha256.txt")
	}
	lines := map[string]string{}
	path := "../../../tests/golden/reference_sha256.txt"
	if old, err := os.ReadFile(path); err == nil { // keep the lines other packages wrote (s2gpu)
		for _, l := range bytes.Split(old, []byte("\n")) {
			f := bytes.Fields(l)
			if len(f) == 2 {
				lines[string(f[0])] = string(f[1])
			}
		}
	}
	dict, _ := kcgpu.CorpusFill('T', kcgpu.SeedD, 0, 1, 64<<10)
	for _, lvl := range levels {
		for _, kind := range []byte{'T', 'H', 'J', 'M'} {
			data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 96, 128<<10)
			if err != nil {
				t.Fatal(err)
			}
			for _, withDict := range []bool{false, true} {
				opts := []zstd.EOption{zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1)}
				name := fmt.Sprintf("zstd.L%d.%c.96x131072", int(lvl), kind)
				if withDict {
					opts = append(opts, zstd.WithEncoderDictRaw(1, dict))
					name += ".rawdict64k"
				}
				ref, _ := zstd.NewWriter(nil, opts...)
				h := sha256.New()
				for i := 0; i < 96; i++ {
					h.Write(ref.EncodeAll(data[i*(128<<10):(i+1)*(128<<10)], nil))
				}
				ref.Close()
				lines[name] = hex.EncodeToString(h.Sum(nil))
			}
		}
	}
	// streams with Flush points: 10 streams of 300001 bytes, Flush after 70000, 70010 and 200000+i bytes
	for _, lvl := range levels {
		for _, kind := range []byte{'T', 'M'} {
			data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 24, 128<<10)
			if err != nil {
				t.Fatal(err)
			}
			ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
			h := sha256.New()
			for i := 0; i < 10; i++ {
				var sink bytes.Buffer
				ref.Reset(&sink)
				u := data[i*300001 : (i+1)*300001]
				ref.Write(u[:70000])
				ref.Flush()
				ref.Write(u[70000:70010])
				ref.Flush()
				ref.Write(u[70010 : 200000+i])
				ref.Flush()
				ref.Write(u[200000+i:])
				ref.Close()
				h.Write(sink.Bytes())
			}
			ref.Close()
			lines[fmt.Sprintf("zstdstream.L%d.%c.10x300001.flush", int(lvl), kind)] = hex.EncodeToString(h.Sum(nil))
		}
	}
	writeGolden(t, path, lines)
}

func writeGolden(t *testing.T, path string, lines map[string]string) {
	names := make([]string, 0, len(lines))
	for n := range lines {
		names = append(names, n)
	}
	sort.Strings(names)
	var b bytes.Buffer
	for _, n := range names {
		fmt.Fprintf(&b, "%s %s\n", n, lines[n])
	}
	if err := os.WriteFile(path, b.Bytes(), 0o644); err != nil {
		t.Fatal(err)
	}
}


// TestConcurrentEncodeAll: (*zstd.Encoder).EncodeAll is safe for concurrent use on ONE encoder (zstd/encoder.go:90-99, 722-729);
// so is the drop-in: 32 goroutines share one Encoder (4 device contexts in its pool) and every frame equals the reference's.
// Run with -race.
func TestConcurrentEncodeAll(t *testing.T) {
	data := corpusT(8 << 20)
	gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderLevel(zstd.SpeedFastest), WithEncoderConcurrency(4))
	if err != nil {
		t.Fatal(err)
	}
	defer gpu.Close()
	ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(zstd.SpeedFastest))
	var wg sync.WaitGroup
	errs := make(chan string, 64)
	for g := 0; g < 32; g++ {
		wg.Add(1)
		go func(g int) {
			defer wg.Done()
			for k := 0; k < 4; k++ {
				lo := ((g*4 + k) * 37 << 10) % (len(data) - (256 << 10))
				unit := data[lo : lo+(64<<10)+g*1000]
				if g%2 == 0 { // a batch through EncodeUnits
					off := cut(unit, 32<<10)
					out, oo, err := gpu.EncodeUnits(unit, off, nil)
					if err != nil {
						errs <- err.Error()
						return
					}
					for i := 0; i+1 < len(off); i++ {
						if !bytes.Equal(out[oo[i]:oo[i+1]], ref.EncodeAll(unit[off[i]:off[i+1]], nil)) {
							errs <- fmt.Sprintf("goroutine %d: unit %d differs", g, i)
							return
						}
					}
				} else if !bytes.Equal(gpu.EncodeAll(unit, nil), ref.EncodeAll(unit, nil)) {
					errs <- fmt.Sprintf("goroutine %d: EncodeAll differs", g)
					return
				}
			}
		}(g)
	}
	wg.Wait()
	close(errs)
	for e := range errs {
		t.Error(e)
	}
}

// TestRouting: below WithDeviceMinBytes the call never reaches the device (the reference encoder serves it, same bytes).
func TestRouting(t *testing.T) {
	data := corpusT(1 << 20)
	gpu, err := New(0, WithEncoderLevel(zstd.SpeedDefault)) // default threshold: 512 MiB
	if err != nil {
		t.Fatal(err)
	}
	defer gpu.Close()
	ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(zstd.SpeedDefault))
	if !bytes.Equal(gpu.EncodeAll(data, nil), ref.EncodeAll(data, nil)) {
		t.Fatal("EncodeAll below the threshold differs from the reference")
	}
	if gpu.created != 0 {
		t.Fatalf("a call below the threshold created %d device context(s)", gpu.created)
	}
}

// TestRoutingUnits: a call must also carry enough independent units (a unit is one wave's work: one stream of any size is not
// device work), and a Writer whose stream cannot be device work does not buffer it.
func TestRoutingUnits(t *testing.T) {
	data := corpusT(1 << 20)
	off := []uint64{0, 1 << 18, 1 << 19, 3 << 18, 1 << 20}
	ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(zstd.SpeedFastest))
	var want []byte
	for i := 0; i+1 < len(off); i++ {
		want = ref.EncodeAll(data[off[i]:off[i+1]], want)
	}
	few, err := New(0, WithDeviceMinBytes(1), WithEncoderLevel(zstd.SpeedFastest)) // bytes rule passed, DefaultDeviceMinUnits not
	if err != nil {
		t.Fatal(err)
	}
	defer few.Close()
	out, _, err := few.EncodeUnits(data, off, nil)
	if err != nil || !bytes.Equal(out, want) {
		t.Fatalf("4 units on the reference path: err %v, equal %v", err, bytes.Equal(out, want))
	}
	if few.created != 0 {
		t.Fatalf("4 units created %d device context(s)", few.created)
	}
	many, err := New(0, WithDeviceMinBytes(1), WithDeviceMinUnits(4), WithEncoderLevel(zstd.SpeedFastest))
	if err != nil {
		t.Fatal(err)
	}
	defer many.Close()
	out, _, err = many.EncodeUnits(data, off, nil)
	if err != nil || !bytes.Equal(out, want) {
		t.Fatalf("4 units past WithDeviceMinUnits(4): err %v, equal %v", err, bytes.Equal(out, want))
	}
	if many.created == 0 && !many.noDevice {
		t.Fatal("4 units past WithDeviceMinUnits(4) did not reach the device")
	}
	var sink bytes.Buffer
	w, err := NewWriter(&sink, 0, WithEncoderLevel(zstd.SpeedFastest)) // defaults: one stream is one unit
	if err != nil {
		t.Fatal(err)
	}
	if w.ref == nil {
		t.Fatal("a default Writer buffers its stream for the device")
	}
	w.Write(data)
	w.Close()
	var wantS bytes.Buffer
	rw, _ := zstd.NewWriter(&wantS, zstd.WithEncoderLevel(zstd.SpeedFastest))
	rw.Write(data)
	rw.Close()
	if !bytes.Equal(sink.Bytes(), wantS.Bytes()) {
		t.Fatal("default Writer differs from the reference Writer")
	}
}

// TestWriterWriteThenReadFrom: ReadFrom first ends the block being filled (zstd/encoder.go:453-458), so Write(p) followed by
// ReadFrom(r) cuts a block at len(p) — on the device path too (the cut travels to kc_zstd_encode_streams_cuts).
func TestWriterWriteThenReadFrom(t *testing.T) {
	data := corpusT(400 << 10)
	for _, lvl := range levels {
		for _, first := range []int{1, 1000, 70000, 128 << 10, 200000} {
			var want, got bytes.Buffer
			ref, _ := zstd.NewWriter(&want, zstd.WithEncoderLevel(lvl), zstd.WithEncoderConcurrency(1))
			ref.Write(data[:first])
			ref.ReadFrom(bytes.NewReader(data[first:]))
			ref.Close()
			gw, err := NewWriter(&got, 0, WithDeviceMinBytes(0), WithEncoderLevel(lvl))
			if err != nil {
				t.Fatal(err)
			}
			gw.Write(data[:first])
			if _, err := gw.ReadFrom(io.LimitReader(bytes.NewReader(data[first:]), int64(len(data)))); err != nil {
				t.Fatal(err)
			}
			if err := gw.Close(); err != nil {
				t.Fatal(err)
			}
			if !bytes.Equal(got.Bytes(), want.Bytes()) {
				t.Fatalf("level %v, Write(%d) + ReadFrom: %d bytes vs the reference's %d", lvl, first, got.Len(), want.Len())
			}
		}
	}
}

// corpusT returns n bytes of the seeded text corpus.
func corpusT(n int) []byte {
	units := (n + (128 << 10) - 1) / (128 << 10)
	data, err := kcgpu.CorpusFill('T', kcgpu.SeedT, 500, units, 128<<10)
	if err != nil {
		panic(err)
	}
	return data[:n]
}


// TestConcurrentBlocks: the job mode (zstd.WithConcurrentBlocks) through the device == the reference's, for plain streams, with
// Flush points and through the Writer face.
func TestConcurrentBlocks(t *testing.T) {
	data := corpusT(48 << 20)
	for _, lvl := range levels {
		for _, win := range []int{1 << 17, 1 << 20, 0} {
			ropts := []zstd.EOption{zstd.WithEncoderLevel(lvl), zstd.WithConcurrentBlocks(true), zstd.WithEncoderConcurrency(4)}
			gopts := []Option{WithDeviceMinBytes(0), WithDeviceJobs(true), WithEncoderLevel(lvl), WithConcurrentBlocks(true), WithEncoderConcurrency(4)}
			if win > 0 {
				ropts = append(ropts, zstd.WithWindowSize(win))
				gopts = append(gopts, WithWindowSize(win))
			}
			for _, cuts := range [][]uint64{nil, {1000, 3 << 20}, {uint64(len(data))}} {
				var want bytes.Buffer
				ref, _ := zstd.NewWriter(&want, ropts...)
				pos := uint64(0)
				for _, c := range cuts {
					ref.Write(data[pos:c])
					ref.Flush()
					pos = c
				}
				ref.Write(data[pos:])
				ref.Close()
				gpu, err := New(0, gopts...)
				if err != nil {
					t.Fatal(err)
				}
				got, err := gpu.EncodeJobs(data, cuts)
				gpu.Close()
				if err != nil {
					t.Fatal(err)
				}
				if !bytes.Equal(got, want.Bytes()) {
					t.Fatalf("level %v window %d cuts %v: %d bytes vs the reference's %d", lvl, win, cuts, len(got), want.Len())
				}
			}
		}
	}
}

// TestEntropyOptionsAndPadding: WithNoEntropyCompression / WithAllLitEntropyCompression through the device == the reference's bytes;
// WithEncoderPadding: the padding is random by design (crypto/rand), so what is compared with the reference is everything else —
// the frame in front of it, the total length, the skippable frame's header — for EncodeAll (with bytes already in dst), the batch
// form and a Writer.
func TestEntropyOptionsAndPadding(t *testing.T) {
	data := corpusT(5 << 20)
	off := cut(data, 128<<10)
	for _, c := range []struct {
		name string
		gpu  []Option
		ref  []zstd.EOption
	}{
		{"no entropy", []Option{WithNoEntropyCompression(true)}, []zstd.EOption{zstd.WithNoEntropyCompression(true)}},
		{"all literals entropy off", []Option{WithAllLitEntropyCompression(false)}, []zstd.EOption{zstd.WithAllLitEntropyCompression(false)}},
		{"all literals entropy on, fastest", []Option{WithEncoderLevel(zstd.SpeedFastest), WithAllLitEntropyCompression(true)},
			[]zstd.EOption{zstd.WithEncoderLevel(zstd.SpeedFastest), zstd.WithAllLitEntropyCompression(true)}},
		{"lower mem", []Option{WithLowerEncoderMem(true)}, []zstd.EOption{zstd.WithLowerEncoderMem(true)}},
	} {
		gpu, err := New(0, append([]Option{WithDeviceMinBytes(0)}, c.gpu...)...)
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, append([]zstd.EOption{zstd.WithEncoderConcurrency(1)}, c.ref...)...)
		checkUnits(t, c.name, gpu, ref, data, off)
		gpu.Close()
		ref.Close()
	}
	for _, pad := range []int{2, 13, 512, 4096} {
		gpu, err := New(0, WithDeviceMinBytes(0), WithEncoderPadding(pad))
		if err != nil {
			t.Fatal(err)
		}
		plain, _ := zstd.NewWriter(nil, zstd.WithEncoderConcurrency(1))
		padded, _ := zstd.NewWriter(nil, zstd.WithEncoderConcurrency(1), zstd.WithEncoderPadding(pad))
		for _, n := range []int{1, 1000, 128 << 10, 300000} {
			src := data[:n]
			for _, pre := range [][]byte{nil, []byte("seven b")} {
				got := gpu.EncodeAll(src, append([]byte{}, pre...))
				want := padded.EncodeAll(src, append([]byte{}, pre...))
				frame := plain.EncodeAll(src, nil)
				if len(got) != len(want) || len(got)%pad != 0 {
					t.Fatalf("pad %d, %d bytes: length %d, the reference's %d", pad, n, len(got), len(want))
				}
				if !bytes.HasPrefix(got[len(pre):], frame) || !bytes.Equal(got[:len(pre)], pre) {
					t.Fatalf("pad %d, %d bytes: the frame in front of the padding is not the reference's", pad, n)
				}
				k := len(pre) + len(frame)
				if k < len(got) && !bytes.Equal(got[k:k+8], want[k:k+8]) {
					t.Fatalf("pad %d, %d bytes: skippable frame header % x, the reference's % x", pad, n, got[k:k+8], want[k:k+8])
				}
			}
		}
		out, outOff, err := gpu.EncodeUnits(data, off, nil)
		if err != nil {
			t.Fatal(err)
		}
		for i := 0; i+1 < len(off); i++ {
			want := padded.EncodeAll(data[off[i]:off[i+1]], nil)
			got := out[outOff[i]:outOff[i+1]]
			frame := plain.EncodeAll(data[off[i]:off[i+1]], nil)
			if len(got) != len(want) || !bytes.HasPrefix(got, frame) {
				t.Fatalf("pad %d, unit %d of the batch: %d bytes, the reference's %d", pad, i, len(got), len(want))
			}
		}
		var a, b bytes.Buffer
		w, err := NewWriter(&a, 0, WithDeviceMinBytes(0), WithEncoderPadding(pad))
		if err != nil {
			t.Fatal(err)
		}
		r, _ := zstd.NewWriter(&b, zstd.WithEncoderPadding(pad))
		w.Write(data[:700000])
		r.Write(data[:700000])
		w.Close()
		r.Close()
		if a.Len() != b.Len() || a.Len()%pad != 0 {
			t.Fatalf("pad %d, Writer: %d bytes, the reference's %d", pad, a.Len(), b.Len())
		}
		dec, _ := zstd.NewReader(nil)
		back, err := dec.DecodeAll(a.Bytes(), nil)
		if err != nil || !bytes.Equal(back, data[:700000]) {
			t.Fatalf("pad %d, Writer: the padded stream does not decode to the input (%v)", pad, err)
		}
		dec.Close()
		gpu.Close()
		plain.Close()
		padded.Close()
	}
}
