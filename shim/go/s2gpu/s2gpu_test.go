package s2gpu

// The parity target is whatever this build of the reference runs: on amd64 its assembly block encoders (the device's
// KC_S2_VARIANT_AMD64), with -tags noasm or elsewhere the portable Go ones (encodeBlockGo / encodeBlockGo64K); the best levels are
// pure Go in every build.

import (
	"bytes"
	"crypto/sha256"
	"encoding/hex"
	"fmt"
	"os"
	"sort"
	"testing"

	"github.com/klauspost/compress/kcgpu"
	"github.com/klauspost/compress/s2"
)

// TestBitExact: EncodeBlocks == N x s2.Encode(nil, block).
func TestBitExact(t *testing.T) {
	x, err := NewCtx(0)
	if err != nil {
		t.Skip(err)
	}
	defer x.Close()
	for _, kind := range []byte{'J', 'T', 'M', 'H'} {
		data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 128, 64<<10)
		if err != nil {
			t.Fatal(err)
		}
		off := kcgpu.Offsets(128, 64<<10)
		dst := make([]byte, 128*(s2.MaxEncodedLen(64<<10)+16)+64)
		out, outOff, err := EncodeBlocks(x, data, off, dst)
		if err != nil {
			t.Fatal(err)
		}
		for i := 0; i < 128; i++ {
			want := s2.Encode(nil, data[off[i]:off[i+1]])
			if !bytes.Equal(out[outOff[i]:outOff[i+1]], want) {
				t.Fatalf("corpus %c block %d: GPU block differs from s2.Encode", kind, i)
			}
		}
	}
}

// TestBitExactLevels: the better and Snappy-compatible levels against s2.EncodeBetter / s2.EncodeSnappy / s2.EncodeSnappyBetter.
func TestBitExactLevels(t *testing.T) {
	x, err := NewCtx(0)
	if err != nil {
		t.Skip(err)
	}
	defer x.Close()
	for _, kind := range []byte{'J', 'T', 'M'} {
		data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 64, 64<<10)
		if err != nil {
			t.Fatal(err)
		}
		for _, unit := range []int{64 << 10, 1 << 20} { // both table variants of every level
			var off []uint64
			for p := 0; p < len(data); p += unit {
				off = append(off, uint64(p))
			}
			off = append(off, uint64(len(data)))
			dst := make([]byte, len(off)*(s2.MaxEncodedLen(unit)+16)+64)
			for _, lv := range []int{LevelBetter, LevelSnappy, LevelSnappyBetter, LevelBest, LevelSnappyBest} {
				out, outOff, err := EncodeBlocksLevel(x, lv, data, off, dst)
				if err != nil {
					t.Fatal(err)
				}
				for i := 0; i+1 < len(off); i++ {
					var want []byte
					switch lv {
					case LevelBetter:
						want = s2.EncodeBetter(nil, data[off[i]:off[i+1]])
					case LevelSnappy:
						want = s2.EncodeSnappy(nil, data[off[i]:off[i+1]])
					case LevelBest:
						want = s2.EncodeBest(nil, data[off[i]:off[i+1]])
					case LevelSnappyBest:
						want = s2.EncodeSnappyBest(nil, data[off[i]:off[i+1]])
					default:
						want = s2.EncodeSnappyBetter(nil, data[off[i]:off[i+1]])
					}
					if !bytes.Equal(out[outOff[i]:outOff[i+1]], want) {
						t.Fatalf("corpus %c unit %d level %d block %d differs from the reference", kind, unit, lv, i)
					}
				}
			}
		}
	}
}

// TestBestLevelsReachTheDeviceInEveryVariant: the best encoders are pure Go in the reference (one form on every platform), so a
// context in the amd64 variant — NewCtx's default on amd64 builds — must serve them on the device, not fall back to the CPU.
func TestBestLevelsReachTheDeviceInEveryVariant(t *testing.T) {
	x, err := NewCtx(0)
	if err != nil {
		t.Skip(err)
	}
	defer x.Close()
	data, err := kcgpu.CorpusFill('J', kcgpu.Seed('J'), 0, 32, 64<<10)
	if err != nil {
		t.Fatal(err)
	}
	var off []uint64
	for p := 0; p <= len(data); p += 64 << 10 {
		off = append(off, uint64(p))
	}
	for _, variant := range []int{VariantGo, VariantAMD64} {
		x.SetVariant(variant)
		for _, lv := range []int{LevelBest, LevelSnappyBest} {
			out, outOff, err := EncodeBlocksLevel(x, lv, data, off, nil)
			if err != nil {
				t.Fatal(err)
			}
			if x.LastBatches() < 1 {
				t.Fatalf("variant %d level %d: the call did not reach the device", variant, lv)
			}
			for i := 0; i+1 < len(off); i++ {
				want := s2.EncodeBest(nil, data[off[i]:off[i+1]])
				if lv == LevelSnappyBest {
					want = s2.EncodeSnappyBest(nil, data[off[i]:off[i+1]])
				}
				if !bytes.Equal(out[outOff[i]:outOff[i+1]], want) {
					t.Fatalf("variant %d level %d block %d differs from the reference", variant, lv, i)
				}
			}
		}
	}
}

// TestCustomEncoderWriter: an s2.Writer that encodes its blocks through the hook writes the same stream as the
// built-in encoder, with the writer calling the hook from WriterConcurrency goroutines at once (s2/writer.go:455-460).
func TestCustomEncoderWriter(t *testing.T) {
	x, err := NewCtx(0)
	if err != nil {
		t.Skip(err)
	}
	defer x.Close()
	data, err := kcgpu.CorpusFill('J', kcgpu.SeedJ, 0, 512, 64<<10)
	if err != nil {
		t.Fatal(err)
	}
	var want, got bytes.Buffer
	w := s2.NewWriter(&want, s2.WriterBlockSize(64<<10), s2.WriterConcurrency(16))
	w.Write(data)
	w.Close()
	g := s2.NewWriter(&got, s2.WriterBlockSize(64<<10), s2.WriterConcurrency(16), s2.WriterCustomEncoder(CustomEncoder(x)))
	g.Write(data)
	g.Close()
	if !bytes.Equal(want.Bytes(), got.Bytes()) {
		t.Fatalf("stream through the GPU hook (%d B) differs from the built-in encoder's (%d B)", got.Len(), want.Len())
	}
}

// TestWriteGolden: sha256 of the REFERENCE's s2.Encode output on the seeded corpora, appended to
// tests/golden/reference_sha256.txt (see zstdgpu.TestWriteGolden).
func TestWriteGolden(t *testing.T) {
	if os.Getenv("KC_WRITE_GOLDEN") == "" {
		t.Skip("set KC_WRITE_GOLDEN=1 to write tests/golden/reference_sha256.txt")
	}
	path := "../../../tests/golden/reference_sha256.txt"
	lines := map[string]string{}
	if old, err := os.ReadFile(path); err == nil {
		for _, l := range bytes.Split(old, []byte("\n")) {
			f := bytes.Fields(l)
			if len(f) == 2 {
				lines[string(f[0])] = string(f[1])
			}
		}
	}
	for _, kind := range []byte{'J', 'T', 'M', 'H'} {
		data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 128, 64<<10)
		if err != nil {
			t.Fatal(err)
		}
		h := sha256.New()
		for i := 0; i < 128; i++ {
			h.Write(s2.Encode(nil, data[i*(64<<10):(i+1)*(64<<10)]))
		}
		lines[fmt.Sprintf("s2.%c.128x65536", kind)] = hex.EncodeToString(h.Sum(nil))
		hb, hs, hsb := sha256.New(), sha256.New(), sha256.New()
		for i := 0; i < 128; i++ {
			hb.Write(s2.EncodeBetter(nil, data[i*(64<<10):(i+1)*(64<<10)]))
			hs.Write(s2.EncodeSnappy(nil, data[i*(64<<10):(i+1)*(64<<10)]))
			hsb.Write(s2.EncodeSnappyBetter(nil, data[i*(64<<10):(i+1)*(64<<10)]))
		}
		lines[fmt.Sprintf("s2better.%c.128x65536", kind)] = hex.EncodeToString(hb.Sum(nil))
		lines[fmt.Sprintf("s2snappy.%c.128x65536", kind)] = hex.EncodeToString(hs.Sum(nil))
		lines[fmt.Sprintf("s2snappybetter.%c.128x65536", kind)] = hex.EncodeToString(hsb.Sum(nil))
	}
	// framed streams: s2.NewWriter(w, WriterBlockSize(64K), <level>, WriterAddIndex(), WriterPadding(4096) with a zero padding source),
	// Write + Flush after 100000 and 100001 bytes, Close — the lines tests/golden/make_reference_go_golden.py writes from the translated
	// Writer (build with -tags noasm: the chunk bodies there are the portable Go encoders')
	for _, st := range []struct {
		name string
		opts []s2.WriterOption
	}{{"s2", nil}, {"s2better", []s2.WriterOption{s2.WriterBetterCompression()}}, {"s2best", []s2.WriterOption{s2.WriterBestCompression()}},
		{"s2snappy", []s2.WriterOption{s2.WriterSnappyCompat()}}} {
		for _, kind := range []byte{'J', 'T'} {
			data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, 16, 64<<10)
			if err != nil {
				t.Fatal(err)
			}
			var sink bytes.Buffer
			opts := append([]s2.WriterOption{s2.WriterBlockSize(64 << 10), s2.WriterAddIndex(), s2.WriterPadding(4096), s2.WriterPaddingSrc(zeroReader{})}, st.opts...)
			w := s2.NewWriter(&sink, opts...)
			w.Write(data[:100000])
			w.Flush()
			w.Write(data[100000:100001])
			w.Flush()
			w.Write(data[100001:])
			w.Close()
			sum := sha256.Sum256(sink.Bytes())
			lines[fmt.Sprintf("s2stream.%s.%c.16x65536.flush.index.pad4096", st.name, kind)] = hex.EncodeToString(sum[:])
		}
	}
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

type zeroReader struct{}

func (zeroReader) Read(p []byte) (int, error) {
	for i := range p {
		p[i] = 0
	}
	return len(p), nil
}
