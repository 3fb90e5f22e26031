package zstdgpu

import (
	"bytes"
	"os"
	"path/filepath"
	"testing"

	"github.com/klauspost/compress/zstd"
)

// TestBitExact asserts that the GPU frames equal the reference encoder's frames (the parity
// criterion of this project) over the reference's shared fixtures, cut into 128 KiB units.
func TestBitExact(t *testing.T) {
	files, _ := filepath.Glob("../../../testdata/*")
	for _, lvl := range []zstd.EncoderLevel{zstd.SpeedFastest} {
		gpu, err := New(0, WithEncoderLevel(lvl))
		if err != nil {
			t.Fatal(err)
		}
		ref, _ := zstd.NewWriter(nil, zstd.WithEncoderLevel(lvl))
		for _, f := range files {
			data, err := os.ReadFile(f)
			if err != nil || len(data) == 0 {
				continue
			}
			var off []uint64
			for p := 0; p < len(data); p += 128 << 10 {
				off = append(off, uint64(p))
			}
			off = append(off, uint64(len(data)))
			out, outOff, err := gpu.EncodeUnits(data, off, nil)
			if err != nil {
				t.Fatal(err)
			}
			for i := 0; i+1 < len(off); i++ {
				want := ref.EncodeAll(data[off[i]:off[i+1]], nil)
				if !bytes.Equal(out[outOff[i]:outOff[i+1]], want) {
					t.Fatalf("%s unit %d level %v: GPU frame differs from reference", f, i, lvl)
				}
			}
		}
		gpu.Close()
		ref.Close()
	}
}
