package kcgpu_test

// The reference's own goroutine-parallel CPU path on the benchmark corpora (BASELINE.md §4, SURVEY.md §8d):
// the number to print next to bench.py's GPU figure on a host with Go.
//
//	cd shim/go && go test -tags noasm -run xxx -bench . -benchtime 5x
//
// b.SetBytes makes `go test` report MB/s of input; GOMAXPROCS is the core count to quote with it.

import (
	"runtime"
	"sync"
	"testing"

	"github.com/klauspost/compress/kcgpu"
	"github.com/klauspost/compress/s2"
	"github.com/klauspost/compress/zstd"
)

const (
	zUnit  = 128 << 10
	zUnits = 8192 // 1 GiB per iteration
	sBlock = 64 << 10
	sUnits = 16384
)

// parallelUnits runs fn(i) for i in [0, n) on GOMAXPROCS goroutines.
func parallelUnits(n int, fn func(worker, i int)) {
	w := runtime.GOMAXPROCS(0)
	var wg sync.WaitGroup
	next := make(chan int, 1024)
	for k := 0; k < w; k++ {
		wg.Add(1)
		go func(k int) {
			defer wg.Done()
			for i := range next {
				fn(k, i)
			}
		}(k)
	}
	for i := 0; i < n; i++ {
		next <- i
	}
	close(next)
	wg.Wait()
}

func benchZstd(b *testing.B, kind byte, level zstd.EncoderLevel, opts ...zstd.EOption) {
	data, err := kcgpu.CorpusFill(kind, kcgpu.Seed(kind), 0, zUnits, zUnit)
	if err != nil {
		b.Skip(err)
	}
	w := runtime.GOMAXPROCS(0)
	encs := make([]*zstd.Encoder, w)
	dsts := make([][]byte, w)
	for k := range encs {
		o := append([]zstd.EOption{zstd.WithEncoderLevel(level), zstd.WithEncoderConcurrency(1)}, opts...)
		encs[k], err = zstd.NewWriter(nil, o...)
		if err != nil {
			b.Fatal(err)
		}
		dsts[k] = make([]byte, 0, encs[k].MaxEncodedSize(zUnit))
	}
	var out int64
	var mu sync.Mutex
	b.SetBytes(int64(len(data)))
	b.ResetTimer()
	for it := 0; it < b.N; it++ {
		parallelUnits(zUnits, func(k, i int) {
			r := encs[k].EncodeAll(data[i*zUnit:(i+1)*zUnit], dsts[k][:0])
			if it == 0 {
				mu.Lock()
				out += int64(len(r))
				mu.Unlock()
			}
		})
	}
	b.StopTimer()
	b.ReportMetric(float64(out)/float64(len(data)), "ratio")
	b.ReportMetric(float64(w), "goroutines")
}

// C2: zstd SpeedFastest, enwik-style text, 128 KiB units.
func BenchmarkReferenceZstdFastestT(b *testing.B) { benchZstd(b, 'T', zstd.SpeedFastest) }

// C2 on the high-entropy corpus.
func BenchmarkReferenceZstdFastestH(b *testing.B) { benchZstd(b, 'H', zstd.SpeedFastest) }

// C3: zstd SpeedDefault.
func BenchmarkReferenceZstdDefaultT(b *testing.B) { benchZstd(b, 'T', zstd.SpeedDefault) }

// C5: zstd SpeedBetterCompression with a 64 KiB raw dictionary on the mixed corpus.
func BenchmarkReferenceZstdBetterDictM(b *testing.B) {
	dict, err := kcgpu.CorpusFill('T', kcgpu.SeedD, 0, 1, 64<<10)
	if err != nil {
		b.Skip(err)
	}
	benchZstd(b, 'M', zstd.SpeedBetterCompression, zstd.WithEncoderDictRaw(1, dict))
}

// C4: s2.Encode, 64 KiB blocks of JSON.
func BenchmarkReferenceS2EncodeJ(b *testing.B) {
	data, err := kcgpu.CorpusFill('J', kcgpu.SeedJ, 0, sUnits, sBlock)
	if err != nil {
		b.Skip(err)
	}
	w := runtime.GOMAXPROCS(0)
	dsts := make([][]byte, w)
	for k := range dsts {
		dsts[k] = make([]byte, s2.MaxEncodedLen(sBlock))
	}
	b.SetBytes(int64(len(data)))
	b.ResetTimer()
	for it := 0; it < b.N; it++ {
		parallelUnits(sUnits, func(k, i int) {
			s2.Encode(dsts[k], data[i*sBlock:(i+1)*sBlock])
		})
	}
}
